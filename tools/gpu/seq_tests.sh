#!/bin/bash
# the behaviour front half (csrc/seq.hip): parity tests
mkdir -p gpurun_out/seq
timeout 900 python -m pytest tests/test_hip_seq.py -q > gpurun_out/seq/tests.log 2>&1
echo "exit $?" >> gpurun_out/seq/tests.log
tail -30 gpurun_out/seq/tests.log
