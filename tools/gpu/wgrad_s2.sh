#!/bin/bash
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_hip_small.py -q -k "staged_stride2 or direct_weight_gradient or batched_weight" > gpurun_out/s2/tests.log 2>&1
echo "exit $?" >> gpurun_out/s2/tests.log
tail -15 gpurun_out/s2/tests.log | cut -c1-250
timeout 300 python tools/time_wgrad_s2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2/time.txt
