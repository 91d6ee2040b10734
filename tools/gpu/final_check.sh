#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
(timeout 2700 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1; echo "exit $?" >> gpurun_out/final/gpu_tests.log)
tail -n 4 gpurun_out/final/gpu_tests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "exit $?" >> gpurun_out/final/smoke.log); tail -n 4 gpurun_out/final/smoke.log
SECONDS=0
(timeout 1200 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "exit $? after $SECONDS s" >> gpurun_out/final/bench.err)
tail -n 2 gpurun_out/final/bench.err
cut -c1-400 gpurun_out/final/bench.json
