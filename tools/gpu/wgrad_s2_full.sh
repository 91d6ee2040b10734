#!/bin/bash
mkdir -p gpurun_out/s2
timeout 300 python tools/time_wgrad_s2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2/time.txt
timeout 1500 python -m pytest tests/test_hip_small.py tests/test_hip_models.py tests/test_hip_training.py -q -x > gpurun_out/s2/tests_full.log 2>&1
echo "exit $?" >> gpurun_out/s2/tests_full.log
tail -5 gpurun_out/s2/tests_full.log | cut -c1-250
timeout 600 python bench.py --no-variants --no-config1 --no-render --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH', round(d['value'],1), round(d['ms_per_step'],3))"
