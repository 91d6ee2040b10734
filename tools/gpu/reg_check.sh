#!/bin/bash
mkdir -p gpurun_out/gan
timeout 1500 python -m pytest tests/test_hip_training.py tests/test_hip_models.py -q -x > gpurun_out/gan/tests_reg.log 2>&1
echo "exit $?" >> gpurun_out/gan/tests_reg.log
tail -4 gpurun_out/gan/tests_reg.log | cut -c1-250
for f in "--regressor" "--regressor --gan"; do
timeout 600 python bench.py $f --no-variants --no-config1 --no-render --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VAR', '$f', round(d['value'],1), round(d['ms_per_step'],3))"
done
