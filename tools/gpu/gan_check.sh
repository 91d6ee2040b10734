#!/bin/bash
mkdir -p gpurun_out/gan
timeout 1500 python -m pytest tests/test_hip_training.py tests/test_hip_models.py -q -x -k "gan or adversarial or disc or graph" > gpurun_out/gan/tests.log 2>&1
echo "exit $?" >> gpurun_out/gan/tests.log
tail -5 gpurun_out/gan/tests.log | cut -c1-250
for t in 1 2; do
timeout 600 python bench.py --gan --no-variants --no-config1 --no-render --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GAN', round(d['value'],1), round(d['ms_per_step'],3), d.get('hip_graph'))"
done
