#!/bin/bash
mkdir -p gpurun_out/final
timeout 900 python bench.py --no-variants --no-config1 --no-render --no-cpu-baseline > gpurun_out/final/bench_short.json 2> /dev/null
python - <<'P'
import json
d=json.loads(open('gpurun_out/final/bench_short.json').read().strip().splitlines()[-1])
print("SPREAD", round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['whole_step_frac'],4))
P
