#!/bin/bash
mkdir -p gpurun_out/final
VUNET_DP_FORCE=1 timeout 900 python bench.py --gpus 1 --no-variants --no-config1 --no-render --no-cpu-baseline > gpurun_out/final/bench_dp_forced.json 2> gpurun_out/final/bench_dp_forced.err
echo "exit $?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/final/bench_dp_forced.json').read().strip().splitlines()[-1])
print(round(d['value'],1), d.get('dp_backend'), d.get('rccl_world_size'), d.get('hip_graph'), d.get('allreduce_ms_per_step'), d.get('allreduce_overlap_frac'))
P
