#!/bin/bash
mkdir -p gpurun_out/seq
timeout 900 python bench.py --steps 10 --warmup 3 --no-variants --no-config1 > gpurun_out/seq/bench_quick.json 2> gpurun_out/seq/bench_quick.err
echo "exit $?"
tail -3 gpurun_out/seq/bench_quick.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/seq/bench_quick.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))
print(json.dumps(d.get('behavior'), indent=1))
print(json.dumps(d.get('render',{}).get('bf16_shared_appearance')))
P
