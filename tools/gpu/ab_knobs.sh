#!/bin/bash
for i in 1 2; do
for t in "" "split_force_nt=2" "split_force_nt=4" "parity_launches=1" "p2_form=2" "s2_fwd_f32=1"; do
VUNET_TUNING="$t" timeout 600 python bench.py --no-variants --no-config1 --no-render --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AB', '$t' or 'default', round(d['value'],1), round(d['ms_per_step'],3))"
done
done
