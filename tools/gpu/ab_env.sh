#!/bin/bash
run() { env $1 timeout 600 python bench.py --no-variants --no-config1 --no-render --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AB', '$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2; do
run "X=0"
run "VUNET_WGRAD_BATCH_PIX=4096"
run "VUNET_WGRAD_BATCH_PIX=8192"
run "VUNET_WN_GROUP=32"
run "VUNET_WN_GROUP=128"
run "VUNET_PACK_OVERLAP=1"
done
