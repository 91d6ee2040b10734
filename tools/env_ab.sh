#!/bin/bash
# Runtime switches of the HIP graph executor against the bs-16 step (run through gpurun from the repo root).  Every run under
# its own `timeout`: ROC_SYSTEM_SCOPE_SIGNAL=0 hung the replay (round 4) and DEBUG_HIP_FORCE_GRAPH_QUEUES=8 failed the capture;
# 1 / 2 / default(4) queues measured 22.34 / 21.80 / 21.84 ms per step (profiles/r04_env_ab.txt).
mkdir -p gpurun_out/r04h
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config1 --no-render --no-variants --no-roofline"
run() { echo "== $1"; timeout 240 env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config'].get('host_issue_ms_per_step'))" || echo "failed / timed out"; }
run "X=0"
for v in "$@"; do run "$v"; done
run "X=1"
