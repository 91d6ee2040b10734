mkdir -p gpurun_out/r04h
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config1 --no-render --no-roofline"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config'].get('host_issue_ms_per_step'))"; }
run "X=0"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=2"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=8"
run "ROC_SYSTEM_SCOPE_SIGNAL=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "X=1"
