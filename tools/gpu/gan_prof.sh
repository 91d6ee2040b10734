#!/bin/bash
mkdir -p gpurun_out/gan
cd /tmp && export TMPDIR=/tmp
export VUNET_TWO_STREAMS=0
BENCH_FLAG=${BENCH_FLAG:---gan}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/gan/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gan/prof -- python3 $GRAFT_REPO_ROOT/bench.py $BENCH_FLAG --steps 10 --warmup 3 --no-variants --no-config1 --no-render --no-roofline --no-cpu-baseline --hip-graph off > $GRAFT_REPO_ROOT/gpurun_out/gan/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/gan/bench.err
cd $GRAFT_REPO_ROOT
cp $(ls gpurun_out/gan/prof/*/*_kernel_stats.csv | head -1) gpurun_out/gan/kernel_stats_1stream.csv
rm -rf gpurun_out/gan/prof
tail -c 300 gpurun_out/gan/bench.json
