#!/bin/bash
mkdir -p gpurun_out/blk
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/blk/prof -o render -- python3 $GRAFT_REPO_ROOT/tools/bench_render.py --iters 10 --chunk 50 --no-ref --only "bf16 blocked+shared_appearance" > $GRAFT_REPO_ROOT/gpurun_out/blk/prof.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/blk/prof.log | cut -c1-300
