#!/bin/bash
mkdir -p gpurun_out/gan
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/gan/prof -o gan -- python3 $GRAFT_REPO_ROOT/bench.py --gan --steps 8 --warmup 3 --no-variants --no-config1 --no-render --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/gan/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/gan/bench.err
cd $GRAFT_REPO_ROOT
tail -c 600 gpurun_out/gan/bench.json
