#!/bin/bash
mkdir -p gpurun_out/seq
{
timeout 600 python tools/time_seq.py --rows 16
timeout 600 python tools/time_seq.py --rows 16 --eager
timeout 600 python tools/time_seq.py --rows 64
timeout 600 python tools/time_seq.py --rows 1
} > gpurun_out/seq/time.log 2>&1
cat gpurun_out/seq/time.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/seq/prof -o seq -- python3 $GRAFT_REPO_ROOT/tools/time_seq.py --rows 16 --reps 10 > $GRAFT_REPO_ROOT/gpurun_out/seq/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/seq/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
