#!/bin/bash
mkdir -p gpurun_out/seq
timeout 600 python tools/time_seq.py --rows 16 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/seq/prof -o seq -- python3 $GRAFT_REPO_ROOT/tools/time_seq.py --rows 16 --reps 10 > $GRAFT_REPO_ROOT/gpurun_out/seq/prof.log 2>&1
