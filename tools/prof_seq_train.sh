#!/bin/bash
# per-kernel times of one stage's training step under rocprofv3 (run through gpurun from the repo root):
#     bash tools/prof_seq_train.sh [flow|cvae]
STAGE=${1:-flow}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_st_$STAGE
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_st_$STAGE -o st -- python3 $GRAFT_REPO_ROOT/tools/time_seq_train.py --stage $STAGE --reps 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py gpurun_out/prof_st_$STAGE/st_results.db --top 16
