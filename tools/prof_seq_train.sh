cd /tmp && export TMPDIR=/tmp
export VUNET_SEQ_WAVES4=1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_st4 -o st -- python3 $GRAFT_REPO_ROOT/tools/time_seq_train.py --reps 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_stats.py gpurun_out/prof_st4/st_results.db --top 12
