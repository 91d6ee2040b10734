cd $GRAFT_REPO_ROOT
bash tools/profile.sh r05 > gpurun_out/r05_profile.log 2>&1
tail -n 20 gpurun_out/r05_profile.log
