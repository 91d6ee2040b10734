cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile.sh r05 > gpurun_out/r05_profile.log 2>&1
(timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_full.json 2> gpurun_out/r05_bench_full.err; echo "exit $?" >> gpurun_out/r05_bench_full.err)
(timeout 600 python tools/time_p2.py 2>&1 | grep -v amdgpu > gpurun_out/r05_time_p2.txt)
(timeout 600 python tools/time_p2.py --form 3 2>&1 | grep -v amdgpu > gpurun_out/r05_time_p2_lockstep.txt)
tail -n 5 gpurun_out/r05_profile.log; tail -n 2 gpurun_out/r05_bench_full.err; cut -c1-300 gpurun_out/r05_bench_full.json
