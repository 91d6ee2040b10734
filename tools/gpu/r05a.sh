cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
(timeout 1500 python -m pytest tests/test_hip_training.py -x -q -m gpu -k "benchmark_size" -s > gpurun_out/r05a/test_bench_size.log 2>&1; echo "exit $?" >> gpurun_out/r05a/test_bench_size.log)
(timeout 900 python -m pytest tests/test_hip_dp_smoke.py -x -q -m gpu > gpurun_out/r05a/test_dp.log 2>&1; echo "exit $?" >> gpurun_out/r05a/test_dp.log)
(timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err; echo "exit $?" >> gpurun_out/r05a/bench.err)
(VUNET_DP_FORCE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --hip-graph on --no-cpu-baseline --no-config1 --no-render --no-variants > gpurun_out/r05a/bench_dp_forced.json 2> gpurun_out/r05a/bench_dp_forced.err; echo "exit $?" >> gpurun_out/r05a/bench_dp_forced.err)
tail -5 gpurun_out/r05a/test_bench_size.log gpurun_out/r05a/test_dp.log
