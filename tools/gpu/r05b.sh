cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
(timeout 900 python -m pytest tests/test_hip_p2.py -x -q -m gpu > gpurun_out/r05b/test_p2.log 2>&1; echo "exit $?" >> gpurun_out/r05b/test_p2.log)
(timeout 600 python -m pytest tests/test_hip_dp_smoke.py -x -q -m gpu -k "failed_capture" > gpurun_out/r05b/test_dp.log 2>&1; echo "exit $?" >> gpurun_out/r05b/test_dp.log)
tail -n 30 gpurun_out/r05b/test_p2.log; tail -n 5 gpurun_out/r05b/test_dp.log
