cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2700 python -m pytest tests/ -x -q -m gpu > gpurun_out/full_gpu_tests.log 2>&1; echo "exit $?" >> gpurun_out/full_gpu_tests.log)
tail -n 12 gpurun_out/full_gpu_tests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "exit $?" >> gpurun_out/smoke.log); tail -n 3 gpurun_out/smoke.log
