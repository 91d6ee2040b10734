cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
(timeout 900 python -m pytest tests/test_hip_p2.py -x -q -m gpu > gpurun_out/r05d/test_p2.log 2>&1; echo "exit $?" >> gpurun_out/r05d/test_p2.log)
for f in 0 1 2 3; do (timeout 600 python tools/time_p2.py --form $f 2>&1 | grep -v amdgpu > gpurun_out/r05d/time_p2_form$f.txt); done
(timeout 900 python -m pytest tests/test_hip_models.py -x -q -m gpu -k "vgg" > gpurun_out/r05d/test_models_vgg.log 2>&1; echo "exit $?" >> gpurun_out/r05d/test_models_vgg.log)
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config1 --no-render --no-variants > gpurun_out/r05d/bench.json 2> gpurun_out/r05d/bench.err; echo "exit $?" >> gpurun_out/r05d/bench.err)
tail -n 6 gpurun_out/r05d/test_p2.log; cat gpurun_out/r05d/time_p2_form2.txt gpurun_out/r05d/time_p2_form3.txt; tail -n 5 gpurun_out/r05d/test_models_vgg.log; tail -n 3 gpurun_out/r05d/bench.err
