cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
(timeout 900 python -m pytest tests/test_hip_p2.py -x -q -m gpu -s > gpurun_out/r05c/test_p2.log 2>&1; echo "exit $?" >> gpurun_out/r05c/test_p2.log)
(timeout 600 python tools/time_p2.py > gpurun_out/r05c/time_p2.txt 2>&1; echo "exit $?" >> gpurun_out/r05c/time_p2.txt)
(timeout 600 python tools/time_p2.py --form 1 > gpurun_out/r05c/time_p2_form1.txt 2>&1)
(timeout 600 python tools/time_p2.py --form 2 > gpurun_out/r05c/time_p2_form2.txt 2>&1)
(timeout 900 python -m pytest tests/test_hip_models.py -x -q -m gpu -k "vgg" > gpurun_out/r05c/test_models_vgg.log 2>&1; echo "exit $?" >> gpurun_out/r05c/test_models_vgg.log)
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config1 --no-render --no-variants > gpurun_out/r05c/bench.json 2> gpurun_out/r05c/bench.err; echo "exit $?" >> gpurun_out/r05c/bench.err)
(VUNET_VGG_P2=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config1 --no-render --no-variants --no-roofline > gpurun_out/r05c/bench_nop2.json 2> gpurun_out/r05c/bench_nop2.err)
tail -n 12 gpurun_out/r05c/test_p2.log; cat gpurun_out/r05c/time_p2.txt; tail -n 5 gpurun_out/r05c/test_models_vgg.log; tail -n 3 gpurun_out/r05c/bench.err
