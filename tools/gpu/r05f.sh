cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
(timeout 900 python -m pytest tests/test_hip_p2.py -x -q -m gpu > gpurun_out/r05f/test_p2.log 2>&1; echo "exit $?" >> gpurun_out/r05f/test_p2.log)
(timeout 900 python -m pytest tests/test_hip_models.py -x -q -m gpu -k "vgg" > gpurun_out/r05f/test_models_vgg.log 2>&1; echo "exit $?" >> gpurun_out/r05f/test_models_vgg.log)
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config1 --no-render --no-variants > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err; echo "exit $?" >> gpurun_out/r05f/bench.err)
(timeout 1500 python -m pytest tests/test_hip_training.py -x -q -m gpu > gpurun_out/r05f/test_training.log 2>&1; echo "exit $?" >> gpurun_out/r05f/test_training.log)
tail -n 6 gpurun_out/r05f/test_p2.log; tail -n 5 gpurun_out/r05f/test_models_vgg.log; tail -n 3 gpurun_out/r05f/bench.err; tail -n 8 gpurun_out/r05f/test_training.log
python -c "
import json; d=json.load(open('gpurun_out/r05f/bench.json')); print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['kernel'], r['frac'], r['avg_launch_us'], r['whole_step_frac'])"
