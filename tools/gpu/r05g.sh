cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
(timeout 900 python -m pytest tests/test_hip_p2.py -x -q -m gpu > gpurun_out/r05g/test_p2.log 2>&1; echo "exit $?" >> gpurun_out/r05g/test_p2.log)
(timeout 600 python tools/time_p2.py --form 0 2>&1 | grep -v amdgpu > gpurun_out/r05g/time_p2_form0.txt)
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config1 --no-render --no-variants > gpurun_out/r05g/bench.json 2> gpurun_out/r05g/bench.err; echo "exit $?" >> gpurun_out/r05g/bench.err)
tail -n 4 gpurun_out/r05g/test_p2.log; cat gpurun_out/r05g/time_p2_form0.txt; tail -n 2 gpurun_out/r05g/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r05g/bench.json')); print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['kernel'], r['frac'], r['avg_launch_us'], r['whole_step_frac'])"
