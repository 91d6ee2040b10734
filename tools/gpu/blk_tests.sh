#!/bin/bash
mkdir -p gpurun_out/blk
timeout 900 python -m pytest tests/test_hip_blk.py tests/test_hip_render.py -q > gpurun_out/blk/tests.log 2>&1
echo "exit $?" >> gpurun_out/blk/tests.log
tail -25 gpurun_out/blk/tests.log | cut -c1-220
timeout 600 python tools/bench_render.py > gpurun_out/blk/bench_render.txt 2>&1
tail -12 gpurun_out/blk/bench_render.txt | cut -c1-250
VUNET_BLK_FUSE_RNB=0 timeout 600 python tools/bench_render.py > gpurun_out/blk/bench_render_unfused.txt 2>&1
tail -6 gpurun_out/blk/bench_render_unfused.txt | cut -c1-250
