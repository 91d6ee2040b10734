#!/bin/bash
mkdir -p gpurun_out/blk
timeout 600 python tools/time_rnb_fused.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/blk/rnb_ab.txt
