#!/bin/bash
mkdir -p gpurun_out/seq
timeout 900 python tools/soak_seq.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/seq/soak.txt
