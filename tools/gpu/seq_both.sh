#!/bin/bash
bash tools/gpu/seq_tests.sh | tail -5
bash tools/gpu/seq_time.sh 2>&1 | grep -v amdgpu.ids
