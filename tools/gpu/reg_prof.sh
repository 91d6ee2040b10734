#!/bin/bash
export BENCH_FLAG=--regressor
bash tools/gpu/gan_prof.sh
