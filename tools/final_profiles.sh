#!/bin/bash
# The round's whole profile set at HEAD (run through gpurun from the repo root):  bash tools/final_profiles.sh r06
# tools/profile.sh (kernel stats 4- / 1-stream, timeline, PMC traffic / SQ, render, seq, seq-train) + the un-profiled timings the
# rocprofv3 runs distort + the ATen census + the bench lines.  Copy gpurun_out/<tag>_* into profiles/ afterwards.
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
bash tools/profile.sh $TAG > $OUT/${TAG}_profile.log 2>&1
python3 tools/time_seq.py --rows 16 > $OUT/${TAG}_seq_time.json 2>> $OUT/${TAG}_profile.log
python3 tools/time_seq.py --rows 64 >> $OUT/${TAG}_seq_time.json 2>> $OUT/${TAG}_profile.log
python3 tools/time_seq_train.py --reps 20 > $OUT/${TAG}_seq_train_time.json 2>> $OUT/${TAG}_profile.log
python3 tools/time_seq_train.py --reps 20 --rows 16 >> $OUT/${TAG}_seq_train_time.json 2>> $OUT/${TAG}_profile.log
python3 tools/time_seq_train.py --stage cvae --reps 20 > $OUT/${TAG}_seq_train_cvae_time.json 2>> $OUT/${TAG}_profile.log
python3 tools/aten_in_step.py > $OUT/${TAG}_aten_in_step.txt 2>> $OUT/${TAG}_profile.log
python3 bench.py > $OUT/${TAG}_bench_h2.json 2> $OUT/${TAG}_bench_h2.err
VUNET_DP_FORCE=1 python3 bench.py --gpus 1 --no-cpu-baseline --no-config1 --no-render --no-variants > $OUT/${TAG}_bench_dp_forced.json 2> $OUT/${TAG}_bench_dp_forced.err
ls -la $OUT/${TAG}_*
