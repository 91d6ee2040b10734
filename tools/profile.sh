#!/bin/bash
# Profiles of the bench.py step on the GPU box (run through gpurun from the repo root):
#
#     gpurun --timeout 1500 -- 'bash tools/profile.sh r02'
#
# 1. rocprofv3 --kernel-trace --stats          -> gpurun_out/<tag>_kernel_stats.csv   (per-kernel time of the timed step, 4 HIP streams)
#                                                 gpurun_out/<tag>_kernel_stats_1stream.csv (same, one stream: exclusive durations)
# 2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE   -> gpurun_out/<tag>_pmc_traffic.json   (HBM bytes per launch; separate
#    passes and the gfx950 FETCH_SIZE x2 correction, as MI355X_MICROARCH.md "HBM" prescribes)
# 3. rocprofv3 --pmc SQ_* (one pass)           -> gpurun_out/<tag>_pmc_sq.json        (MFMA-pipe busy / wave cycles)
# 4. the same FETCH / WRITE passes over tools/bench_render.py -> gpurun_out/<tag>_pmc_traffic_render.json
# The program itself follows `--` (no env / bash -c hop: the profiler initialises the GPU before the program starts).
# Copy the summaries you want judged from gpurun_out/ into profiles/ and commit them.
set -u
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
# (PMC passes and the one-stream trace issue the step eagerly: the same kernels, one dispatch per launch for the counters)
seqtrain() {
# 6. the flow stage's training step (BASELINE config 4): per-kernel times of tools/time_seq_train.py (replayed graph) and the HBM
#    bytes of the step (eager issue, one stream: 3 warm-up + 2 timed steps)
rm -rf $OUT/${TAG}_st $OUT/${TAG}_stfetch $OUT/${TAG}_stwrite
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_st -- python3 tools/time_seq_train.py --reps 10 > $OUT/${TAG}_seq_train_time.json 2> $OUT/${TAG}_st.err
cp $(ls $OUT/${TAG}_st/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_seq_train_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_stfetch -- python3 tools/time_seq_train.py --reps 2 --eager > /dev/null 2> $OUT/${TAG}_stfetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_stwrite -- python3 tools/time_seq_train.py --reps 2 --eager > /dev/null 2> $OUT/${TAG}_stwrite.err
python3 tools/pmc_summary.py traffic $(ls $OUT/${TAG}_stfetch/*/*_counter_collection.csv | head -1) $(ls $OUT/${TAG}_stwrite/*/*_counter_collection.csv | head -1) > $OUT/${TAG}_pmc_traffic_seq_train.json
rm -rf $OUT/${TAG}_st $OUT/${TAG}_stfetch $OUT/${TAG}_stwrite
# the cVAE stage's step at the same configuration's sizes
python3 tools/time_seq_train.py --stage cvae --reps 10 > $OUT/${TAG}_seq_train_cvae_time.json 2>> $OUT/${TAG}_st.err
}
if [ "${2:-}" = "seqtrain" ]; then seqtrain; ls -la $OUT/${TAG}_*; exit 0; fi
BENCH="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-config1 --no-render --no-variants --hip-graph off"

rm -rf $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-config1 --no-render --no-variants > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
cp $(ls $OUT/${TAG}_trace/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null
# per-step GPU timeline of the replayed graph (idle / 1 / 2 / 3+ kernels running, idle gaps by neighbouring kernels)
python3 tools/timeline.py $(ls $OUT/${TAG}_trace/*/*_kernel_trace.csv | head -1) > $OUT/${TAG}_timeline_graph.txt 2>&1
# the same trace with everything on ONE HIP stream: per-kernel durations without the other streams' kernels beside them
# (what bench.py's `roofline` region measures with HIP events; compare its avg_launch_us with the AverageNs here)
export VUNET_TWO_STREAMS=0
rm -rf $OUT/${TAG}_trace1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-config1 --no-render --no-variants --hip-graph off > $OUT/${TAG}_trace1_bench.json 2> $OUT/${TAG}_trace1.err
cp $(ls $OUT/${TAG}_trace1/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_1stream.csv 2>/dev/null
rm -rf $OUT/${TAG}_trace1
unset VUNET_TWO_STREAMS

rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_write.err
python3 tools/pmc_summary.py traffic $(ls $OUT/${TAG}_fetch/*/*_counter_collection.csv | head -1) $(ls $OUT/${TAG}_write/*/*_counter_collection.csv | head -1) > $OUT/${TAG}_pmc_traffic.json

rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/${TAG}_sq -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_sq.err
python3 tools/pmc_summary.py sq $(ls $OUT/${TAG}_sq/*/*_counter_collection.csv | head -1) > $OUT/${TAG}_pmc_sq.json
# 4. the render loop (BASELINE config 5): HBM bytes per launch of its bf16 convolution, same two separate passes
rm -rf $OUT/${TAG}_rfetch $OUT/${TAG}_rwrite
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_rfetch -- python3 tools/bench_render.py --iters 1 --chunk 50 --no-ref --only "bf16 blocked+shared_appearance" > /dev/null 2> $OUT/${TAG}_rfetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_rwrite -- python3 tools/bench_render.py --iters 1 --chunk 50 --no-ref --only "bf16 blocked+shared_appearance" > /dev/null 2> $OUT/${TAG}_rwrite.err
python3 tools/pmc_summary.py traffic $(ls $OUT/${TAG}_rfetch/*/*_counter_collection.csv | head -1) $(ls $OUT/${TAG}_rwrite/*/*_counter_collection.csv | head -1) > $OUT/${TAG}_pmc_traffic_render.json
rm -rf $OUT/${TAG}_rfetch $OUT/${TAG}_rwrite
# 5. the behaviour front half (BASELINE config 5): per-kernel times of tools/time_seq.py and the HBM bytes of the flow's reverse
#    pass (eager issue, one dispatch per launch for the counters; 3 warm-up + 4 timed passes = 7 passes of 31 coupling launches)
rm -rf $OUT/${TAG}_seq $OUT/${TAG}_sfetch $OUT/${TAG}_swrite
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_seq -- python3 tools/time_seq.py --rows 16 --reps 10 > $OUT/${TAG}_seq_time.json 2> $OUT/${TAG}_seq.err
cp $(ls $OUT/${TAG}_seq/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_seq_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_sfetch -- python3 tools/time_seq.py --rows 16 --reps 4 --eager --only reverse > /dev/null 2> $OUT/${TAG}_sfetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_swrite -- python3 tools/time_seq.py --rows 16 --reps 4 --eager --only reverse > /dev/null 2> $OUT/${TAG}_swrite.err
python3 tools/pmc_summary.py traffic $(ls $OUT/${TAG}_sfetch/*/*_counter_collection.csv | head -1) $(ls $OUT/${TAG}_swrite/*/*_counter_collection.csv | head -1) > $OUT/${TAG}_pmc_traffic_seq.json
rm -rf $OUT/${TAG}_seq $OUT/${TAG}_sfetch $OUT/${TAG}_swrite
seqtrain
# the raw per-dispatch tables are large: keep the summaries only
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_sq
ls -la $OUT/${TAG}_*
