# one-stream kernel trace of the step (exclusive per-kernel durations): gpurun_out/<tag>_kernel_stats_1stream.csv
TAG=${1:-r05x}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
export VUNET_TWO_STREAMS=0
rm -rf $OUT/${TAG}_trace1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-config1 --no-render --no-variants --hip-graph off > $OUT/${TAG}_trace1_bench.json 2> $OUT/${TAG}_trace1.err
cp $(ls $OUT/${TAG}_trace1/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_1stream.csv
rm -rf $OUT/${TAG}_trace1
head -45 $OUT/${TAG}_kernel_stats_1stream.csv | cut -c1-150
