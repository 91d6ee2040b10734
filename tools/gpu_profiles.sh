#!/bin/bash
# Re-take the round's profile set on a GPU box and stamp the summaries with the commit they belong to (run HERE, from the repo root):
#     bash tools/gpu_profiles.sh r06
# writes .head_commit (the box has no .git: tools/pmc_summary.py reads the stamp), runs tools/final_profiles.sh through gpurun and
# copies what came back from gpurun_out/ into profiles/ under the names the docs and bench.py cite.
set -eu
TAG=${1:-r06}
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- behavior_driven_video_synthesis_amd include bench.py tools)" ]; then
  echo "uncommitted changes under the package / include / bench.py / tools: commit first, the stamp would name the wrong tree" >&2
  exit 1
fi
git rev-parse --short=12 HEAD > .head_commit
/usr/local/graft/bin/gpurun --timeout 3000 -- "bash tools/final_profiles.sh $TAG > gpurun_out/final_profiles.log 2>&1; tail -5 gpurun_out/final_profiles.log"
for n in aten_in_step.txt bench_dp_forced.json bench_h2.json pmc_sq.json pmc_traffic.json pmc_traffic_render.json pmc_traffic_seq.json \
         pmc_traffic_seq_train.json seq_kernel_stats.csv seq_time.json seq_train_cvae_time.json seq_train_kernel_stats.csv \
         seq_train_time.json timeline_graph.txt; do
  cp gpurun_out/${TAG}_$n profiles/${TAG}_$n
done
cp gpurun_out/${TAG}_kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats.csv
cp gpurun_out/${TAG}_kernel_stats_1stream.csv profiles/${TAG}_rocprofv3_kernel_stats_1stream.csv
grep -h '"head"' profiles/${TAG}_pmc_*.json | sort | uniq -c
