#!/usr/bin/env python3
"""GPU timeline of the training steps from a rocprofv3 kernel trace: per STEP (the interval between two launches of the
step's first kernel, the batched weight fold) how much of the time has no kernel running (dependency / launch gaps), one,
two, three or more; per-queue busy time; the largest idle gaps with the kernels around them.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-config1 --no-render --no-variants
    python tools/timeline.py gpurun_out/tl/*/*_kernel_trace.csv [marker kernel substring, default wn_scale_multi_kernel]
"""
import csv
import statistics
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "wn_scale_multi_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows))
starts = [s for s, e, q, k in ev if marker in k]
if len(starts) < 3:
    raise SystemExit(f"marker kernel {marker!r} found {len(starts)} times")
iv = [(a, b) for a, b in zip(starts, starts[1:])]
med = statistics.median(b - a for a, b in iv)
steps = [(a, b) for a, b in iv if 0.7 * med <= b - a <= 1.3 * med]       # the back-to-back steps (not the ones around a sync)
print(f"{len(iv)} step intervals, median {med / 1e6:.3f} ms; {len(steps)} within 30 % of it are analysed")
hist, busy, span, nk = defaultdict(int), defaultdict(int), 0, 0
gaps = []
for a, b in steps:
    inside = [(max(s, a), min(e, b), q, k) for s, e, q, k in ev if e > a and s < b]
    nk += sum(1 for s, e, q, k in ev if a <= s < b)
    pts = sorted([(s, 1, k) for s, e, q, k in inside] + [(e, -1, k) for s, e, q, k in inside])
    depth, last, last_end_name = 0, a, "(step start)"
    for t, dlt, k in pts:
        hist[min(depth, 3)] += t - last
        if depth == 0 and t - last > 0 and dlt == 1:
            gaps.append((t - last, last_end_name, k))
        depth += dlt
        if dlt == -1 and depth == 0:
            last_end_name = k
        last = t
    hist[min(depth, 3)] += b - last
    for s, e, q, k in inside:
        busy[q] += e - s
    span += b - a
n = len(steps)
print(f"per step: {span / n / 1e6:.3f} ms, {nk / n:.0f} kernels")
for d in sorted(hist):
    print(f"  {d}{'+' if d == 3 else ' '} kernels running: {hist[d] / n / 1e6:7.3f} ms ({100.0 * hist[d] / span:5.1f} %)")
for q, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: busy {v / n / 1e6:7.3f} ms per step ({100.0 * v / span:5.1f} %)")
gaps.sort(reverse=True)
print(f"idle gaps: {len(gaps) / n:.0f} per step, {sum(g for g, _, _ in gaps) / n / 1e6:.3f} ms per step; > 5 us: "
      f"{sum(1 for g, _, _ in gaps if g > 5000) / n:.0f} per step, {sum(g for g, _, _ in gaps if g > 5000) / n / 1e6:.3f} ms")
agg = defaultdict(lambda: [0, 0])
for g, ka, kb in gaps:
    key = (ka.split("(")[0][:60], kb.split("(")[0][:60])
    agg[key][0] += g
    agg[key][1] += 1
print("idle time by (kernel that ended before the gap -> kernel that ended it), us per step:")
for (ka, kb), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {g / n / 1e3:8.1f} us  x{c / n:5.1f}  {ka}  ->  {kb}")
