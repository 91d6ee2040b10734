#!/usr/bin/env python3
"""GPU timeline of the timed bench steps from a rocprofv3 kernel trace: how much of the wall time has NO kernel running
(host-bound / dependency gaps), how much has exactly one, and the per-queue busy time.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-config1
    python tools/timeline.py gpurun_out/tl/*/*_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows))
t_end = ev[-1][1]
# last ~60 % of the trace = steady-state steps
t0 = ev[0][0] + int(0.4 * (t_end - ev[0][0]))
ev = [e for e in ev if e[0] >= t0]
span = ev[-1][1] - ev[0][0]
pts = []
for s, e, q, k in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
depth, last, hist = 0, pts[0][0], defaultdict(int)
for t, d in pts:
    hist[min(depth, 3)] += t - last
    depth += d
    last = t
print(f"window {span / 1e6:.1f} ms, kernels {len(ev)}")
for k in sorted(hist):
    print(f"  {k}{'+' if k == 3 else ' '} kernels running: {hist[k] / 1e6:8.2f} ms ({100.0 * hist[k] / span:5.1f} %)")
busy = defaultdict(int)
for s, e, q, k in ev:
    busy[q] += e - s
for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: busy {b / 1e6:8.2f} ms ({100.0 * b / span:5.1f} %)")
# the largest idle gaps and what follows them
gaps = []
cur_end = ev[0][1]
for s, e, q, k in ev[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, k))
    cur_end = max(cur_end, e)
gaps.sort(reverse=True)
print("largest idle gaps (us) and the kernel that ends them:")
for g, k in gaps[:12]:
    print(f"  {g / 1e3:8.1f}  {k[:90]}")
print(f"  total idle in gaps > 5 us: {sum(g for g, _ in gaps if g > 5000) / 1e6:.2f} ms over {sum(1 for g, _ in gaps if g > 5000)} gaps")
