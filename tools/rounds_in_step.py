#!/usr/bin/env python3
"""Per kernel form of the training step: workgroups per launch against the chip's resident slots (256 CUs x the occupancy the
launch's LDS / registers / waves allow) -- launches that run a nearly empty last round of workgroups show up as a small fractional
part of `rounds`.  Input: the kernel trace of a rocprofv3 --kernel-trace run of bench.py (csv).  The trace's LDS size is the
STATIC allocation only: for kernels with dynamic LDS (conv_h2 / p2 / wgrad families) `resident/CU` is an upper bound; the workgroup
counts are exact.

    python tools/rounds_in_step.py <..._kernel_trace.csv> [top N]
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = rows[len(rows) * 2 // 3:]          # the timed steps
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"][:72]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    lds = int(r["LDS_Block_Size"])
    vg = int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count", 0) or 0)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault((name, wg, grid // wg, lds, vg), [0, 0.0])
    a[0] += 1
    a[1] += dur
out = []
for (name, wg, wgs, lds, vg), (n, t) in agg.items():
    waves = max(1, wg // 64)
    occ_l = 160 * 1024 // lds if lds else 99
    per_simd = max(1, min(8, 512 // max(vg, 1)))
    occ_v = max(1, (per_simd * 4) // waves)
    occ = max(1, min(occ_l, occ_v, 32 // waves))
    out.append((t, name, wg, wgs, lds, vg, occ, wgs / (256 * occ), n))
out.sort(reverse=True)
for t, name, wg, wgs, lds, vg, occ, rounds, n in out[:top]:
    print(f"{t / n:8.1f} us x{n:3d} total {t:8.0f} us  wgs {wgs:6d} x {wg:4d} thr  lds {lds:6d}  regs {vg:3d}  resident/CU {occ}  rounds {rounds:6.2f}  {name}")
