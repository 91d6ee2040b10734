#!/usr/bin/env python3
"""Where in the step does a kernel sit?  For every dispatch whose name contains PATTERN in a rocprofv3 kernel trace, the
kernels dispatched just before and after it on the same queue, counted.   python tools/neighbours.py TRACE.csv PATTERN [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
cnt = collections.Counter()
short = lambda n: n.replace("void ", "")[:46]
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if pat in r["Kernel_Name"]:
            p = short(rs[i - 1]["Kernel_Name"]) if i else "-"
            n = short(rs[i + 1]["Kernel_Name"]) if i + 1 < len(rs) else "-"
            g = f'{r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?")}'
            cnt[(p, n, g)] += 1
for (p, n, g), c in cnt.most_common(40):
    print(f"{c / steps:7.1f}/step  grid {g:>9s}  {p:46s} -> [{pat}] -> {n}")
