#!/usr/bin/env python3
"""One training step of a rocprofv3 kernel trace as a listing: start (us from the step's first kernel), duration, queue, idle
time before the kernel (nothing running on any queue), name.   python tools/step_listing.py TRACE.csv [step_index] [from_us] [to_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("wn_scale_multi_kernel")]
a, b = marks[k], marks[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
busy_until = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = s - busy_until if busy_until and s > busy_until else 0
    if lo <= s / 1e3 <= hi:
        print(f"{s / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  q{r['Queue_Id']}  {'idle %5.1f' % (gap / 1e3) if gap > 2000 else '          '}  "
              f"{r['Kernel_Name'].replace('void ', '')[:70]}  grid {r['Grid_Size_X']}")
    busy_until = max(busy_until, e)
