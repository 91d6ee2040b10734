#!/usr/bin/env python3
"""Summaries of rocprofv3 --pmc passes over bench.py (driven by tools/profile.sh).

    pmc_summary.py traffic <FETCH_SIZE counter csv> <WRITE_SIZE counter csv>   -> per-kernel HBM bytes per launch
    pmc_summary.py sq <SQ_* counter csv>                                       -> per-kernel SQ counter sums per launch

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB and collected in
separate passes; on gfx950 FETCH_SIZE reports one half of the bytes of wide coalesced reads (checked in round 1 on
act_bwd_out_kernel, which reads two tensors and writes one of the same size: FETCH == WRITE in the raw counters), so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES / SQ_BUSY_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (same guide, cycle-constants table).
"""
import collections
import csv
import json
import subprocess
import sys


def head():
    """Commit the profiled tree was built from: `git rev-parse` here, the .head_commit stamp on the GPU box (no .git there;
    written by `git rev-parse --short=12 HEAD > .head_commit` before the gpurun call)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        h = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        if h:
            return h
    except OSError:
        pass
    try:
        return open(os.path.join(root, ".head_commit")).read().strip()
    except OSError:
        return ""


def agg(path):
    """kernel -> counter -> [n, sum]"""
    d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        c = d[k][r["Counter_Name"]]
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    return d


mode = sys.argv[1]
if mode == "traffic":
    f, w = agg(sys.argv[2]), agg(sys.argv[3])
    out = {}
    for k in f:
        n, fv = f[k]["FETCH_SIZE"]
        wn, wv = w.get(k, {}).get("WRITE_SIZE", [0, 0.0])
        if n == 0 or wn == 0:
            continue
        out[k] = {"launches_sampled": n, "fetch_size_kib_per_launch": fv / n, "write_size_kib_per_launch": wv / wn,
                  "hbm_bytes_per_launch": (2.0 * fv / n + wv / wn) * 1024.0}
    print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 4 "
                                "--warmup 2 (tools/profile.sh)",
                      "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
                      "head": head(), "kernels": out}, indent=1))
else:
    s = agg(sys.argv[2])
    out = {}
    for k, cs in s.items():
        e = {"launches_sampled": max(v[0] for v in cs.values())}
        for c, (n, v) in cs.items():
            e[c] = v / max(n, 1)
        if e.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            # MFMA pipe busy cycles summed over SIMDs vs 4 SIMDs x busy quad-cycles x 4 cycles
            e["mfma_busy_over_wave_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * e["SQ_WAVE_CYCLES"]) if e.get("SQ_WAVE_CYCLES") else None
        out[k] = e
    print(json.dumps({"source": "rocprofv3 --pmc SQ_* (one pass) of bench.py --steps 4 --warmup 2 (tools/profile.sh)",
                      "units": "per launch; SQ_VALU_MFMA_BUSY_CYCLES in cycles, the other SQ cycle counters in quad-cycles, summed over the chip",
                      "head": head(), "kernels": out}, indent=1))
