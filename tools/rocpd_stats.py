#!/usr/bin/env python3
"""Per-kernel statistics (grouped by kernel name and grid) from a rocprofv3 rocpd database (``*_results.db``).

    python tools/rocpd_stats.py gpurun_out/seq/prof/seq_results.db [--top 20] [--csv out.csv]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=20)
    ap.add_argument("--csv")
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, d.grid_size_x / d.workgroup_size_x, d.grid_size_y, d.grid_size_z, count(*), avg(d.end - d.start), "
         f"min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by 1, 2, 3, 4 order by 9 desc limit {args.top}")
    rows = list(c.execute(q))
    lines = ["name,grid_x,grid_y,grid_z,calls,avg_us,min_us,max_us,total_ms,vgprs,agprs"]
    for r in rows:
        name = r[0].replace(".kd", "")
        lines.append(f"\"{name}\",{r[1]},{r[2]},{r[3]},{r[4]},{r[5] / 1e3:.2f},{r[6] / 1e3:.2f},{r[7] / 1e3:.2f},{r[8] / 1e6:.3f},{r[9]},{r[10]}")
        print(f"{name[:64]:64s} grid=({r[1]},{r[2]},{r[3]}) n={r[4]:5d} avg={r[5] / 1e3:7.2f}us min={r[6] / 1e3:6.2f} max={r[7] / 1e3:7.2f} "
              f"tot={r[8] / 1e6:8.2f}ms vgpr={r[9]}")
    if args.csv:
        open(args.csv, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
