#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, grouped by
grid size for the conv kernel.  Usage: tools/prof_summary.py gpurun_out/prof/r01_results.db [skip_first_n_dispatches]"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, grid_y, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
t_first, t_last = rows[0][1], rows[-1][2]
agg = {}
for name, s, e, gx, gy, wx, lds, vg in rows:
    short = name.split("(")[0]
    if "conv_igemm" in name:
        short = "conv_igemm<%s> grid=%dx%d" % (name.split("ConvCfg<")[1].split(">")[0] if "ConvCfg<" in name else "?", gx // wx, gy)
    if len(short) > 110:
        short = short[:107] + "..."
    a = agg.setdefault(short, [0, 0.0, lds, vg])
    a[0] += 1
    a[1] += (e - s) / 1e3
tot = sum(a[1] for a in agg.values())
print("# %d dispatches, %.1f ms of kernel time over a %.1f ms window" % (len(rows), tot / 1e3, (t_last - t_first) / 1e6))
print("%-112s %7s %12s %10s %6s %7s %5s" % ("kernel", "calls", "total_us", "avg_us", "%", "lds", "vgpr"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-112s %7d %12.1f %10.1f %6.2f %7d %5d" % (k, a[0], a[1], a[1] / a[0], 100 * a[1] / tot, a[2], a[3]))
