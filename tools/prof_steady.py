#!/usr/bin/env python3
"""Steady-state per-step kernel breakdown from a rocprofv3 rocpd trace of bench.py: uses the last `n` steps, a step
being delimited by the encoder's stem kernel (one launch per encoder pass).
Usage: tools/prof_steady.py results.db [n_steps]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = c.execute("select name, start, end, grid_x, grid_y, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
stem = [r for r in rows if "stem_conv7x7" in r[0] or "stem_s2d" in r[0]]          # (one of the two per encoder pass: the exact stem, or the f16x3 mode's space-to-depth pass)
assert len(stem) >= n + 1, "not enough steps in the trace"
t0, t1 = stem[-n - 1][1], stem[-1][1]
agg = {}
for name, s, e, gx, gy, wx, lds, vg in rows:
    if s < t0 or e > t1:
        continue
    short = name.split("(")[0]
    if "conv_igemm" in name:
        short = "conv_igemm<%s> grid=%dx%d" % (name.split("ConvCfg<")[1].split(">")[0], gx // wx, gy)
    short = short.replace("void ", "").replace("stemseg::", "")
    if len(short) > 100:
        short = short[:97] + "..."
    a = agg.setdefault(short, [0, 0.0, lds, vg])
    a[0] += 1
    a[1] += (e - s) / 1e3
tot = sum(a[1] for a in agg.values())
print("# steady state: %d steps, wall %.2f ms/step, kernel-busy %.2f ms/step, %.1f kernel launches per step (%d kernel names; the 40 busiest below)"
      % (n, (t1 - t0) / 1e6 / n, tot / 1e3 / n, sum(a[0] for a in agg.values()) / n, len(agg)))
print("%-102s %9s %11s %10s %6s %7s %5s" % ("kernel", "calls/stp", "us/step", "avg_us", "%", "lds", "vgpr"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-102s %9.1f %11.1f %10.1f %6.2f %7d %5d" % (k, a[0] / n, a[1] / n, a[1] / a[0], 100 * a[1] / tot, a[2], a[3]))
