#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of tools/pmc_conv.py (tools/gpu_round.sh pmc: one counter group per run, kernel-trace only)
into profiles/<round>_pmc_conv3d_block4x.txt and profiles/pmc_conv3d_block4x_latest.json (what bench.py reports as
roofline.traffic).  FETCH_SIZE is doubled (gfx950: a wide coalesced read is tallied at half its bytes, MI355X_MICROARCH.md
'HBM'); WRITE_SIZE is taken as reported.  Usage (on the GPU box, after the pmc passes): python tools/pmc_summary.py r02 <reps>"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prec = os.environ.get("PREC", "f32")
latest = "pmc_conv3d_block4x_latest.json" if prec == "f32" else "pmc_conv3d_block4x_%s_latest.json" % prec


def counters(tag):
    dbs = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s*" % tag, "**", "*.db"), recursive=True)
    if not dbs:
        return {}
    con = sqlite3.connect(dbs[0])
    out = {}
    for name, cname, v, n in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        k = "conv_igemm" if "conv_igemm" in name else "splitk_reduce" if "splitk" in name else None
        if k:
            out.setdefault(cname, {}).setdefault(k, [0.0, 0])
            out[cname][k][0] += v
            out[cname][k][1] += n
    return out


fetch, write = counters("FETCH_SIZE").get("FETCH_SIZE", {}), counters("WRITE_SIZE").get("WRITE_SIZE", {})
sq = counters("SQ_WAVES")
sq.update(counters("SQ_LDS_BANK"))
lines = ["# rocprofv3 --pmc passes on tools/pmc_conv.py (block_4x conv of BASELINE configs[1]: Cin 256 -> Cout 128 over [8,120,216]), %d convs per pass, precision %s" % (reps, prec)]
res = {}
if fetch and write:
    f = {k: v[0] * 1024 * 2 / reps for k, v in fetch.items()}      # KB -> bytes, x2 gfx950 correction, per conv
    w = {k: v[0] * 1024 / reps for k, v in write.items()}
    tot = sum(f.values()) + sum(w.values())
    for k in f:
        lines.append("%-14s fetch %.3f GB (raw %.3f x2)  write %.3f GB   per conv, %d launches per conv" % (k, f[k] / 1e9, f[k] / 2e9, w.get(k, 0) / 1e9, fetch[k][1] // max(reps, 1)))
    lines.append("traffic per conv (corrected) %.3f GB vs algorithmic 0.322 GB (input 212.3 MB + weights 3.5 MB read, output 106.2 MB written)" % (tot / 1e9))
    res = {"gb_per_launch_group": round(tot / 1e9, 3), "algorithmic_gb": 0.322,
           "note": "GB per block_4x conv (all its kernel launches incl. the split-K reduce of the planner's cut), FETCH_SIZE x2 + WRITE_SIZE "
                   "from separate rocprofv3 --pmc passes, profiles/%s_pmc_conv3d_block4x.txt" % rnd}
for cname, d in sorted(sq.items()):
    lines.append("%-28s %s   (per conv, summed over XCDs / SIMDs)" % (cname, ", ".join("%s %.4g" % (k, v[0] / reps) for k, v in d.items())))
if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
    busy = sum(v[0] for v in sq["SQ_VALU_MFMA_BUSY_CYCLES"].values()) / 1024.0        # per SIMD (256 CUs x 4)
    act = sum(v[0] for v in sq["GRBM_GUI_ACTIVE"].values()) / 8.0                      # per XCD
    lines.append("MFMA-pipe utilisation         %.1f %%  (busy cycles per SIMD / active cycles per XCD, all kernels of the conv)" % (100.0 * busy / act))
if "SQ_WAIT_ANY" in sq and "SQ_WAVE_CYCLES" in sq:
    wc = sum(v[0] for v in sq["SQ_WAVE_CYCLES"].values())
    lines.append("wave time parked (s_waitcnt / barrier)  %.1f %%, issue-stalled %.1f %%, issuing %.1f %%" % tuple(
        100.0 * sum(v[0] for v in sq[k].values()) / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")))
# effective clock of the conv kernel in the pass that carried GRBM_GUI_ACTIVE: active cycles per XCD / kernel time of the same dispatches (the
# chip clocks to its power budget: MI355X_MICROARCH.md "DVFS give-back"; 2.4 GHz is what the roofline's peak assumes)
if "GRBM_GUI_ACTIVE" in sq:
    try:
        dbs = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_SQ_WAVES*", "**", "*.db"), recursive=True)
        con = sqlite3.connect(dbs[0])
        ns = sum(e - b for name, b, e in con.execute("select name, start, end from kernels") if "conv_igemm" in name)
        cyc = sq["GRBM_GUI_ACTIVE"].get("conv_igemm", [0.0])[0] / 8.0
        if ns > 0 and cyc > 0:
            ghz = cyc / ns
            lines.append("effective clock               %.2f GHz  (GRBM_GUI_ACTIVE per XCD / kernel time of the same dispatches; the roofline's peak assumes 2.4)" % ghz)
            res["effective_clock_ghz"] = round(ghz, 3)
    except Exception as e:      # (older rocprofv3 schema: no kernels view)
        lines.append("effective clock               n/a (%r)" % (e,))
lines += [l.rstrip() for l in open(os.path.join(ROOT, "gpurun_out", "pmc_FETCH_SIZE.log")).read().splitlines() if l.startswith("conv3d_k3")][:1] \
    if os.path.exists(os.path.join(ROOT, "gpurun_out", "pmc_FETCH_SIZE.log")) else []
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "%s_pmc_conv3d_block4x.txt" % rnd), "w").write("\n".join(lines) + "\n")
if res:
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", latest), "w"))
print("\n".join(lines))
