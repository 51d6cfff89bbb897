#!/usr/bin/env python3
"""Lane-soak probe: L captured pipelines (own workspaces, own streams) replay the bench step concurrently for many rounds; after
every round EVERY lane-owned buffer (encoder workspace by plan segment, zero-haloed FPN inputs, decoder workspaces by plan
segment, step outputs) is compared bitwise with the same lane's lone replay of the same input batch.  Prints the rate and, for a
round that differs, which buffers differ, where (segment, channel / voxel range) and by how much -- the earliest segment in
execution order is where the difference entered.

    python tools/soak_probe.py --workload ytvis --lanes 3 --reps 200 [--precision f16x3] [--small]

Memory: the probe keeps TWO reference copies of everything a lane owns (one per input batch).  At the DAVIS / YouTube-VIS sizes that is a few
tens of GB; at the KITTI size (608 x 1952) with three lanes it asks for more than the 288 GB of the card (torch raises OutOfMemoryError
before anything runs) -- use --small or two lanes there; the -m gpu soaks (tests/test_gpu_soak.py) compare signatures, not buffers.
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import bench  # noqa: E402
from stemseg_amd import hip  # noqa: E402

ENC_NAMES = ["S0", "X1", "A", "B"] + ["Cst%d" % i for i in range(4)] + ["M1_%d" % i for i in range(4)] + ["M2", "DS", "XS"] + \
            ["L%d" % i for i in range(4)] + ["FO%d" % i for i in range(4)] + ["SK", "total"]


def ru(a, b):
    return (a + b - 1) // b * b


def padded_total(Cn, T, H, W):
    pitch = ru(W + 2, 4)
    return Cn * (T + 2) * (H + 2) * pitch + 64


def decoder_segments(mod, T, H4, W4, layout, nb=1):
    """csrc/decoder.hip make_plan, restated: [(name, offset in floats)] in workspace order -- ``nb`` clip plans (c<clip>.<slice>), then the
    split-K scratch the clips share."""
    cin, (c32, c16, c8, c4) = mod.in_channels, mod.inter_channels
    h = [H4 >> (3 - i) for i in range(4)]
    w = [W4 >> (3 - i) for i in range(4)]
    pool, ts = [int(bool(f)) for f in mod.pool_flags], list(mod.t_scales)
    pooled = lambda t, on: (t + 1) // 2 if on else t
    Ta1 = pooled(T, pool[0]); Ta2 = pooled(Ta1, pool[1]); Ta3 = pooled(Ta2, pool[2])
    Tb1 = pooled(T, pool[0])
    T16 = Ta3 * ts[0]; T8 = T16 * ts[1]
    segs, off = [], 0

    def take(name, floats):
        nonlocal off
        segs.append((name, off))
        off += ru(floats, 64) + 64              # (+ the slice's guard block: csrc/common.h, workspace canaries)
    if layout != 2:
        for i in range(4):
            take("pin%d" % i, padded_total(cin, T, h[i], w[i]))
    cs = [c32, c16, c8, c4]
    for i in range(4):
        take("D%d" % i, cs[i] * T * h[i] * w[i])
    take("P32b", padded_total(c32, Ta1, h[0], w[0])); take("P32c", padded_total(c32, Ta2, h[0], w[0]))
    take("X32", c32 * Ta3 * h[0] * w[0]); take("cat16", (c32 + c16) * T16 * h[1] * w[1])
    take("P16b", padded_total(c16, Tb1, h[1], w[1])); take("X16", c16 * T16 * h[1] * w[1])
    take("cat8", (c16 + c8) * T8 * h[2] * w[2]); take("X8", c8 * T8 * h[2] * w[2])
    take("cat4", (c8 + c4) * T * h[3] * w[3]); take("X4", c4 * T * h[3] * w[3])
    for i in range(4):
        take("stats%d" % i, 128)
        take("gn_scratch%d" % i, 2 * max((mod.gn_groups or 1) * 32768 * 2, (mod.gn_groups or 1) * 128))
    clip_floats = ru(off, 256)
    segs = [("c%d.%s" % (c, n), c * clip_floats + o) for c in range(nb) for n, o in segs]
    off = nb * clip_floats
    for i, k in enumerate((16, 4, 4, 2)):
        take("S%d" % i, nb * k * cs[i] * T * h[i] * w[i])
    segs.append(("total", off))
    return segs


def lane_tensors(pipe, graph, lane, NC):
    """{name: (tensor, [(segment, float offset)] | None)} of everything lane ``lane`` owns."""
    m = pipe.model._model
    out = {}
    bb = m.backbone
    for k, v in bb._ws.items():
        if k[4] != lane:
            continue
        offs = (C.c_int64 * 25)()
        hip.check(hip.lib().stemseg_hip_encoder_plan_offsets(C.byref(bb._desc(k[0], k[1], k[2], NC)), offs))
        segs = sorted((o, n) for o, n in zip(list(offs), ENC_NAMES) if o >= 0)
        out["encoder_ws"] = (v, [(n, o) for o, n in segs])
    for k, blk in sorted(pipe.model._pads.items()):
        if k[-1] == lane:
            for slot, v in enumerate(blk["pads"]):
                for lvl, (bf, _) in enumerate(v):
                    out["fpn_pad_slot%d_%dx" % (slot, (32, 16, 8, 4)[lvl])] = (bf, None)
    for name in ("embedding_head", "seediness_head", "semseg_head"):
        mod = getattr(m, name)
        if mod is None:
            continue
        for k, v in mod._workspaces.items():
            if k[5] == lane:
                out["%s_ws_nb%d" % (name, k[6])] = (v, decoder_segments(mod, k[0], k[1], k[2], k[3], k[6]))
    for i, o in enumerate(graph.out):
        n = o["frame_offsets"][-1:] if "frame_offsets" in o else None      # (device scalar: rows beyond N are never written)
        for kk, v in o.items():
            if torch.is_tensor(v):
                if n is not None and kk in ("voxel_index", "labels"):
                    v = torch.where(torch.arange(v.numel(), device=v.device) < n, v.reshape(-1), torch.zeros_like(v.reshape(-1)))
                out["out%d.%s" % (i, kk)] = (v, None)
    return out


def as_i32(t):
    t = t.reshape(-1)
    if t.dtype == torch.uint8:
        n = t.numel() // 4 * 4
        return t[:n].view(torch.int32)
    if t.element_size() == 8:
        return t.view(torch.int32)
    return t.view(torch.int32)


def describe(name, t, ref, segs, other=None):
    a, b = as_i32(t), as_i32(ref)
    c = as_i32(other) if other is not None else None
    bad = torch.nonzero(a != b).flatten()
    if bad.numel() == 0:
        return None
    lines = []
    if bad.numel() <= 64:                                  # few words: list them (offset, got, lone-replay value)
        fa, fb = a[bad].view(torch.float32).tolist(), b[bad].view(torch.float32).tolist()
        lines.append("    words: " + ", ".join("+%d %.6g (lone %.6g)" % (int(o), x, y) for o, x, y in zip(bad.tolist(), fa, fb)))
    if segs:
        bounds = [o for _, o in segs]
        for (sn, o), o2 in zip(segs[:-1], bounds[1:]):
            sel = bad[(bad >= o) & (bad < o2)]
            if sel.numel():
                fa, fb = a[sel].view(torch.float32), b[sel].view(torch.float32)
                d = (fa - fb).abs()
                lines.append("    %-12s %9d of %11d words differ, first +%d last +%d, max |d| %.3e (ref magnitude %.3e)"
                             % (sn, sel.numel(), o2 - o, int(sel[0]) - o, int(sel[-1]) - o, float(d.max()), float(fb.abs().max())))
                if sel.numel() <= 40:                      # few words: list (offset in the segment, got, lone replay, lone replay of the OTHER batch)
                    fc = c[sel].view(torch.float32).tolist() if c is not None else [float("nan")] * sel.numel()
                    lines.append("        " + "; ".join("+%d got %.7g lone %.7g other-batch %.7g" % (int(q) - o, x, y, z)
                                                          for q, x, y, z in zip(sel.tolist(), fa.tolist(), fb.tolist(), fc)))
                    # the whole 32-word neighbourhood of the first differing word: got / lone / other
                    q0 = int(sel[0]) // 16 * 16
                    nb = slice(q0, q0 + 32)
                    lines.append("        neighbourhood +%d..: got   %s" % (q0 - o, " ".join("%.5g" % v for v in a[nb].view(torch.float32).tolist())))
                    lines.append("        neighbourhood +%d..: lone  %s" % (q0 - o, " ".join("%.5g" % v for v in b[nb].view(torch.float32).tolist())))
                    if c is not None:
                        lines.append("        neighbourhood +%d..: other %s" % (q0 - o, " ".join("%.5g" % v for v in c[nb].view(torch.float32).tolist())))
    else:
        fa, fb = a[bad].view(torch.float32), b[bad].view(torch.float32)
        lines.append("    %9d of %11d words differ, first %d last %d, max |d| %.3e" % (bad.numel(), a.numel(), int(bad[0]), int(bad[-1]),
                                                                                   float((fa - fb).abs().max()) if t.dtype == torch.float32 else -1.0))
    return "  %s:\n%s" % (name, "\n".join(lines))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ytvis")
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--nc", type=int, default=4)
    ap.add_argument("--small", action="store_true", help="quarter-size frames (faster rounds)")
    ap.add_argument("--backbone", default=None)
    ap.add_argument("--max-reports", type=int, default=4)
    ap.add_argument("--same-batch", action="store_true", help="all lanes replay the same batch every round")
    args = ap.parse_args()
    if args.backbone:
        bench.BACKBONE = args.backbone
    if args.small:
        for wl in bench.WORKLOADS.values():
            wl["H"], wl["W"] = wl["H"] // 2 // 32 * 32, wl["W"] // 2 // 32 * 32
            wl["valid"] = (wl["H"], wl["W"])
    bench.select_workload(args.workload)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    hip.require_gpu()
    pipe, _ = bench.build_pipeline(dev)
    pipe.model.set_precision(args.precision)
    pipe.model.overlap_decoders = False
    NC = args.nc
    batches = [torch.cat([bench.make_clip(1000 + i * NC + c, dev) for c in range(NC)], 0) for i in range(2)]
    pipe.step_batch(batches[0], NC)
    torch.cuda.synchronize()
    lanes = [pipe.capture(batches[0], n_clips=NC, lane=k) for k in range(args.lanes)]
    refs = []
    for k, g in enumerate(lanes):
        per = []
        for b in range(2):
            g.run(batches[b])
            torch.cuda.synchronize()
            per.append({n: t.clone() for n, (t, _) in lane_tensors(pipe, g, k, NC).items()})
        # a second lone replay must reproduce the first (sanity: the lone lane is bit-stable)
        g.run(batches[1])
        torch.cuda.synchronize()
        for n, (t, s) in lane_tensors(pipe, g, k, NC).items():
            if not torch.equal(as_i32(t), as_i32(per[1][n])):
                print("LONE replay of lane %d differs in %s" % (k, n))
        refs.append(per)
    print("soak: workload %s%s, %s, %d lanes x %d rounds, %d clips per step, precision %s"
          % (args.workload, " (half size)" if args.small else "", bench.BACKBONE, args.lanes, args.reps, NC, args.precision), flush=True)
    bad_rounds, reports, first_seg = 0, 0, {}
    results = 0
    for rep in range(args.reps):
        which = [0 if args.same_batch else (rep + k) % 2 for k in range(args.lanes)]
        for k, g in enumerate(lanes):
            g.run_async(batches[which[k]])
        for g in lanes:
            g.wait()
        torch.cuda.synchronize()
        for k, g in enumerate(lanes):
            results += NC
            cur = lane_tensors(pipe, g, k, NC)
            ref = refs[k][which[k]]
            differing = [n for n, (t, _) in cur.items() if not torch.equal(as_i32(t), as_i32(ref[n]))]
            if differing:
                bad_rounds += 1
                key = differing[0]
                first_seg[key] = first_seg.get(key, 0) + 1
                if reports < args.max_reports:
                    reports += 1
                    print("round %d lane %d (batch %d): %d buffers differ from the lone replay" % (rep, k, which[k], len(differing)), flush=True)
                    for n in differing:
                        d = describe(n, cur[n][0], ref[n], cur[n][1], refs[k][1 - which[k]].get(n))
                        if d:
                            print(d, flush=True)
    print("RESULT: %d of %d lane-rounds differ (%d clip results checked); first differing buffer: %s" % (bad_rounds, args.reps * args.lanes, results, first_seg))


if __name__ == "__main__":
    main()
