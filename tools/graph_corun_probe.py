#!/usr/bin/env python3
"""Reproducer attempt for the round-3/4 lane differences: TWO SMALL hipGraphs replayed
concurrently on two streams -- a victim that is one launch of the encoder's stem kernel, and an aggressor that is a short run of the
library's MFMA convolutions -- with the victim's output compared bitwise with its lone replay after every round.

The failures of rounds 3 / 4 were only ever seen with whole captured pipelines replaying concurrently, and every one of them started
in the VALU stem kernel; the two stand-alone probes of round 4 (tools/stem_corun_probe.py, tools/microbench/valu_corun_probe.hip)
launched EAGERLY and stayed clean.  This probe closes that gap: captured victim x captured aggressor, plus the three mixed /
eager combinations as controls, for both stem forms.

The VALU stem is not in the product library.  Build the experiment library and point the probe at it:

    STEMSEG_BUILD_DEFINES=-DSS_EXPERIMENTS STEMSEG_BUILD_TAG=exp python stem-seg_amd/build.py
    STEMSEG_HIP_LIB=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_exp.so STEMSEG_STEM=valu python tools/graph_corun_probe.py --rounds 300
    STEMSEG_HIP_LIB=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_exp.so python tools/graph_corun_probe.py --rounds 300      # MFMA stem

Prints one line per (victim mode, aggressor mode, aggressor kind): rounds with a differing victim output, and for the first few the
word offsets decoded to (channel, frame, row, column run) -- the round-4 signature was 5-13 wrong words inside ONE 16-column run.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
from stemseg_amd import hip  # noqa: E402


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)).cuda()


def make_victim(T, H, W, seed=1):
    frames = (torch.from_numpy(np.random.RandomState(seed).randint(0, 256, (T, 3, H, W)).astype(np.float32)) - 110.0).cuda().contiguous()
    wt = rnd((64, 3, 7, 7), 2, 1.0 / np.sqrt(147.0)).reshape(64, 147).t().contiguous()      # tap-major, as stemseg_hip_stem_conv takes it
    b = rnd((64,), 3)
    out = torch.empty(64, T, H // 2, W // 2, dtype=torch.float32, device="cuda")

    def run():                                                 # exactly ONE kernel: the stem
        hip.check(hip.lib().stemseg_hip_stem_conv(hip.ptr(frames), hip.ptr(wt), hip.ptr(b), hip.ptr(out), T, H, W, hip.stream()))
        return out
    return run


def make_product_victim(kind):
    """The product's own VALU-bound kernels as victims (do they show what the VALU stem shows?): the fused heads (128-channel dot
    products + activations), the trilinear x2 up-sampling, GroupNorm-apply + ReLU + AvgPool3d -- at the decoders' 4x / 8x sizes."""
    if kind == "heads":
        x = rnd((128, 8, 120, 216), 31)
        w, b = rnd((6, 128), 32, 1.0 / np.sqrt(128.0)), rnd((6,), 33)
        return lambda: hip.heads(x, w, b, [0, 0, 2, 2, 3, 3], [0] * 6, None, None, None)
    if kind == "upsample":
        x = rnd((128, 8, 60, 108), 34)
        out = torch.empty(128, 8, 120, 216, device="cuda")

        def run():
            hip.upsample_trilinear(x, 1, 2, 2, hip.dense_volume(out))
            return out
        return run
    if kind == "gn":
        x = rnd((128, 8, 60, 108), 35)
        stats = torch.stack([rnd((32,), 36, 0.1), 1.0 + rnd((32,), 37, 0.1).abs()], 1).reshape(-1).contiguous()
        gam, bet = 1.0 + rnd((128,), 38, 0.1), rnd((128,), 39, 0.1)
        out = torch.empty(128, 4, 60, 108, device="cuda")

        def run():
            hip.gn_relu_pool(x, 32, stats, gam, bet, 1, hip.dense_volume(out))
            return out
        return run
    raise ValueError(kind)


def make_aggressor(kind, prec):
    """A run of library convolutions (MFMA streams with LDS staging and barriers), ~1-2 ms per replay."""
    ops = []
    if kind in ("k3", "mix"):
        Cin, Cout, T, H, W = 256, 128, 8, 120, 216               # the decoders' block_4x convolution
        buf, g = hip.alloc_padded(Cin, T, H, W)
        hip.copy_to_volume(rnd((Cin, T, H, W), 5), 0, hip.padded_interior_view(buf, g, Cin, T, H, W))
        pw = hip.pack_conv_weight_any(rnd((Cout, Cin, 3, 3, 3), 6, 1.0 / np.sqrt(Cin * 27.0)), prec)
        out = torch.empty(Cout, T, H, W, device="cuda")
        ops.append(lambda: hip.conv3d(hip.padded_halo_view(buf, g, Cin, T, H, W), pw, None, hip.dense_volume(out), 3, 0, None, dict(precision=prec)))
        outs = [out]
    else:
        outs = []
    if kind in ("k1", "mix", "k1_f32", "k1_bf16x6", "k1_big", "k1_wide"):
        p1 = {"k1_f32": "f32", "k1_bf16x6": "bf16x6"}.get(kind, prec)
        tcfg = {"k1_big": 1, "k1_wide": 3}.get(kind, 0)        # 0: the launcher's choice (128 co x 128 voxels, four waves); 1: 128 x 256; 3: 256 x 256 on eight waves
        Cin, Cout, V = 256, 1024, 32 * 30 * 54                   # a layer-3 expansion of the encoder
        x = rnd((Cin, V), 7)
        pw1 = hip.pack_conv_weight_any(rnd((Cout, Cin, 1, 1, 1), 8, 1.0 / np.sqrt(Cin)), p1)
        o1 = torch.empty(Cout, V, device="cuda")
        for _ in range(4):
            ops.append(lambda: hip.conv3d(hip.flat_volume(x), pw1, None, hip.flat_volume(o1), 1, tcfg, None, dict(precision=p1, relu=1)))
        outs.append(o1)
    if kind in ("k2", "k2flat"):                                 # a layer-3 3x3 convolution: k2 = the four-wave 4-row tile (no scratch), k2flat = the product's eight-wave flat tile
        Cin, Cout, T, H, W = 256, 256, 32, 30, 54
        pitch = (W + 2 + 3) // 4 * 4
        buf = torch.zeros(Cin, T, H + 2, pitch, device="cuda")
        buf[:, :, 1:H + 1, 1:W + 1] = rnd((Cin, T, H, W), 9)
        vin = hip.Volume(buf.data_ptr(), T * (H + 2) * pitch, (H + 2) * pitch, pitch, Cin, T, H + 2, W + 2, buf.numel())
        outs.append(buf)          # (keeps the input alive: the Volume holds a bare pointer -- round 5's first run of this kind compared freed memory)
        pw2 = hip.pack_conv_weight_any(rnd((Cout, Cin, 1, 3, 3), 10, 1.0 / np.sqrt(Cin * 9.0)), prec)
        o2 = torch.empty(Cout, T, H, W, device="cuda")
        sc2 = torch.empty(32 << 20, device="cuda") if kind == "k2flat" else None
        for _ in range(3):
            ops.append(lambda: hip.conv3d(vin, pw2, None, hip.dense_volume(o2), (1, 3, 3), 0, sc2, dict(precision=prec, relu=1)))
        outs.append(o2)
    if kind == "stream":                                         # no matrix cores, no LDS: a streaming kernel (trilinear x2)
        xs = rnd((128, 8, 60, 108), 11)
        o3 = torch.empty(128, 16, 120, 216, device="cuda")
        for _ in range(6):
            ops.append(lambda: hip.upsample_trilinear(xs, 2, 2, 2, hip.dense_volume(o3)))
        outs.append(o3)
    if kind == "stem":                                           # a second instance of the victim's own kernel
        v2 = make_victim(16, 480, 864, seed=21)
        ops.append(v2)
        outs.append(v2())

    def run():
        for f in ops:
            f()
        return outs
    return run


class Replayable(object):
    """fn() either captured into a hipGraph (replayed on a stream) or launched eagerly on that stream."""

    def __init__(self, fn, captured):
        self.fn, self.captured = fn, captured
        self.stream = torch.cuda.Stream()
        with torch.cuda.stream(self.stream):
            self.out = fn()
        torch.cuda.synchronize()
        if captured:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                self.out = fn()
            torch.cuda.synchronize()

    def go(self, times=1):
        with torch.cuda.stream(self.stream):
            for _ in range(times):
                if self.captured:
                    self.graph.replay()
                else:
                    self.out = self.fn()
        return self.out

    def go_sync(self):
        """one run, finished: the outputs may be read from any stream"""
        out = self.go()
        self.stream.synchronize()
        return out


def same_bits(a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


def decode(idx, T, Ho, Wo):
    ch, r = divmod(int(idx), T * Ho * Wo)
    t, r = divmod(r, Ho * Wo)
    y, x = divmod(r, Wo)
    return "c%d t%d y%d x%d" % (ch, t, y, x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=300)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--aggressors", default="k3,k1,mix", help="comma list of k3 | k1 | k1_big | k1_wide | mix | k1_f32 | k1_bf16x6 | k2 | k2flat | stream | stem")
    ap.add_argument("--modes", default="gg,ge,eg,ee", help="(victim, aggressor) launch modes: g = captured graph, e = eager")
    ap.add_argument("--victims", default="stem", help="comma list of stem | heads | upsample | gn (the last three: the product's own VALU-bound kernels)")
    args = ap.parse_args()
    hip.require_gpu()
    T, H, W = args.frames, 480, 864
    Ho, Wo = H // 2, W // 2
    stem = "VALU (experiment build)" if os.environ.get("STEMSEG_STEM", "").startswith("v") else "MFMA (product)"
    print("graph_corun_probe: stem kernel %s, library %s, %d rounds per combination, aggressors in %s" % (stem, hip.LIB_PATH, args.rounds, args.precision), flush=True)
    total_bad = 0
    for vic_kind, agg_kind in [(v_, a_) for v_ in args.victims.split(",") for a_ in args.aggressors.split(",")]:
        for mode in args.modes.split(","):
            v_cap, a_cap = mode[0] == "g", mode[1] == "g"
            if True:
                vic = Replayable(make_victim(T, H, W) if vic_kind == "stem" else make_product_victim(vic_kind), v_cap)
                agg = Replayable(make_aggressor(agg_kind, args.precision), a_cap)
                ref = vic.go_sync().clone()
                torch.cuda.synchronize()
                agg_ref = [o.clone() for o in agg.go_sync()]
                torch.cuda.synchronize()
                # controls: both kernels are bit-stable when they run ALONE (compared as bit patterns: NaNs would not compare equal as floats)
                lone_v = sum(int(not same_bits(vic.go_sync(), ref)) for _ in range(3))
                torch.cuda.synchronize()
                lone_a = sum(int(any(not same_bits(a, b) for a, b in zip(agg.go_sync(), agg_ref))) for _ in range(3))
                torch.cuda.synchronize()
                nonfinite = sum(int((~torch.isfinite(o)).sum()) for o in agg_ref) + int((~torch.isfinite(ref)).sum())
                bad, bad_agg, notes = 0, 0, []
                hist_q, hist_pix, hist_reg = [0, 0, 0, 0], [0, 0], {}
                for r in range(args.rounds):
                    agg.go(2)                                  # the aggressor brackets the victim in time
                    out = vic.go()
                    agg.go(2)
                    torch.cuda.synchronize()
                    if not same_bits(out, ref):
                        bad += 1
                        idx = torch.nonzero((out.view(torch.int32) != ref.view(torch.int32)).reshape(-1)).reshape(-1)
                        if bad <= 20 and vic_kind == "stem":
                            # VALU stem: thread (py, px) of a 256-thread workgroup owns output columns px and px + 32 of row py of an 8 x 64 tile and
                            # all 64 channels (two passes of 32: register k = channel % 32, acc0 / acc1 = the two columns); lane = (py & 1) * 32 + px
                            ii = idx.cpu().numpy()
                            ch, rem = np.divmod(ii, T * Ho * Wo)
                            yy, xx = np.divmod(rem % (Ho * Wo), Wo)
                            lane = (yy % 2) * 32 + (xx % 32)
                            for q_ in range(4):
                                hist_q[q_] += int(((lane >> 4) == q_).sum())
                            hist_pix[0] += int(((xx % 64) < 32).sum())
                            hist_pix[1] += int(((xx % 64) >= 32).sum())
                            for k_ in (ch % 32).tolist()[:2000]:
                                hist_reg[k_] = hist_reg.get(k_, 0) + 1
                        if len(notes) < 3:
                            d = (out.reshape(-1)[idx] - ref.reshape(-1)[idx])[:4].tolist()
                            where = "first at %s, last at %s" % (decode(idx[0], T, Ho, Wo), decode(idx[-1], T, Ho, Wo)) if vic_kind == "stem" else \
                                "flat indices %d ... %d" % (int(idx[0]), int(idx[-1]))
                            notes.append("round %d: %d words, %s, off by %s" % (r, idx.numel(), where, [round(float(v), 4) for v in d]))
                    if any(not same_bits(a, b) for a, b in zip(agg.out, agg_ref)):
                        bad_agg += 1
                        if bad_agg <= 2:
                            for a, b in zip(agg.out, agg_ref):
                                di = torch.nonzero((a.view(torch.int32) != b.view(torch.int32)).reshape(-1)).reshape(-1)
                                if di.numel():
                                    dd = (a.reshape(-1)[di] - b.reshape(-1)[di])[:4].tolist()
                                    notes.append("aggressor output, round %d: %d of %d words differ, first flat index %d, last %d, off by %s"
                                                 % (r, di.numel(), a.numel(), int(di[0]), int(di[-1]), [round(float(v), 5) for v in dd]))
                total_bad += bad + bad_agg
                print("victim %-8s %-5s x aggressor %-5s (%-9s): %3d of %d rounds with a differing victim output, %d with a differing aggressor output "
                      "[alone: victim %d of 3, aggressor %d of 3 differ; %d non-finite words]"
                      % (vic_kind, "graph" if v_cap else "eager", "graph" if a_cap else "eager", agg_kind, bad, args.rounds, bad_agg, lone_v, lone_a, nonfinite), flush=True)
                for n_ in notes:
                    print("    " + n_, flush=True)
                if bad and vic_kind == "stem":
                    print("    wrong words by lane quarter (lanes 0-15, 16-31, 32-47, 48-63): %s; by pixel of the thread (acc0, acc1): %s; registers hit (channel %% 32): %s"
                          % (hist_q, hist_pix, sorted(hist_reg.items())), flush=True)
                del vic, agg
    print("total differing rounds: %d" % total_bad)


if __name__ == "__main__":
    main()
