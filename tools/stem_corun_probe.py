#!/usr/bin/env python3
"""Which co-runner makes a kernel's output differ from its lone run?  A VICTIM kernel of the library runs on one HIP stream, over and
over, while an AGGRESSOR kernel of the library runs on a second stream; every victim output is compared bitwise with the victim's
lone result.  (In the three-lane soak -- tools/soak_probe.py -- every differing step started with a few wrong words in the STEM's
output: one accumulator register, lanes 48..63 of one wave.)

    python tools/stem_corun_probe.py [--reps 200] [--victims stem,heads,...] [--aggressors none,k3_f16x3,...]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
from stemseg_amd import hip  # noqa: E402

torch.manual_seed(0)


def haloed(x, kt):
    Cn, T, H, W = x.shape
    if kt == 3:
        buf, g = hip.alloc_padded(Cn, T, H, W)
        hip.copy_to_volume(x, 0, hip.padded_interior_view(buf, g, Cn, T, H, W))
        return buf, hip.padded_halo_view(buf, g, Cn, T, H, W)
    pitch = (W + 2 + 3) // 4 * 4
    buf = torch.zeros(Cn, T, H + 2, pitch, device="cuda")
    buf[:, :, 1:H + 1, 1:W + 1] = x
    return buf, hip.Volume(buf.data_ptr(), T * (H + 2) * pitch, (H + 2) * pitch, pitch, Cn, T, H + 2, W + 2, buf.numel())


def make_conv(kind, prec, Cin, Cout, T, H, W, res=False):
    """-> (run(), output tensor, keepalive)"""
    kt = 3 if kind == "k3" else 1
    if kind == "k1":
        x = torch.randn(Cin, T * H * W, device="cuda")
        w = torch.randn(Cout, Cin, 1, 1, 1, device="cuda") / Cin ** 0.5
        vin, k = hip.flat_volume(x), 1
        out = torch.zeros(Cout, T * H * W, device="cuda")
        vout = hip.flat_volume(out)
        keep = [x]
    else:
        x = torch.randn(Cin, T, H, W, device="cuda")
        w = torch.randn(Cout, Cin, kt, 3, 3, device="cuda") / (Cin * 9 * kt) ** 0.5
        buf, vin = haloed(x, kt)
        k = (kt, 3, 3)
        out = torch.zeros(Cout, T, H, W, device="cuda")
        vout = hip.dense_volume(out)
        keep = [buf]
    pw = hip.pack_conv_weight_any(w, prec)
    b = torch.randn(Cout, device="cuda")
    epi = dict(precision=prec)
    if res:
        r = torch.randn_like(out)
        epi.update(relu=1, residual=r, res_strides=(out[0].numel(), 0, 0) if kind == "k1" else (T * H * W, H * W, W))
        keep.append(r)
    keep += [pw, b]

    def run():
        hip.conv3d(vin, pw, b, vout, k, 0, None, epi)
    return run, out, keep


def make_stem(T=32, H=480, W=864):
    frames = (torch.randint(0, 256, (T, 3, H, W), device="cuda").float() - 110.0)
    w = torch.randn(64, 3, 7, 7, device="cuda") * (2.0 / 147) ** 0.5
    b = torch.randn(64, device="cuda")
    holder = {}

    def run():
        holder["out"] = hip.stem_conv(frames, w, b)
    run()
    return run, holder, [frames, w, b]


def make_heads():
    x = torch.randn(128, 8, 120, 216, device="cuda")
    w = torch.randn(6, 128, device="cuda") * 0.1
    b = torch.randn(6, device="cuda")
    holder = {}

    def run():
        holder["out"] = hip.heads(x, w, b, [0, 0, 0, 0, 3, 3], [0] * 6, None, None, None)
    run()
    return run, holder, [x, w, b]


def make_upsample():
    x = torch.randn(128, 4, 60, 108, device="cuda")
    holder = {}

    def run():
        holder["out"] = hip.upsample_trilinear(x, 2, 2, 2)
    run()
    return run, holder, [x]


VICTIMS = {
    "stem": make_stem,
    "heads": make_heads,
    "upsample": make_upsample,
    "k3_f16x3": lambda: _conv_victim("k3", "f16x3", 256, 128, 8, 120, 216),
    "k3_bf16x6": lambda: _conv_victim("k3", "bf16x6", 256, 128, 8, 120, 216),
    "k1_f16x3": lambda: _conv_victim("k1", "f16x3", 64, 256, 32, 120, 216, True),
    "k2_f16x3": lambda: _conv_victim("k2", "f16x3", 256, 256, 32, 30, 54),
}


def _conv_victim(*a):
    run, out, keep = make_conv(*a)
    holder = {"out": out}
    return run, holder, keep


AGGRESSORS = {
    "none": None,
    "k3_f16x3": lambda: make_conv("k3", "f16x3", 256, 128, 8, 120, 216)[0::2],
    "k3_bf16x6": lambda: make_conv("k3", "bf16x6", 256, 128, 8, 120, 216)[0::2],
    "k3_f32": lambda: make_conv("k3", "f32", 256, 128, 8, 120, 216)[0::2],
    "k2_f16x3": lambda: make_conv("k2", "f16x3", 256, 256, 32, 30, 54)[0::2],
    "k1_f16x3_expand": lambda: make_conv("k1", "f16x3", 64, 256, 32, 120, 216, True)[0::2],
    "k1_f16x3_reduce": lambda: make_conv("k1", "f16x3", 1024, 256, 32, 30, 54)[0::2],
    "stem": lambda: make_stem()[0::2],
    "upsample": lambda: make_upsample()[0::2],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=150)
    ap.add_argument("--victims", default="stem")
    ap.add_argument("--aggressors", default=",".join(AGGRESSORS))
    ap.add_argument("--per-rep", type=int, default=2, help="aggressor launches enqueued per victim launch")
    args = ap.parse_args()
    hip.require_gpu()
    sv, sa = torch.cuda.Stream(), torch.cuda.Stream()
    for vn in args.victims.split(","):
        vrun, vh, vkeep = VICTIMS[vn]()
        torch.cuda.synchronize()
        vrun()
        torch.cuda.synchronize()
        ref = vh["out"].clone()
        for an in args.aggressors.split(","):
            arun = None
            if AGGRESSORS[an] is not None:
                arun, akeep = AGGRESSORS[an]()
                torch.cuda.synchronize()
            bad = words = 0
            lanes = [0, 0, 0, 0]
            for rep in range(args.reps):
                if arun is not None:
                    with torch.cuda.stream(sa):
                        for _ in range(args.per_rep):
                            arun()
                with torch.cuda.stream(sv):
                    vrun()
                torch.cuda.synchronize()
                o = vh["out"]
                if not torch.equal(o.view(torch.int32), ref.view(torch.int32)):
                    bad += 1
                    idx = torch.nonzero(o.view(torch.int32).reshape(-1) != ref.view(torch.int32).reshape(-1)).flatten()
                    words += idx.numel()
                    if vn == "stem":                      # lane of the thread that produced the word (8 x 64 tile: thread = row * 32 + x % 32)
                        Wo, Ho = o.shape[-1], o.shape[-2]
                        x, y = idx % Wo, (idx // Wo) % Ho
                        lane = ((y % 8) & 1) * 32 + ((x % 64) & 31)
                        for q in range(4):
                            lanes[q] += int(((lane >> 4) == q).sum())
                    if bad <= 3:
                        d = (o.reshape(-1)[idx] - ref.reshape(-1)[idx])
                        print("    %s vs %s rep %d: %d words differ, first %s, values off by %s" % (vn, an, rep, idx.numel(), idx[:6].tolist(), [round(float(v), 4) for v in d[:6]]), flush=True)
            print("victim %-10s aggressor %-18s: %3d of %d victim launches differ from the lone run, %d words%s"
                  % (vn, an, bad, args.reps, words, ("; by lane quarter [0-15 16-31 32-47 48-63] = %s" % lanes) if vn == "stem" else ""), flush=True)


if __name__ == "__main__":
    main()
