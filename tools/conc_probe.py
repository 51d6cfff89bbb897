#!/usr/bin/env python3
"""Concurrency probe: the same convolution launched back to back on two streams (own buffers, same data) must give bit-identical
outputs to a lone launch.  PREC selects the mode (default f16x3)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
from stemseg_amd import hip  # noqa: E402

PREC = os.environ.get("PREC", "f16x3")
REPS = int(os.environ.get("REPS", "40"))
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


def run(kind, Cin, Cout, T, H, W, cfg, sk):
    kt = {"k3": 3, "k2": 1, "k1": 1}[kind]
    k = (kt, 3, 3) if kind != "k1" else 1
    w = torch.randn((Cout, Cin, kt, 3, 3) if kind != "k1" else (Cout, Cin, 1, 1, 1), device="cuda") * 0.05
    pw = hip.pack_conv_weight_any(w, PREC)
    b = torch.randn(Cout, device="cuda")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    sets = []
    for s in range(3):
        x = torch.randn(Cin, T, H, W, device="cuda", generator=None) if s == 0 else sets[0][0].clone()
        if kind == "k1":
            xin = x.reshape(Cin, -1).contiguous()
            keep, vin = xin, hip.flat_volume(xin)
            out = torch.zeros(Cout, T * H * W, device="cuda")
            vout = hip.flat_volume(out)
        else:
            keep, vin = haloed(x, kt)
            out = torch.zeros(Cout, T, H, W, device="cuda")
            vout = hip.dense_volume(out)
        scratch = torch.zeros(8 * Cout * T * H * W, device="cuda") if sk else None
        sets.append((x, keep, vin, out, vout, scratch))
    torch.cuda.synchronize()
    x, keep, vin, out, vout, scratch = sets[2]
    hip.conv3d(vin, pw, b, vout, k, cfg, scratch, dict(precision=PREC))
    torch.cuda.synchronize()
    ref = out.clone()
    bad = 0
    for rep in range(REPS):
        for s in (0, 1):
            with torch.cuda.stream(streams[s]):
                x, keep, vin, out, vout, scratch = sets[s]
                hip.conv3d(vin, pw, b, vout, k, cfg, scratch, dict(precision=PREC))
        torch.cuda.synchronize()
        for s in (0, 1):
            if not torch.equal(sets[s][3], ref):
                bad += 1
    print("%s %4d->%4d [%d,%d,%d] cfg%d%s: %d of %d concurrent launches differ" % (kind, Cin, Cout, T, H, W, cfg, "+sk" if sk else "", bad, 2 * REPS))
    return bad


cases = [("k1", 256, 64, 8, 24, 40), ("k1", 64, 256, 8, 24, 40), ("k1", 1024, 256, 8, 6, 10), ("k1", 256, 1024, 8, 6, 10), ("k1", 2048, 512, 8, 3, 5),
         ("k2", 64, 64, 8, 24, 40), ("k2", 128, 128, 8, 12, 20), ("k2", 256, 256, 8, 6, 10), ("k2", 512, 512, 8, 3, 5), ("k2", 256, 256, 8, 24, 40),
         ("k3", 256, 256, 8, 3, 5), ("k3", 256, 256, 8, 6, 10), ("k3", 256, 128, 8, 12, 20), ("k3", 256, 128, 8, 24, 40)]
total = 0
for c in cases:
    for sk in (0, 1):
        total += run(*c, 0, sk)
print("TOTAL differing launches:", total)
