#!/usr/bin/env python3
"""Times the encoder/decoder convolution shapes of BASELINE config 1 through the C-ABI for every tile config, with and
without split-K scratch -- the data behind the tile-selection heuristics in conv_igemm.hip.  Usage: python tools/conv_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import torch  # noqa: E402
from stemseg_amd import hip  # noqa: E402

hip.require_gpu()
T = int(os.environ.get("SWEEP_T", "8"))       # frames per encoder pass (bench default: 4 clips x 8)
REPS = int(os.environ.get("REPS", "20"))
PREC = os.environ.get("PREC", "f32")          # f32 | bf16x6 | f16x3
ONLY = os.environ.get("ONLY", "")             # "dec": decoder shapes only; "enc": encoder shapes only


def timeit(fn):
    for _ in range(8):                        # (the first configuration of a shape otherwise pays ~5 % for cold caches / clocks)
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3   # us


def k1(name, cin, cout, h, w, residual):
    V = T * h * w
    x = torch.randn(cin, V, device="cuda")
    wt = hip.pack_conv_weight_any(torch.randn(cout, cin, 1, 1, 1, device="cuda") * 0.02, PREC)
    b = torch.randn(cout, device="cuda")
    out = torch.empty(cout, V, device="cuda")
    res = torch.randn(cout, V, device="cuda") if residual else None
    scratch = torch.empty(32 << 20, device="cuda")
    epi = dict(relu=1, precision=PREC)
    if residual:
        epi.update(residual=res, res_strides=(V, 0, 0))
    fl = 2.0 * cin * cout * V
    row = []
    for cfg in ((0, 1, 2, 3) if PREC in ("bf16x6", "f16x3") and cout % 256 == 0 else (0, 1, 2)):           # 0 = the launcher's own choice
        for sc in (None, scratch):
            us = timeit(lambda: hip.conv3d(hip.flat_volume(x), wt, b, hip.flat_volume(out), 1, cfg, sc, epi))
            row.append("cfg%d%s %7.1f us %5.1f TF" % (cfg, "+sk" if sc is not None else "   ", us, fl / us / 1e6))
    print("%-28s %s" % (name, " | ".join(row)), flush=True)


def k2(name, cin, cout, h, w):
    g = hip.padded_geometry(cin, 1, h, w)
    pitch = g["pitch"]
    buf = torch.zeros(cin, T, h + 2, pitch, device="cuda")
    buf[:, :, 1:h + 1, 1:w + 1] = torch.randn(cin, T, h, w, device="cuda")
    vin = hip.Volume(buf.data_ptr(), T * (h + 2) * pitch, (h + 2) * pitch, pitch, cin, T, h + 2, w + 2, buf.numel())
    wt = hip.pack_conv_weight_any(torch.randn(cout, cin, 1, 3, 3, device="cuda") * 0.02, PREC)
    b = torch.randn(cout, device="cuda")
    out = torch.empty(cout, T, h, w, device="cuda")
    scratch = torch.empty(32 << 20, device="cuda")
    fl = 2.0 * cin * 9 * cout * T * h * w
    row = []
    for cfg in ((0, 1, 2, 3, 5) if PREC == "f16x3" and cout % 128 == 0 and pitch <= 224 else (0, 1, 2, 3)):     # 5 = the flat split-staged tile
        for sc in (None, scratch):
            us = timeit(lambda: hip.conv3d(vin, wt, b, hip.dense_volume(out), (1, 3, 3), cfg, sc, dict(relu=1, precision=PREC)))
            row.append("cfg%d%s %7.1f us %5.1f TF" % (cfg, "+sk" if sc is not None else "   ", us, fl / us / 1e6))
    print("%-28s %s" % (name, " | ".join(row)), flush=True)


def k3(name, cin, cout, t, h, w):
    buf, g = hip.alloc_padded(cin, t, h, w)
    hip.copy_to_volume(torch.randn(cin, t, h, w, device="cuda"), 0, hip.padded_interior_view(buf, g, cin, t, h, w))
    wt = hip.pack_conv_weight_any(torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.02, PREC)
    b = torch.randn(cout, device="cuda")
    out = torch.empty(cout, t, h, w, device="cuda")
    scratch = torch.empty(64 << 20, device="cuda")
    fl = 2.0 * cin * 27 * cout * t * h * w
    row = []
    for cfg in (0, 1, 2, 3):
        for sc in (None, scratch):
            us = timeit(lambda: hip.conv3d(hip.padded_halo_view(buf, g, cin, t, h, w), wt, b, hip.dense_volume(out), 3, cfg, sc, dict(precision=PREC)))
            row.append("cfg%d%s %7.1f us %5.1f TF" % (cfg, "+sk" if sc is not None else "   ", us, fl / us / 1e6))
    print("%-28s %s" % (name, " | ".join(row)), flush=True)


print("# precision %s, encoder frames per pass %d" % (PREC, T), flush=True)
if ONLY != "enc":
    k3("block_4x 256->128 T8", 256, 128, 8, 120, 216)
    k3("block_8x 256->128 T8", 256, 128, 8, 60, 108)
    k3("block_16x 256->256 T8", 256, 256, 8, 30, 54)
    k3("block_16x 256->256 T4", 256, 256, 4, 30, 54)
    k3("block_32x 256->256 T8", 256, 256, 8, 15, 27)
    k3("block_32x 256->256 T4", 256, 256, 4, 15, 27)
    k3("block_32x 256->256 T2", 256, 256, 2, 15, 27)
    _T = T
    for nm, cin, cout, t, h, w in (("conv_16 512->256", 512, 256, 2, 30, 54), ("conv_8 384->128", 384, 128, 4, 60, 108), ("conv_4 256->128", 256, 128, 8, 120, 216)):
        T = t
        k1(nm, cin, cout, h, w, False)
    T = _T
for st, (h, w) in enumerate(((120, 216), (60, 108), (30, 54), (15, 27)) if ONLY != "dec" else ()):
    mid, cout = 64 << st, 256 << st
    k1("L%d conv1 %d->%d" % (st + 1, cout, mid), cout, mid, h, w, False)
    k2("L%d conv2 %d->%d 3x3" % (st + 1, mid, mid), mid, mid, h, w)
    k1("L%d conv3 %d->%d +res" % (st + 1, mid, cout), mid, cout, h, w, True)
    k1("fpn_inner%d %d->256" % (st + 1, cout), cout, 256, h, w, False)
    k2("fpn_layer%d 256->256 3x3" % (st + 1), 256, 256, h, w)
