#!/usr/bin/env python3
"""A/B of two builds of the library on the step's convolution shapes: for every shape the launcher's own tile choice, the CRC-32 of
the output on seeded inputs (bit-identity across builds) and the time per launch.  Run once per build
(STEMSEG_HIP_LIB=<path> python tools/ab_conv.py) and diff the outputs.  PREC = f16x3 | bf16x6 | f32, SWEEP_T = frames per pass."""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import torch  # noqa: E402
from stemseg_amd import hip  # noqa: E402

hip.require_gpu()
T = int(os.environ.get("SWEEP_T", "32"))
REPS = int(os.environ.get("REPS", "30"))
PREC = os.environ.get("PREC", "f16x3")
PLAN = (T, 32, 32 << 20)
ZERO = os.environ.get("ZERO", "0") == "1"        # all-zero activations (same instruction stream, less switching power: the DVFS test)
CFG = int(os.environ.get("CFG", "0"))            # tile_cfg for every launch (0: the launcher's own choice)


def timeit(fn):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3


def crc(t):
    return zlib.crc32(t.detach().cpu().numpy().tobytes())


def report(name, fl, fn, out):
    out.zero_()
    fn()
    torch.cuda.synchronize()
    c = crc(out)
    us = timeit(fn)
    print("%-30s crc %08x  %8.1f us %6.1f TF" % (name, c, us, fl / us / 1e6), flush=True)


def k1(name, cin, cout, h, w, residual, frames=T, plan=True):
    V = frames * h * w
    g = torch.Generator(device="cuda").manual_seed(cin * 131 + cout)
    x = torch.randn(cin, V, device="cuda", generator=g)
    if ZERO:
        x.zero_()
    wt = hip.pack_conv_weight_any(torch.randn(cout, cin, 1, 1, 1, device="cuda", generator=g) * 0.02, PREC)
    b = torch.randn(cout, device="cuda", generator=g)
    out = torch.empty(cout, V, device="cuda")
    res = torch.randn(cout, V, device="cuda", generator=g) if residual else None
    scratch = torch.empty(32 << 20, device="cuda")
    epi = dict(relu=1, precision=PREC)
    if plan:
        epi["plan"] = PLAN
    if residual:
        epi.update(residual=res, res_strides=(V, 0, 0))
    report(name, 2.0 * cin * cout * V, lambda: hip.conv3d(hip.flat_volume(x), wt, b, hip.flat_volume(out), 1, CFG, scratch, epi), out)


def k2(name, cin, cout, h, w):
    g0 = hip.padded_geometry(cin, 1, h, w)
    pitch = g0["pitch"]
    g = torch.Generator(device="cuda").manual_seed(cin * 137 + cout + h)
    buf = torch.zeros(cin, T, h + 2, pitch, device="cuda")
    buf[:, :, 1:h + 1, 1:w + 1] = torch.randn(cin, T, h, w, device="cuda", generator=g)
    if ZERO:
        buf.zero_()
    vin = hip.Volume(buf.data_ptr(), T * (h + 2) * pitch, (h + 2) * pitch, pitch, cin, T, h + 2, w + 2, buf.numel())
    wt = hip.pack_conv_weight_any(torch.randn(cout, cin, 1, 3, 3, device="cuda", generator=g) * 0.02, PREC)
    b = torch.randn(cout, device="cuda", generator=g)
    out = torch.empty(cout, T, h, w, device="cuda")
    scratch = torch.empty(32 << 20, device="cuda")
    report(name, 2.0 * cin * 9 * cout * T * h * w,
           lambda: hip.conv3d(vin, wt, b, hip.dense_volume(out), (1, 3, 3), CFG, scratch, dict(relu=1, precision=PREC, plan=PLAN)), out)
    del buf


def k3(name, cin, cout, t, h, w):
    g = torch.Generator(device="cuda").manual_seed(cin * 139 + cout + h)
    buf, geo = hip.alloc_padded(cin, t, h, w)
    hip.copy_to_volume(torch.randn(cin, t, h, w, device="cuda", generator=g) * (0.0 if ZERO else 1.0), 0, hip.padded_interior_view(buf, geo, cin, t, h, w))
    wt = hip.pack_conv_weight_any(torch.randn(cout, cin, 3, 3, 3, device="cuda", generator=g) * 0.02, PREC)
    b = torch.randn(cout, device="cuda", generator=g)
    out = torch.empty(cout, t, h, w, device="cuda")
    scratch = torch.empty(64 << 20, device="cuda")
    report(name, 2.0 * cin * 27 * cout * t * h * w,
           lambda: hip.conv3d(hip.padded_halo_view(buf, geo, cin, t, h, w), wt, b, hip.dense_volume(out), 3, CFG, scratch, dict(precision=PREC)), out)


print("# %s, precision %s, %d frames per encoder pass, tile_cfg %d" % (os.path.basename(hip.LIB_PATH), PREC, T, CFG), flush=True)
ENC_ONLY = os.environ.get("ONLY", "") == "enc"
if not ENC_ONLY:
  k3("block_4x 256->128 T8", 256, 128, 8, 120, 216)
  k3("block_8x 256->128 T8", 256, 128, 8, 60, 108)
  k3("block_16x 256->256 T8", 256, 256, 8, 30, 54)
  k3("block_32x 256->256 T4", 256, 256, 4, 15, 27)
  k1("conv_16 512->256", 512, 256, 30, 54, False, frames=2, plan=False)
  k1("conv_8 384->128", 384, 128, 60, 108, False, frames=4, plan=False)
  k1("conv_4 256->128", 256, 128, 120, 216, False, frames=8, plan=False)
for st, (h, w) in enumerate(((120, 216), (60, 108), (30, 54), (15, 27))):
    mid, cout = 64 << st, 256 << st
    k1("L%d conv1 %d->%d" % (st + 1, cout, mid), cout, mid, h, w, False)
    k2("L%d conv2 %d->%d 3x3" % (st + 1, mid, mid), mid, mid, h, w)
    k1("L%d conv3 %d->%d +res" % (st + 1, mid, cout), mid, cout, h, w, True)
    k1("fpn_inner%d %d->256" % (st + 1, cout), cout, 256, h, w, False)
    k2("fpn_layer%d 256->256 3x3" % (st + 1), 256, 256, h, w)
