#!/usr/bin/env python3
"""One encoder convolution shape of BASELINE config 1 (32 frames per pass), a few launches, for rocprofv3 --pmc passes.
Usage: PREC=f16x3 python tools/pmc_shape.py l3conv1 | l3conv2 | l3conv3 | l1conv3 | fpn1 [reps] [tile_cfg]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import torch  # noqa: E402
from stemseg_amd import hip  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "l3conv1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
PREC = os.environ.get("PREC", "f16x3")
T = 32
SHAPES = {"l3conv1": ("k1", 1024, 256, 30, 54, False), "l3conv3": ("k1", 256, 1024, 30, 54, True), "l3conv2": ("k2", 256, 256, 30, 54, False),
          "l1conv3": ("k1", 64, 256, 120, 216, True), "l2conv3": ("k1", 128, 512, 60, 108, True), "fpn1": ("k2", 256, 256, 120, 216, False)}
k, cin, cout, h, w, residual = SHAPES[kind]
hip.require_gpu()
scratch = torch.empty(32 << 20, device="cuda")
b = torch.randn(cout, device="cuda")
if k == "k1":
    V = T * h * w
    x = torch.randn(cin, V, device="cuda")
    wt = hip.pack_conv_weight_any(torch.randn(cout, cin, 1, 1, 1, device="cuda") * 0.02, PREC)
    out = torch.empty(cout, V, device="cuda")
    epi = dict(relu=1, precision=PREC, plan=(T, 32, 32 << 20))
    if residual:
        res = torch.randn(cout, V, device="cuda")
        epi.update(residual=res, res_strides=(V, 0, 0))
    fn = lambda: hip.conv3d(hip.flat_volume(x), wt, b, hip.flat_volume(out), 1, cfg, scratch, epi)
    fl, by = 2.0 * cin * cout * V, 4.0 * V * (cin + cout * (2 if residual else 1))
else:
    pitch = hip.padded_geometry(cin, 1, h, w)["pitch"]
    buf = torch.zeros(cin, T, h + 2, pitch, device="cuda")
    buf[:, :, 1:h + 1, 1:w + 1] = torch.randn(cin, T, h, w, device="cuda")
    vin = hip.Volume(buf.data_ptr(), T * (h + 2) * pitch, (h + 2) * pitch, pitch, cin, T, h + 2, w + 2, buf.numel())
    wt = hip.pack_conv_weight_any(torch.randn(cout, cin, 1, 3, 3, device="cuda") * 0.02, PREC)
    out = torch.empty(cout, T, h, w, device="cuda")
    fn = lambda: hip.conv3d(vin, wt, b, hip.dense_volume(out), (1, 3, 3), cfg, scratch, dict(relu=1, precision=PREC, plan=(T, 32, 32 << 20)))
    fl, by = 2.0 * cin * 9 * cout * T * h * w, 4.0 * T * h * w * (cin + cout)
for _ in range(3):
    fn()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    fn()
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) / reps * 1e3
print("%s %s %d->%d on %d x [%d,%d] cfg %d: %.1f us/launch, %.1f TF-eq, %.2f TB/s of its own tensors (%.0f MB)" % (kind, PREC, cin, cout, T, h, w, cfg, us, fl / us / 1e6, by / us / 1e6, by / 1e6))
