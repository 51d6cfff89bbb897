#!/usr/bin/env python3
"""Runs only the dominant kernel (3x3x3 implicit-GEMM conv, block_4x shape of BASELINE config 1: Cin 256 -> Cout 128 over
[8,120,216]) a few times, for rocprofv3 --pmc passes.  Usage: python tools/pmc_conv.py [reps] [tile_cfg]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import torch  # noqa: E402
from stemseg_amd import hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Cin, Cout, T, H, W = 256, 128, 8, 120, 216
hip.require_gpu()
buf, g = hip.alloc_padded(Cin, T, H, W)
x = torch.randn(Cin, T, H, W, device="cuda")
hip.copy_to_volume(x, 0, hip.padded_interior_view(buf, g, Cin, T, H, W))
PREC = os.environ.get("PREC", "f32")
w = hip.pack_conv_weight_any(torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.02, PREC)
b = torch.randn(Cout, device="cuda")
out = torch.empty(Cout, T, H, W, device="cuda")
scratch = torch.empty(2 * out.numel(), device="cuda") if os.environ.get("PMC_NO_SCRATCH") is None else None   # enables the row-balanced launch
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    hip.conv3d(hip.padded_halo_view(buf, g, Cin, T, H, W), w, b, hip.dense_volume(out), 3, cfg, scratch, dict(precision=PREC))
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
fl = 2.0 * Cin * 27 * Cout * T * H * W
print("conv3d_k3 %dx%d [%d,%d,%d] cfg %d: %.3f ms/launch, %.1f TFLOP/s; algorithmic bytes in %.1f MB + w %.1f MB, out %.1f MB"
      % (Cin, Cout, T, H, W, cfg, ms, fl / ms / 1e9, Cin * T * H * W * 4 / 1e6, Cin * Cout * 27 * 4 / 1e6, Cout * T * H * W * 4 / 1e6))
