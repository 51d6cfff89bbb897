#!/usr/bin/env python3
"""Times the encoder alone (R-101-FPN, 32 frames of 480x864: one bench step's pass): ms per pass by CUDA events, for A/B work on the encoder
kernels.  Under ``rocprofv3 --kernel-trace`` + tools/prof_steady.py it gives the per-kernel table (a pass starts with the stem kernel).
    python tools/enc_bench.py [--frames 32] [--h 480] [--w 864] [--no-fuse] [--passes 6]          (STEMSEG_HIP_LIB selects an A/B library)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--h", type=int, default=480)
ap.add_argument("--w", type=int, default=864)
ap.add_argument("--passes", type=int, default=6)
ap.add_argument("--no-fuse", action="store_true")
ap.add_argument("--fuse-tail", type=int, default=None, help="StemsegEncoderDesc.fuse_tail (7 | 8: stage 3 on the 16-column form, 7 | 16: on the one-wave-per-SIMD form)")
ap.add_argument("--backbone", default="R-101-FPN")
ap.add_argument("--precision", default="f16x3")
a = ap.parse_args()
from stemseg_amd import hip  # noqa: E402
from stemseg_amd.modeling.backbone import ResNetFPN  # noqa: E402
hip.require_gpu()
torch.manual_seed(1)
bb = ResNetFPN(a.backbone).eval()
with torch.no_grad():
    for n_, p_ in bb.named_parameters():
        if p_.dim() >= 2:
            p_.normal_(0, (2.0 / p_[0].numel()) ** 0.5)
bb = bb.cuda()
bb.fuse_tail = (not a.no_fuse) if a.fuse_tail is None else a.fuse_tail
bb.precision = a.precision
x = torch.randn(a.frames, 3, a.h, a.w, device="cuda") * 50
outs = [torch.empty(256, a.frames, a.h // s, a.w // s, device="cuda") for s in (4, 8, 16, 32)]
vols = [hip.dense_volume(o) for o in outs]
for _ in range(2):
    bb.run_backbone_into(x, vols)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.passes + 1)]
ev[0].record()
for i in range(a.passes):
    bb.run_backbone_into(x, vols)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.passes))
crc = int(outs[0].view(torch.int32).sum(dtype=torch.int64).item())
print("encoder %s %s T=%d %dx%d fuse=%s lib=%s: median %.3f ms per pass (min %.3f), bitsum %d"
      % (a.backbone, a.precision, a.frames, a.h, a.w, bb.fuse_tail, os.path.basename(hip.LIB_PATH), ms[len(ms) // 2], ms[0], crc))
