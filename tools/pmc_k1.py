#!/usr/bin/env python3
"""Runs the two layer3 1x1 convolutions of the encoder at the bench's shape (32 frames of 30 x 54: V = 51 840 voxels) a few
times, for rocprofv3 --pmc passes: 1024 -> 256 (+ bias, ReLU, decoded into the zero-haloed layout in the real encoder; dense
here) and 256 -> 1024 + residual + ReLU.  Usage: python tools/pmc_k1.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import torch  # noqa: E402
from stemseg_amd import hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
V = 32 * 30 * 54
hip.require_gpu()
scratch = torch.empty(32 << 20, device="cuda")
for name, cin, cout, res in (("conv1 1024->256", 1024, 256, False), ("conv3 256->1024 +res", 256, 1024, True)):
    x = torch.randn(cin, V, device="cuda")
    w = hip.pack_conv_weight(torch.randn(cout, cin, 1, 1, 1, device="cuda") * 0.02)
    b = torch.randn(cout, device="cuda")
    out = torch.empty(cout, V, device="cuda")
    epi = dict(relu=1)
    if res:
        epi.update(residual=torch.randn(cout, V, device="cuda"), res_strides=(V, 0, 0))
    for _ in range(2):
        hip.conv3d(hip.flat_volume(x), w, b, hip.flat_volume(out), 1, 0, scratch, epi)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        hip.conv3d(hip.flat_volume(x), w, b, hip.flat_volume(out), 1, 0, scratch, epi)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    print("%s over V=%d: %.1f us/launch, %.1f TFLOP/s" % (name, V, 1e3 * ms, 2.0 * cin * cout * V / ms / 1e9))
