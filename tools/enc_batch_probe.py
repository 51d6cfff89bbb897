#!/usr/bin/env python3
"""Encoder time per clip when 1, 2 or 4 clips (T = 8, 16, 32 frames) go through one encoder call."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline(dev)
bb = pipe.model._model.backbone
for n in (1, 2, 4):
    x = torch.cat([bench.make_clip(i, dev) for i in range(n)], 0)
    for _ in range(2):
        bb.forward_channel_major(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        bb.forward_channel_major(x)
    b.record()
    torch.cuda.synchronize()
    print("encoder T=%d: %.2f ms per call, %.2f ms per clip" % (8 * n, a.elapsed_time(b) / 5, a.elapsed_time(b) / 5 / n), flush=True)
