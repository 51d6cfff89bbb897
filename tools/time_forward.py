#!/usr/bin/env python3
"""Times the reference-shaped driver (TrackGenerator.do_inference = InferenceModel.forward with its frame cache, then
do_clustering) on a 64-frame DAVIS-shape sequence, next to the sharded sequence driver.  Usage: python tools/time_forward.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline(dev)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = torch.cat([bench.make_clip(5000 + i, dev) for i in range((F + 7) // 8)], 0)[:F].contiguous()
from stemseg_amd.inference.main import TrackGenerator  # noqa: E402
tg = TrackGenerator(pipe.model, "davis", seediness_thresh=0.25, frame_overlap=4)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb, fg, _ = tg.do_inference(frames)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res = tg.do_clustering(emb, fg)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d: do_inference %.1f ms, do_clustering %.1f ms, %d clips -> %.1f clips/s" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), len(emb), len(emb) / (t2 - t0)))
