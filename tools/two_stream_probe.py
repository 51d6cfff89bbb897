#!/usr/bin/env python3
"""Throughput with ONE captured step in flight vs TWO independent ones (own model instance, workspaces, stream): kernels of the
second fill the tail rounds / memory-bound phases of the first.  Usage: python tools/two_stream_probe.py [clips_per_step] [steps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import faulthandler  # noqa: E402
faulthandler.dump_traceback_later(150, exit=True)
import torch  # noqa: E402
import bench  # noqa: E402
from stemseg_amd import hip  # noqa: E402

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
pipes = [bench.build_pipeline(dev)[0] for _ in range(2)]
clips = [torch.cat([bench.make_clip(1000 + i * NC + c, dev) for c in range(NC)], 0) for i in range(2)]
graphs, streams = [], [torch.cuda.Stream(device=dev) for _ in range(2)]
for p in pipes:
    p.step_batch(clips[0], NC)
    torch.cuda.synchronize()
    graphs.append(p.capture(clips[0], n_clips=NC))
torch.cuda.synchronize()


def run(k):
    t0 = time.perf_counter()
    for i in range(STEPS):
        outs = []
        for j in range(k):
            with torch.cuda.stream(streams[j]):
                outs.append(graphs[j].run(clips[(i + j) % 2]))
        for j in range(k):
            with torch.cuda.stream(streams[j]):
                for o in outs[j]:
                    hip.read_cluster_meta(o["meta"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return STEPS * k * NC / dt


for k in (1, 2, 1, 2):
    print("%d graph(s) in flight: %.2f clips/s" % (k, run(k)), flush=True)
