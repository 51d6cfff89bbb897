#!/usr/bin/env python3
"""Absolute time of one clustering call (N = 207 360 points, K = 20 instances): eager launches vs hipGraph replay,
per-kernel durations from the in-process hipEvent bracket.  Usage: python tools/cluster_probe.py [N] [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from stemseg_amd import hip  # noqa: E402
from stemseg_amd.inference.clusterers import SequentialClustering  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 207360
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hip.require_gpu()
rs = np.random.RandomState(0)
lab = rs.randint(0, K + 1, N)                              # K instances + background
centres = rs.uniform(-1, 1, (K + 1, 4)).astype(np.float32) * 3
emb = (centres[lab] + rs.standard_normal((N, 4)).astype(np.float32) * 0.03).astype(np.float32)
emb[lab == K] = rs.uniform(8, 12, ((lab == K).sum(), 4))
seed = np.where(lab < K, 1.0 - np.abs(rs.standard_normal(N)) * 0.05, rs.uniform(0, 0.2, N)).astype(np.float32).clip(0, 1)
bw = (25 + rs.uniform(0, 1, (N, 2))).astype(np.float32)
e, b, s = torch.from_numpy(emb).cuda(), torch.from_numpy(bw).cuda(), torch.from_numpy(seed).cuda()
cl = SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cuda:0")


def call():
    return cl.enqueue(e, b, s, 1, None)


for _ in range(3):
    labels, meta_dev, _, _ = call()
meta = hip.read_cluster_meta(meta_dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    call()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 50 * 1e6
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    call()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = call()
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
replay = (time.perf_counter() - t0) / 200 * 1e6
a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200):
    g.replay()
bb.record()
torch.cuda.synchronize()
dev_us = a.elapsed_time(bb) / 200 * 1e3
print("cluster N=%d K=%d (found %d): eager %.1f us per call (host-launch bound), hipGraph replay %.1f us wall / %.1f us device per call; "
      "%d launches; %.1f GB/s on the 36 B/point compulsory model" % (N, K, meta.K, eager, replay, dev_us, cl.max_instances + 2, N * 36 / dev_us / 1e3))
