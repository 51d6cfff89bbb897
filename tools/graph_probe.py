#!/usr/bin/env python3
"""Debug probe: capture ONE stage of the step into a hipGraph and replay it N times, comparing with the eager result.
Usage: python tools/graph_probe.py {encoder|decoders|cluster|all} [replays]   (one stage per process: a GPU fault kills it)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import faulthandler  # noqa: E402

faulthandler.dump_traceback_later(90, exit=True)
import torch  # noqa: E402
import bench  # noqa: E402

stage = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline(dev)
pipe.model.overlap_decoders = False
clip = bench.make_clip(1, dev)
m = pipe.model
ref = pipe.step(clip)
torch.cuda.synchronize()
emb, bw, seed = ref["emb"].clone(), ref["bw"].clone(), ref["seed"].clone()


def encoder():
    T, _, H, W = clip.shape
    pads = m._padded_feature_buffers(T, H, W, dev)
    from stemseg_amd import hip
    vols = {s: hip.padded_interior_view(buf, g, 256, T, H // s, W // s) for (buf, g), s in zip(pads, (32, 16, 8, 4))}
    m._model.backbone.run_backbone_into(clip, [vols[s] for s in (4, 8, 16, 32)])
    return pads[3][0]


def decoders():
    T, _, H, W = clip.shape
    return m._run_heads(m._padded_feature_buffers(T, H, W, dev), T, H, W, dev)[0]


def cluster():
    return pipe.cluster(emb, bw, seed)["labels"]


def everything():
    return pipe.step(clip)["labels"]


fn = {"encoder": encoder, "decoders": decoders, "cluster": cluster, "all": everything}[stage]
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
want = fn().clone()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fn()
torch.cuda.synchronize()
bad = 0
for i in range(n):
    g.replay()
    torch.cuda.synchronize()
    if not torch.equal(out, want):
        bad += 1
print("graph probe %-9s: %d replays, %d mismatching" % (stage, n, bad))
