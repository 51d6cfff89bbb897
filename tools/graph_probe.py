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
mode = sys.argv[3] if len(sys.argv) > 3 else "same"      # same | alt (alternate two clips through a static input) | altmeta
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline(dev)
pipe.model.overlap_decoders = False
SEED = int(os.environ.get("PROBE_SEED", "1"))
clip = bench.make_clip(SEED, dev).clone()
clip_b = bench.make_clip(SEED + 1, dev)
clip_a = clip.clone()
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


pipe_out = None


def everything():
    global pipe_out
    pipe_out = pipe.step(clip)
    return pipe_out["labels"]


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
if mode == "same":
    for i in range(n):
        g.replay()
        torch.cuda.synchronize()
        if not torch.equal(out, want):
            bad += 1
else:
    from stemseg_amd import hip
    assert stage == "all"
    res = []
    for i in range(n):
        clip.copy_(clip_b if i % 2 else clip_a, non_blocking=True)
        g.replay()
        if mode == "altmeta":
            hip.read_cluster_meta(pipe_out["meta"])
        res.append(out.clone())
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(r, res[i % 2]) else 1 for i, r in enumerate(res))
    bad += 0 if torch.equal(res[0], want) else 100
print("graph probe %-9s: %d replays, %d mismatching" % (stage, n, bad))
