#!/usr/bin/env python3
"""Lane determinism probe: the two-lane scenario of tests/test_gpu_parity.py::test_step_batch_shares_the_encoder_pass, printing
where a second lane's graph differs from the first lane's on the same input (alone and with both in flight)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
import synth  # noqa: E402
from stemseg_amd import config  # noqa: E402
from stemseg_amd.modeling.inference_model import InferenceModel  # noqa: E402
from stemseg_amd.pipeline import ClipPipeline  # noqa: E402

config.load_preset("davis")
config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"


def make_pipe():
    model = InferenceModel()
    sd = model._model.state_dict()
    new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 29))).reshape(v.shape) for k, v in sd.items()}
    new["seediness_head.conv_out.weight"] = new["seediness_head.conv_out.weight"] * 40.0
    model._model.load_state_dict(new)
    return ClipPipeline(model, seediness_thresh=0.5)


pipe = make_pipe()
pipe1 = make_pipe() if os.environ.get("SEPARATE_MODELS") == "1" else pipe      # lane 1 on its own model instance (own packed weights, caches)
clips = [torch.as_tensor(synth.synth_frames(8, 96, 160, seed=s).astype(np.float32).transpose(0, 3, 1, 2) - 110.0).cuda() for s in (4, 5, 6)]
x, rev = torch.cat(clips, 0), torch.cat(clips[::-1], 0)
for c in clips:
    pipe.step(c)
pipe.step_batch(x, 3)
g = pipe.capture(x, n_clips=3)
g1 = pipe1.capture(x, n_clips=3, lane=1)
snap = lambda outs: [{k: v.clone() for k, v in o.items() if torch.is_tensor(v)} for o in outs]
r0 = snap(g.run(rev))
r0b = snap(g.run(rev))
r1 = snap(g1.run(rev))
torch.cuda.synchronize()
def cmp(tag, a, b):
    for i, (o, r) in enumerate(zip(a, b)):
        for k in ("emb", "bw", "seed", "labels"):
            if k in o and not torch.equal(o[k], r[k]):
                d = (o[k].float() - r[k].float()).abs()
                print("%s clip %d %s differs: max %.3e at %s, n = %d" % (tag, i, k, float(d.max()), np.unravel_index(int(d.argmax()), d.shape), int((d > 0).sum())))
                break
        else:
            print("%s clip %d identical" % (tag, i))
cmp("lane0 twice", r0b, r0)
cmp("lane1 alone vs lane0", r1, r0)
rx = snap(g.run(x))
torch.cuda.synchronize()
for rep in range(int(os.environ.get("REPS", "4"))):
    a = g.run_async(x)
    b = g1.run_async(rev)
    g.wait(); g1.wait(); torch.cuda.synchronize()
    cmp("rep %d lane0 concurrent" % rep, snap(a), rx)
    cmp("rep %d lane1 concurrent" % rep, snap(b), r0)
