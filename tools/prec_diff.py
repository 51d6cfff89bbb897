#!/usr/bin/env python3
"""Runs one bench-workload clip through the pipeline in two conv precisions and prints how far the outputs (and the encoder's
feature maps) are apart -- PREC_A / PREC_B environment (default bf16x6 vs f16x3), WORKLOAD as in bench.py."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

A, B = os.environ.get("PREC_A", "bf16x6"), os.environ.get("PREC_B", "f16x3")
bench.select_workload(os.environ.get("WORKLOAD", "davis"))
dev = torch.device("cuda:0")
pipe, sd = bench.build_pipeline(dev)
clip = bench.make_clip(7, dev)
outs = {}
for prec in (A, B):
    pipe.model.set_precision(prec)
    feats = None
    o = pipe.embed(clip)
    torch.cuda.synchronize()
    outs[prec] = ([t.float().cpu().numpy() for t in o], None if feats is None else [f.float().cpu().numpy() for f in feats])
for i, (a, b) in enumerate(zip(outs[A][0], outs[B][0])):
    print("output %d shape %s  max|%s| %.4g  nan/inf in %s: %d  max|diff| %.3e  (rel to max %.3e)" %
          (i, a.shape, A, np.abs(a).max(), B, int((~np.isfinite(b)).sum()), np.nanmax(np.abs(a - b)), np.nanmax(np.abs(a - b)) / max(np.abs(a).max(), 1e-30)))
if outs[A][1] is not None:
    for i, (a, b) in enumerate(zip(outs[A][1], outs[B][1])):
        print("feature %d shape %s  max %.4g  max|diff| %.3e" % (i, a.shape, np.abs(a).max(), np.nanmax(np.abs(a - b))))
