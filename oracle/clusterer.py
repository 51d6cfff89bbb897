"""Oracle: SequentialClustering restated with numpy bookkeeping + torch-CPU fp32 elementwise math.

Follows /root/reference/stemseg/inference/clusterers.py:52-58 (distance / prob), :60-166 (_process),
:168-175 (seed pick) -- including the three quirks documented in SURVEY.md Appendix A.2
(secondary assignment takes the MAX distance, stale availability mask after max_instances rounds,
primary masks not updated by the secondary pass).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Floating point: the three fp32 expressions (distance, probability, std) are evaluated with the very
torch CPU ops the reference uses, because numpy's float32 ``exp`` and a sequential inner-dim sum are
NOT bit-identical to torch's (measured: 39 % / 0.3 % of random inputs differ in the last ulp); all
label / index bookkeeping is numpy integer work.  Checked bit-for-bit against the reference in
tests/test_oracle_vs_golden.py.
"""
import numpy as np
import torch

F32 = np.float32


def _distance(emb, center, bw):
    """clusterers.py:56-58 : (pow(x - c, 2) * bw).sum(-1).sqrt() in fp32."""
    e, c, b = torch.from_numpy(emb), torch.from_numpy(np.ascontiguousarray(center)), torch.from_numpy(np.ascontiguousarray(bw))
    return (torch.pow(e - c, 2) * b).sum(dim=-1).sqrt().numpy()


def _prob(d):
    return (-0.5 * torch.from_numpy(np.ascontiguousarray(d))).exp().numpy()      # clusterers.py:52-54


def sequential_clustering(emb, bw, seed, label_start=1, primary=0.5, secondary=0.3, min_seediness=0.8,
                          free_dim_stds=(), max_instances=20, return_masks=False, return_probs=False):
    """emb [N,E], bw [N,Ev], seed [N] or [N,1]  ->  labels int64 [N], meta dict."""
    emb = np.ascontiguousarray(emb, F32)
    bw = np.ascontiguousarray(bw, F32)
    seed = np.ascontiguousarray(seed, F32).reshape(-1)
    N = emb.shape[0]
    meta = {"instance_labels": [], "instance_centers": [], "instance_stds": [], "instance_masks": []}
    if return_probs:
        meta["instance_probs"] = []
    if N == 0:                                                   # clusterers.py:62-69
        return np.zeros(0, np.int64), meta
    free_bw = (1. / (torch.tensor(list(free_dim_stds), dtype=torch.float32) ** 2)).numpy()      # clusterers.py:101-105
    labels = np.full(N, -1, np.int64)
    dists = []
    avail = labels == -1
    n_un = N
    for i in range(max_instances):                               # clusterers.py:106-146
        avail = labels == -1
        n_un = int(avail.sum())
        if n_un == 0:
            break
        idx = np.flatnonzero(avail)
        j = idx[int(np.argmax(seed[idx]))]                       # first maximal element (torch.argmax)
        if seed[j] < F32(min_seediness):
            break
        center = emb[j].copy()
        b = np.concatenate([bw[j], free_bw]).astype(F32)
        lab = i + label_start
        meta["instance_labels"].append(int(lab))
        meta["instance_centers"].append(center.tolist())
        meta["instance_stds"].append((1. / torch.from_numpy(b)).clamp(min=1e-8).sqrt().tolist())   # :124
        d = np.full(N, 1e8, F32)
        d[avail] = _distance(emb[avail], center, b)
        dists.append(d)
        p = np.zeros(N, F32)
        p[avail] = _prob(d[avail])
        match = (p > F32(primary)) & avail
        labels[match] = lab
        if return_masks:
            meta["instance_masks"].append(match.copy())
        if return_probs:
            meta["instance_probs"].append(p)
    if n_un > 0 and dists:                                       # clusterers.py:148-159
        D = np.stack(dists, 1)
        a = np.argmax(D, 1)                                      # QUIRK: max distance, first index on ties
        m = D[np.arange(N), a]
        upd = (_prob(m) > F32(secondary)) & avail                # `avail` may be stale by one round
        labels[upd] = a[upd] + label_start
    return labels, meta
