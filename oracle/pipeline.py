"""Oracle: clip-level glue of the hot path on CPU -- bandwidth activation, fg mask, fg gather, and the
whole embed+cluster step for one clip (used for parity checks and as bench.py's ``cpu_baseline``).

Follows /root/reference/stemseg/modeling/inference_model.py:130-162 (head calls, channel split,
``exp()*10``), inference/main.py:93-103 (fg mask = mean seediness over clips > thr) and
inference/online_chainer.py:11-22,244-289 (per-frame nonzero -> gather -> concat -> clusterer).
TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import decoder as odec
from . import encoder as oenc
from .clusterer import sequential_clustering


@torch.no_grad()
def preprocess_frames(frames_u8, min_dim, max_dim, mean=(102.9801, 115.9465, 122.7717), std=(1.0, 1.0, 1.0), unit_scale=False,
                      flip_channels=False):
    """uint8 [T,H0,W0,3] -> float32 [T,3,H,W] (inference_image_loader.py:23-43: per-frame F.interpolate bilinear to
    compute_resize_params_2's size; data/common.py:12-30: (/255), (x - mean) / std, channel flip; image_list.py:93-104:
    zero pad to multiples of 32)."""
    import math
    from .masks import compute_resize_params_2
    x = torch.as_tensor(np.asarray(frames_u8)).permute(0, 3, 1, 2).float()
    H0, W0 = x.shape[-2:]
    nw, nh, _ = compute_resize_params_2((W0, H0), min_dim, max_dim)
    x = torch.cat([F.interpolate(x[i:i + 1], (nh, nw), mode="bilinear", align_corners=False) for i in range(x.shape[0])], 0)
    if unit_scale:
        x = x / 255.
    x = (x - torch.tensor(mean, dtype=torch.float32)[None, :, None, None]) / torch.tensor(std, dtype=torch.float32)[None, :, None, None]
    if flip_channels:
        x = x.flip(dims=[1])
    H, W = int(math.ceil(nh / 32)) * 32, int(math.ceil(nw / 32)) * 32
    return F.pad(x, (0, W - nw, 0, H - nh)), (nh, nw)


def bandwidth_activation(var):
    return torch.as_tensor(var).exp() * 10.0                       # inference_model.py:148


def fg_mask_from_seediness(clips, thr=0.25):
    """clips: list of (frames, seediness[1,T',h,w]).  Mean over the clips containing each frame, > thr.
    Returns uint8 [n_frames_seen, h, w] ordered by sorted frame id (inference/main.py:93-103)."""
    acc, cnt = {}, {}
    for frames, sd in clips:
        sd = torch.as_tensor(sd)[0]
        for i, t in enumerate(frames):
            acc[t] = sd[i] + acc[t] if t in acc else 0.0 + sd[i]
            cnt[t] = cnt.get(t, 0.0) + 1.0
    return torch.stack([(acc[t] / cnt[t]) for t in sorted(acc)], 0).gt(thr).to(torch.uint8)


def gather_fg(emb, bw, seed, fg):
    """emb[E,T,h,w], bw[Ev,T,h,w], seed[1,T,h,w], fg[T,h,w] -> emb[N,E], bw[N,Ev], seed[N,1], counts[T].
    Point order = frame-major, then row-major (online_chainer.py:16-20,267-281)."""
    emb, bw, seed = (np.asarray(a, np.float32) for a in (emb, bw, seed))
    fg = np.asarray(fg).astype(bool)
    e = np.moveaxis(emb, 0, -1)[fg]
    b = np.moveaxis(bw, 0, -1)[fg]
    s = np.moveaxis(seed, 0, -1)[fg]
    return e, b, s, fg.reshape(fg.shape[0], -1).sum(1).astype(np.int64)


@torch.no_grad()
def embed_clip(frames_bgr_f32, sd, backbone_type, mode, embedding_size=4, separate_seediness=True, tanh=True):
    """frames [T,3,H,W] (mean-subtracted, padded) -> emb[E,T,h,w], bw[Ev,T,h,w], seed[1,T,h,w] (torch, CPU)."""
    feats = oenc.resnet_fpn(frames_bgr_f32, sd, backbone_type)
    stacks = [feats[s].permute(1, 0, 2, 3).contiguous() for s in (32, 16, 8, 4)]     # [C,T,h,w]
    out = odec.embedding_decoder(stacks, sd, mode, tanh)
    E = odec.nb_embedding_dims(mode)
    Ev = embedding_size - odec.nb_free_dims(mode)
    emb, var = out[:E], out[E:E + Ev]
    if separate_seediness:
        seed = odec.seediness_decoder(stacks, sd)
    else:
        seed = out[E + Ev:E + Ev + 1]
    return emb, bandwidth_activation(var), seed


def cluster_clip(emb, bw, seed, fg, label_start=1, **kw):
    e, b, s, counts = gather_fg(np.asarray(emb), np.asarray(bw), np.asarray(seed), np.asarray(fg))
    labels, meta = sequential_clustering(e, b, s, label_start=label_start, **kw)
    return labels, meta, counts


def embed_and_cluster_clip(frames, sd, backbone_type="R-101-FPN", mode="xyff", embedding_size=4,
                           separate_seediness=True, fg_thr=0.25, **cluster_kw):
    emb, bw, seed = embed_clip(frames, sd, backbone_type, mode, embedding_size, separate_seediness)
    fg = fg_mask_from_seediness([(list(range(seed.shape[1])), seed)], fg_thr)
    labels, meta, counts = cluster_clip(emb, bw, seed, fg, **cluster_kw)
    return dict(emb=emb, bw=bw, seed=seed, fg=fg, labels=labels, meta=meta, counts=counts)
