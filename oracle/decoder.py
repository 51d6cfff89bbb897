"""Oracle: 3-D squeeze-expand decoders (embedding + seediness) on CPU, plain torch fp32 ops.

Follows /root/reference/stemseg/modeling/embedding_decoder.py:13-145, seediness_decoder.py:13-112,
common.py:8-35,69-78 and embedding_utils.py:4-120.  TEST INFRASTRUCTURE -- see oracle/__init__.py.
Weights are taken from a flat dict keyed by the reference's state-dict names.
"""
import numpy as np
import torch
import torch.nn.functional as F

# common.py:8-35 -- topology keyed on the clip length
_POOLS = {2: (0, 0, 0), 4: (1, 0, 0), 8: (1, 1, 0), 16: (1, 1, 1), 24: (1, 1, 1), 32: (1, 1, 1)}
_TSCALES = {2: (1, 1, 1), 4: (1, 1, 2), 8: (1, 2, 2), 16: (2, 2, 2), 24: (2, 2, 2), 32: (2, 2, 2)}

# embedding_utils.py:4-25
_NB_DIMS = {"xy": 2, "ff": 2, "xyt": 3, "xyf": 3, "xytf": 4, "xyff": 4, "xytff": 5, "xyfff": 5}
_NB_FREE = {"xyf": 1, "xytf": 1, "xyff": 2, "xytff": 2, "xyfff": 3}


def nb_embedding_dims(mode):
    return _NB_DIMS[mode]


def nb_free_dims(mode):
    return _NB_FREE.get(mode, 0)


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))


def grid_vectors(H, W, T, t_scale=1.0):
    """embedding_utils.py:28-41: linspace in fp32; extents max(1, W/H), max(1, H/W), t_scale."""
    xa = max(1.0, W / float(H))
    ya = max(1.0, H / float(W))
    x = torch.linspace(-xa, xa, W, dtype=torch.float32)
    y = torch.linspace(-ya, ya, H, dtype=torch.float32)
    t = torch.linspace(-t_scale, t_scale, T, dtype=torch.float32)
    return t, y, x


def add_offset(emb, mode, t_scale=1.0):
    """embedding_utils.py:44-120: leading channels get (t,) y, x ; free dims get zero ; 'ff' nothing."""
    C, T, H, W = emb.shape[-4:]
    t, y, x = grid_vectors(H, W, T, t_scale)
    chans = []
    if mode == "ff":
        return emb
    if "t" in mode:
        chans.append(t[:, None, None].expand(T, H, W))
    chans.append(y[None, :, None].expand(T, H, W))
    chans.append(x[None, None, :].expand(T, H, W))
    grid = torch.zeros(C, T, H, W, dtype=emb.dtype)
    for i, g in enumerate(chans):
        grid[i] = g
    return emb + grid


# model_builder.py:29-33: POOL_TYPE 'avg' | 'max', NORMALIZATION_LAYER 'gn' | 'none' (every preset uses avg / gn)
VARIANT = {"pool": "avg", "norm": "gn"}


def _block(x, sd, prefix, idx, pool):
    """conv3x3x3(+bias) -> GroupNorm(32) | Identity -> ReLU -> [AvgPool3d | MaxPool3d (3,(2,1,1),1)]  (embedding_decoder.py:20-60)"""
    x = F.conv3d(x, _t(sd[prefix + "%d.weight" % idx]), _t(sd[prefix + "%d.bias" % idx]), padding=1)
    if VARIANT["norm"] == "gn":
        x = F.group_norm(x, 32, _t(sd[prefix + "%d.weight" % (idx + 1)]), _t(sd[prefix + "%d.bias" % (idx + 1)]), 1e-5)
    x = F.relu(x)
    if pool:
        x = F.avg_pool3d(x, 3, stride=(2, 1, 1), padding=1) if VARIANT["pool"] == "avg" else F.max_pool3d(x, 3, stride=(2, 1, 1), padding=1)
    return x


def _up(x, ts):
    return F.interpolate(x, scale_factor=(ts, 2, 2), mode="trilinear", align_corners=False)


def trunk(feats, sd, prefix, T):
    """Shared trunk D1-D12.  feats: [f32, f16, f8, f4], each [1,C,T,h,w]."""
    pools, ts = _POOLS[T], _TSCALES[T]
    f32, f16, f8, f4 = feats
    x = f32
    for i in range(3):
        x = _block(x, sd, prefix + "block_32x.", 4 * i, pools[i])
    x = _up(x, ts[0])
    y = f16
    for i in range(2):
        y = _block(y, sd, prefix + "block_16x.", 4 * i, pools[i])
    x = F.conv3d(torch.cat((x, y), 1), _t(sd[prefix + "conv_16.weight"]))
    x = _up(x, ts[1])
    y = _block(f8, sd, prefix + "block_8x.", 0, pools[0])
    x = F.conv3d(torch.cat((x, y), 1), _t(sd[prefix + "conv_8.weight"]))
    x = _up(x, ts[2])
    y = _block(f4, sd, prefix + "block_4x.", 0, False)
    x = F.conv3d(torch.cat((x, y), 1), _t(sd[prefix + "conv_4.weight"]))
    return x


@torch.no_grad()
def embedding_decoder(feats, sd, mode, tanh=True, prefix="embedding_head.", t_scale=1.0):
    """Returns [E_out + Ev (+1), T, H/4, W/4]: (embeddings, raw variances, seediness?) -- the module output
    of embedding_decoder.py:101-145 for batch size 1 (bandwidth activation is NOT applied here)."""
    feats = [_t(f).float() for f in feats]
    feats = [f[None] if f.dim() == 4 else f for f in feats]
    T = feats[0].shape[2]
    x = trunk(feats, sd, prefix, T)
    emb = F.conv3d(x, _t(sd[prefix + "conv_embedding.weight"]))
    if tanh:
        emb = (emb * 0.25).tanh()
    emb = add_offset(emb[0], mode, t_scale)[None]
    var = F.conv3d(x, _t(sd[prefix + "conv_variance.weight"]), _t(sd[prefix + "conv_variance.bias"]))
    outs = [emb, var]
    if (prefix + "conv_seediness.weight") in sd:
        outs.append(F.conv3d(x, _t(sd[prefix + "conv_seediness.weight"])).sigmoid())
    return torch.cat(outs, 1)[0]


@torch.no_grad()
def seediness_decoder(feats, sd, prefix="seediness_head."):
    """seediness_decoder.py:92-112 -> [1, T, H/4, W/4]."""
    feats = [_t(f).float() for f in feats]
    feats = [f[None] if f.dim() == 4 else f for f in feats]
    x = trunk(feats, sd, prefix, feats[0].shape[2])
    return F.conv3d(x, _t(sd[prefix + "conv_out.weight"])).sigmoid()[0]


@torch.no_grad()
def semseg_decoder(feats, sd, prefix="semseg_head."):
    """semseg_decoder.py:93-116 -> raw class logits [num_classes(+1), T, H/4, W/4].  NOTE the reference's forward takes
    its list ordered 4x, 8x, 16x, 32x and reverses it (:93); ``feats`` here is ALREADY 32x, 16x, 8x, 4x (trunk order)."""
    feats = [_t(f).float() for f in feats]
    feats = [f[None] if f.dim() == 4 else f for f in feats]
    x = trunk(feats, sd, prefix, feats[0].shape[2])
    return F.conv3d(x, _t(sd[prefix + "conv_out.weight"]))[0]


@torch.no_grad()
def semseg_masks(mean_logits, output_type="probs"):
    """inference_model.py:197-231 on per-frame MEAN logits [T, C, H, W]: C > 2 -> last channel is the fg logit
    (sigmoid), the rest are class logits (logits | softmax | argmax); C == 2 -> fg = softmax[:, 1], no class map."""
    x = _t(mean_logits).float()
    if x.shape[1] > 2:
        multi, fg = x.split((x.shape[1] - 1, 1), dim=1)
        if output_type == "logits":
            mc = multi
        elif output_type == "probs":
            mc = F.softmax(multi, dim=1)
        elif output_type == "argmax":
            mc = multi.argmax(dim=1)
        else:
            mc = None
        return fg.squeeze(1).sigmoid(), mc
    return F.softmax(x, dim=1)[:, 1], None


def decoder_param_shapes(prefix, inter=(256, 256, 128, 128), cin=256, kind="embedding", mode="xyff",
                         embedding_size=4, seediness_output=False, n_classes=1):
    """(key, shape) list in the reference's state-dict naming."""
    c32, c16, c8, c4 = inter
    out = []

    def conv_gn(block, idx, ci, co):
        out.append((prefix + "%s.%d.weight" % (block, idx), (co, ci, 3, 3, 3)))
        out.append((prefix + "%s.%d.bias" % (block, idx), (co,)))
        out.append((prefix + "%s.%d.weight" % (block, idx + 1), (co,)))
        out.append((prefix + "%s.%d.bias" % (block, idx + 1), (co,)))
    conv_gn("block_32x", 0, cin, c32)
    conv_gn("block_32x", 4, c32, c32)
    conv_gn("block_32x", 8, c32, c32)
    conv_gn("block_16x", 0, cin, c16)
    conv_gn("block_16x", 4, c16, c16)
    conv_gn("block_8x", 0, cin, c8)
    conv_gn("block_4x", 0, cin, c4)
    out.append((prefix + "conv_16.weight", (c16, c32 + c16, 1, 1, 1)))
    out.append((prefix + "conv_8.weight", (c8, c16 + c8, 1, 1, 1)))
    out.append((prefix + "conv_4.weight", (c4, c8 + c4, 1, 1, 1)))
    if kind == "embedding":
        ev = embedding_size - nb_free_dims(mode)
        out.append((prefix + "conv_embedding.weight", (nb_embedding_dims(mode), c4, 1, 1, 1)))
        out.append((prefix + "conv_variance.weight", (ev, c4, 1, 1, 1)))
        out.append((prefix + "conv_variance.bias", (ev,)))
        if seediness_output:
            out.append((prefix + "conv_seediness.weight", (1, c4, 1, 1, 1)))
        out.append((prefix + "time_scale", ()))
    elif kind == "semseg":
        out.append((prefix + "conv_out.weight", (n_classes, c4, 1, 1, 1)))     # n_classes incl. the fg channel if any
    else:
        out.append((prefix + "conv_out.weight", (1, c4, 1, 1, 1)))
    return out
