"""First-layer convolutions shared between decoders (stemseg_hip_shared_convs_forward): block_32x.0 / block_16x.0 / block_8x.0 /
block_4x.0 of the decoders read the same FPN maps (embedding_decoder.py:111-127, seediness_decoder.py:92-108,
semseg_decoder.py:96-112) and run as one convolution per branch with the output channels concatenated.  Every output channel is
the same dot product as in the decoder's own convolution; only the split-K partition (chosen from the launch's size) can differ,
so the results agree to fp32 rounding -- checked here against the unshared path, which the oracle tests pin.  The shared form is
opt-in (InferenceModel.share_first_convs / STEMSEG_SHARE_FIRST_CONVS=1): it measured no faster on the DAVIS pair."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def _model(preset, precision, seed=23):
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    config.load_preset(preset)
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    model = InferenceModel()
    sd = model._model.state_dict()
    model._model.load_state_dict({k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, seed))).reshape(v.shape) for k, v in sd.items()})
    model.set_precision(precision)
    model.overlap_decoders = False
    model.share_first_convs = True                               # (opt-in: STEMSEG_SHARE_FIRST_CONVS=1)
    return model.cuda()


def _frames(T, H, W, seed):
    return torch.as_tensor(synth.synth_frames(T, H, W, seed=seed).astype(np.float32).transpose(0, 3, 1, 2) - 110.0).cuda()


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "f32"])
def test_davis_pair_shared_vs_own_first_convs(precision):
    from stemseg_amd import config, hip
    hip.require_gpu()
    try:
        model = _model("davis", precision)
        x = _frames(8, 128, 192, 5)
        got = [t.clone() for t in model.embed_frames(x)]
        sf = list(model._shared.values())
        assert len(sf) == 1 and sf[0].shared_branches() == [True] * 4 and sf[0]._workspaces        # the shared path DID run
        model.share_first_convs = False
        want = model.embed_frames(x)
        for g, w, name in zip(got, want, ("emb", "bw", "seed")):
            assert torch.isfinite(g).all()
            assert _rel(g, w) < 2e-5, (name, _rel(g, w))
        # and it is deterministic: the same input again, bit for bit
        model.share_first_convs = True
        again = model.embed_frames(x)
        for g, a in zip(got, again):
            assert torch.equal(g, a)
    finally:
        config.load_preset("defaults")


@pytest.mark.parametrize("preset,branches", [("kittimots", [True] * 4), ("ytvis", [True, True, False, False])])
def test_semseg_pair_shared_vs_own_first_convs(preset, branches):
    """embedding + semseg decoders: KITTI-MOTS shares all four branches, YouTube-VIS (semseg INTER_CHANNELS 256 x 4: 8 channels per
    group on the 8x / 4x branches vs the embedding decoder's 4) only the first two."""
    from stemseg_amd import config, hip
    hip.require_gpu()
    try:
        model = _model(preset, "f16x3")
        T, H, W = 8, 128, 192
        x = _frames(T, H, W, 9)
        got = [t.clone() for t in model.embed_frames(x)]
        assert list(model._shared.values())[0].shared_branches() == branches
        assert len(model._semseg_first) == 1
        got_logits = model.semseg_logits_clip(T, H, W, x.device).clone()
        assert not model._semseg_first                                                         # consumed
        model.share_first_convs = False
        want = model.embed_frames(x)
        want_logits = model.semseg_logits_clip(T, H, W, x.device)
        for g, w, name in zip(got + [got_logits], list(want) + [want_logits], ("emb", "bw", "seed", "logits")):
            assert torch.isfinite(g).all()
            assert _rel(g, w) < 2e-5, (name, _rel(g, w))
    finally:
        config.load_preset("defaults")


def test_shared_convs_c_abi_slices_match_conv3d_gn():
    """The entry point itself: the channel / group slices it returns are each decoder's own conv (+ bias) and GroupNorm statistics."""
    from stemseg_amd import hip
    hip.require_gpu()
    torch.manual_seed(3)
    Cin, T, H4, W4 = 64, 4, 32, 48
    couts = [(64, 32), (64, 32), (32, 32), (32, 64)]             # (decoder a, decoder b) per branch
    groups = (8, 4)                                              # 8 / 8 / 4 | 8 channels per group in a; 8 / 8 / 8 | 16 in b
    feats, bufs = [], []
    for i in range(4):
        h, w = H4 >> (3 - i), W4 >> (3 - i)
        f = torch.randn(Cin, T, h, w, device="cuda")
        buf, g = hip.alloc_padded(Cin, T, h, w)
        hip.copy_to_volume(f, 0, hip.padded_interior_view(buf, g, Cin, T, h, w))
        feats.append(f)
        bufs.append((buf, g))
    d = hip.SharedConvsDesc()
    d.struct_bytes = C.sizeof(hip.SharedConvsDesc)
    d.in_channels, d.T, d.H4, d.W4, d.precision, d.gn_eps = Cin, T, H4, W4, hip.PRECISIONS["f32"], 1e-5
    ws_, bs_, packed = [], [], []
    for i in range(4):
        ca, cb = couts[i]
        shared = ca // groups[0] == cb // groups[1]
        d.cout[i] = ca + cb if shared else 0
        d.gn_groups[i] = sum(groups) if shared else 0
        w = torch.randn(ca + cb, Cin, 3, 3, 3, device="cuda") / (27 * Cin) ** 0.5
        b = torch.randn(ca + cb, device="cuda")
        ws_.append(w)
        bs_.append(b)
        packed.append(hip.pack_conv_weight_any(w, "f32") if shared else None)
    assert [d.cout[i] for i in range(4)] == [96, 96, 0, 0]
    nbytes = hip.lib().stemseg_hip_shared_convs_workspace_bytes(C.byref(d))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    arr = C.c_void_p * 4
    outp, statp = arr(), arr()
    hip.check(hip.lib().stemseg_hip_shared_convs_forward(
        C.byref(d), arr(*[(p.data_ptr() if p is not None else None) for p in packed]), arr(*[b.data_ptr() for b in bs_]),
        arr(*[b.data_ptr() for b, _ in bufs]), hip.ptr(ws), ws.numel(), outp, statp, hip.stream()))
    torch.cuda.synchronize()
    assert outp[2] is None and statp[3] is None
    base = ws.data_ptr()
    for i in range(2):
        h, w = H4 >> (3 - i), W4 >> (3 - i)
        n = d.cout[i] * T * h * w
        off = (outp[i] - base) // 4
        got = ws.view(torch.float32)[off:off + n].view(d.cout[i], T, h, w)
        want = torch.nn.functional.conv3d(feats[i][None].double(), ws_[i].double(), bs_[i].double(), padding=1)[0]
        assert _rel(got.double(), want) < 1e-5
        soff = (statp[i] - base) // 4
        st = ws.view(torch.float32)[soff:soff + 2 * d.gn_groups[i]].view(-1, 2)
        grp = want.reshape(d.gn_groups[i], -1)
        assert torch.allclose(st[:, 0].double(), grp.mean(1), atol=1e-5)
        assert torch.allclose(st[:, 1].double(), (grp.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)
    # a group count that does not divide the channels is refused
    d.cout[3], d.gn_groups[3] = 96, 64
    assert hip.lib().stemseg_hip_shared_convs_workspace_bytes(C.byref(d)) == 0
