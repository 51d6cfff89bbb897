"""GPU parity tests: every HIP kernel / entry point of libstemseg_hip.so, called through the C-ABI (ctypes), vs
the CPU oracle on the same seeded inputs, the reference-generated golden fixtures, and size-independent properties
at BASELINE sizes.  Tolerances: float stages <= 1e-3 absolute (BASELINE.json north_star), most far tighter;
integer outputs exact."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import decoder as odec
from oracle import encoder as oenc
from oracle import pipeline as opipe
from oracle.clusterer import sequential_clustering
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from stemseg_amd import hip as h
    h.require_gpu()
    return h


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def report(name, got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    i = int(np.argmax(err)) if err.size else 0
    print("[parity] %-38s max|err| %.3e at %s (got %.6g ref %.6g)  max|ref| %.3g" %
          (name, err.max() if err.size else 0, np.unravel_index(i, err.shape) if err.size else (), got.flat[i] if err.size else 0,
           ref.flat[i] if err.size else 0, np.abs(ref).max() if ref.size else 0))
    return float(err.max()) if err.size else 0.0


# ------------------------------------------------------------------------------------------------ conv
def _rand(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def test_pack_conv_weight(hip):
    w = _rand((64, 12, 3, 3, 3), 0)
    got = hip.pack_conv_weight(dev(w)).cpu().numpy()
    ref = w.reshape(64, 3, 4, 27).transpose(1, 3, 2, 0).reshape(-1)          # [Cin/4][taps][4][Cout]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(8, 64, 3, 5, 37), (256, 128, 2, 9, 40), (12, 32, 1, 2, 3), (16, 160, 4, 17, 70)])
def test_conv3d_k3(hip, cfg, shape):
    Cin, Cout, T, H, W = shape
    x = _rand((Cin, T, H, W), 1)
    w = _rand((Cout, Cin, 3, 3, 3), 2, 1.0 / np.sqrt(Cin * 27))
    b = _rand((Cout,), 3)
    ref = F.conv3d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), padding=1)[0].numpy()
    buf, g = hip.alloc_padded(Cin, T, H, W)
    hip.copy_to_volume(dev(x), 0, hip.padded_interior_view(buf, g, Cin, T, H, W))
    assert np.array_equal(hip.padded_to_dense(buf, g, Cin, T, H, W).cpu().numpy(), x)
    out = torch.full((Cout, T, H, W), float("nan"), device="cuda")
    hip.conv3d(hip.padded_halo_view(buf, g, Cin, T, H, W), hip.pack_conv_weight(dev(w)), dev(b), hip.dense_volume(out), 3, cfg)
    torch.cuda.synchronize()
    assert report("conv3d_k3 cfg%d %s" % (cfg, shape), out.cpu().numpy(), ref) <= 2e-4
    # split-K path (input channels spread over up to 16 workgroups per tile, fixed-order reduction): same result, and
    # run-to-run deterministic
    scratch = torch.full((16 * Cout * T * H * W,), float("nan"), device="cuda")
    outs = []
    for _ in range(2):
        o = torch.full((Cout, T, H, W), float("nan"), device="cuda")
        hip.conv3d(hip.padded_halo_view(buf, g, Cin, T, H, W), hip.pack_conv_weight(dev(w)), dev(b), hip.dense_volume(o), 3, cfg, scratch)
        outs.append(o.cpu().numpy())
    assert report("conv3d_k3 split-K cfg%d %s" % (cfg, shape), outs[0], ref) <= 2e-4
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("kind", ["k3", "k2"])
def test_conv_kitti_shape_planner_and_tiling_stress(hip, kind):
    """BASELINE config 4 (KITTI-MOTS, max_dim 1948 -> padded 608x1952): the 4x map is 152 x 488 -- 19 row tiles x 16 column
    tiles x T, thousands of workgroups, so the launch planner's row cut (whole rows + split-K rows) is what runs.  Compared
    with torch's CPU convolution (channel counts reduced to keep the CPU side in seconds), NaN-poisoned output / scratch,
    twice for determinism."""
    T, H, W = 8, 152, 488
    Cin, Cout = (32, 128) if kind == "k3" else (64, 128)
    x = _rand((Cin, T, H, W), 5)
    if kind == "k3":
        w = _rand((Cout, Cin, 3, 3, 3), 6, 1.0 / np.sqrt(Cin * 27))
        b = _rand((Cout,), 7)
        ref = F.conv3d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), padding=1)[0].numpy()
        buf, g = hip.alloc_padded(Cin, T, H, W)
        hip.copy_to_volume(dev(x), 0, hip.padded_interior_view(buf, g, Cin, T, H, W))
        vin, k = hip.padded_halo_view(buf, g, Cin, T, H, W), 3
    else:
        w = _rand((Cout, Cin, 1, 3, 3), 6, 1.0 / np.sqrt(Cin * 9))
        b = _rand((Cout,), 7)
        ref = F.conv2d(torch.from_numpy(x).permute(1, 0, 2, 3), torch.from_numpy(w[:, :, 0]), torch.from_numpy(b), padding=1).permute(1, 0, 2, 3).numpy()
        pitch = hip.padded_geometry(Cin, 1, H, W)["pitch"]
        buf = torch.zeros(Cin, T, H + 2, pitch, device="cuda")
        buf[:, :, 1:H + 1, 1:W + 1] = dev(x)
        vin, k = hip.Volume(buf.data_ptr(), T * (H + 2) * pitch, (H + 2) * pitch, pitch, Cin, T, H + 2, W + 2, buf.numel()), (1, 3, 3)
    scratch = torch.full((2 * Cout * T * H * W,), float("nan"), device="cuda")
    outs = []
    for _ in range(2):
        o = torch.full((Cout, T, H, W), float("nan"), device="cuda")
        hip.conv3d(vin, hip.pack_conv_weight(dev(w)), dev(b), hip.dense_volume(o), k, 0, scratch)
        outs.append(o.cpu().numpy())
    assert report("conv %s KITTI 4x shape (planner)" % kind, outs[0], ref) <= 2e-4
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("shape", [(64, 128, 1000), (384, 128, 4 * 60 * 108), (512, 256, 1001), (8, 32, 5)])
def test_conv3d_k1(hip, cfg, shape):
    Cin, Cout, V = shape
    x = _rand((Cin, V), 4)
    w = _rand((Cout, Cin, 1, 1, 1), 5, 1.0 / np.sqrt(Cin))
    ref = torch.from_numpy(w.reshape(Cout, Cin)) @ torch.from_numpy(x)
    xd = dev(x)
    out = torch.full((Cout, V), float("nan"), device="cuda")
    hip.conv3d(hip.flat_volume(xd), hip.pack_conv_weight(dev(w)), None, hip.flat_volume(out), 1, cfg)
    torch.cuda.synchronize()
    assert report("conv3d_k1 cfg%d %s" % (cfg, shape), out.cpu().numpy(), ref.numpy()) <= 2e-4


@pytest.mark.parametrize("shape", [(8, 64, 2, 9, 37), (64, 64, 3, 17, 40), (256, 256, 2, 6, 27), (128, 128, 8, 30, 54)])
def test_conv2d_3x3_fused_epilogue(hip, shape):
    """(1,3,3) conv over every t-plane (= the encoder's frames) with bias + ReLU fused, zero-haloed 2-D input."""
    Cin, Cout, T, H, W = shape
    x = _rand((Cin, T, H, W), 21)
    w = _rand((Cout, Cin, 3, 3), 22, 1.0 / np.sqrt(Cin * 9))
    b = _rand((Cout,), 23)
    ref = F.relu(F.conv2d(torch.from_numpy(x).permute(1, 0, 2, 3), torch.from_numpy(w), torch.from_numpy(b), padding=1)).permute(1, 0, 2, 3).numpy()
    pitch = (W + 2 + 3) // 4 * 4
    buf = torch.zeros(Cin, T, H + 2, pitch, device="cuda")
    buf[:, :, 1:H + 1, 1:W + 1] = dev(x)
    vin = hip.Volume(buf.data_ptr(), T * (H + 2) * pitch, (H + 2) * pitch, pitch, Cin, T, H + 2, W + 2, buf.numel())
    scratch = torch.empty(16 * Cout * T * H * W, device="cuda")
    for sk in (None, scratch):
        out = torch.full((Cout, T, H, W), float("nan"), device="cuda")
        hip.conv3d(vin, hip.pack_conv_weight(dev(w)), dev(b), hip.dense_volume(out), (1, 3, 3), 0, sk, dict(relu=1))
        assert report("conv2d 3x3+relu %s splitk=%s" % (shape, sk is not None), out.cpu().numpy(), ref) <= 2e-4


@pytest.mark.parametrize("shape", [(64, 256, 2, 9, 37), (256, 64, 3, 17, 40), (1024, 256, 2, 6, 27), (64, 64, 1, 5, 8)])
def test_conv1x1_flat_decode_residual_relu(hip, shape):
    """1x1 conv on the flat [C][V] view: (a) + residual + ReLU, dense out; (b) decoded into a zero-haloed 2-D layout."""
    Cin, Cout, T, H, W = shape
    V = T * H * W
    x = _rand((Cin, V), 24)
    w = _rand((Cout, Cin, 1, 1), 25, 1.0 / np.sqrt(Cin))
    b = _rand((Cout,), 26)
    r = _rand((Cout, V), 27)
    z = torch.from_numpy(w.reshape(Cout, Cin)) @ torch.from_numpy(x) + torch.from_numpy(b)[:, None]
    xd, rd, wd, bd = dev(x), dev(r), hip.pack_conv_weight(dev(w)), dev(b)
    scratch = torch.empty(16 * Cout * V, device="cuda")
    for sk in (None, scratch):
        out = torch.full((Cout, V), float("nan"), device="cuda")
        hip.conv3d(hip.flat_volume(xd), wd, bd, hip.flat_volume(out), 1, 0, sk, dict(relu=1, residual=rd, res_strides=(V, 0, 0)))
        assert report("conv1x1+res+relu %s splitk=%s" % (shape, sk is not None), out.cpu().numpy(), F.relu(z + torch.from_numpy(r)).numpy()) <= 2e-4
        pitch = (W + 2 + 3) // 4 * 4
        buf = torch.zeros(Cout, T, H + 2, pitch, device="cuda")
        vout = hip.Volume(buf.data_ptr() + 4 * (pitch + 1), T * (H + 2) * pitch, (H + 2) * pitch, pitch, Cout, T, H, W, buf.numel() - pitch - 1)
        hip.conv3d(hip.flat_volume(xd), wd, bd, vout, 1, 0, sk, dict(relu=1, decode=(H, W)))
        got = buf[:, :, 1:H + 1, 1:W + 1].cpu().numpy()
        assert report("conv1x1 decode->haloed %s splitk=%s" % (shape, sk is not None), got, F.relu(z).reshape(Cout, T, H, W).numpy()) <= 2e-4
        halo = buf.clone()
        halo[:, :, 1:H + 1, 1:W + 1] = 0
        assert float(halo.abs().max()) == 0.0


@pytest.mark.parametrize("btype", ["R-50-FPN", "R-101-FPN"])
def test_encoder_vs_golden(hip, golden, btype):
    """HIP encoder (stem, bottlenecks with fused epilogues, FPN) vs the reference's outputs (golden) and the oracle."""
    from stemseg_amd.modeling.backbone import ResNetFPN
    g = golden("encoder")
    tag = btype.replace("-", "")
    H, W, seed, stride = g[tag + "__meta"].tolist()
    bb = ResNetFPN(btype).eval()
    sd = {k: torch.from_numpy(np.asarray(synth.synth_param("backbone." + k, v.shape, seed))).reshape(v.shape) for k, v in bb.state_dict().items()}
    bb.load_state_dict(sd)
    bb = bb.cuda()
    x = synth.synth_frames(2, H, W, seed=seed).astype(np.float32)
    x = torch.from_numpy(x).permute(0, 3, 1, 2) - torch.tensor([102.9801, 115.9465, 122.7717])[None, :, None, None]
    feats = bb.run_backbone(x.cuda())
    for s in (4, 8, 16, 32):
        ref = g["%s_s%d" % (tag, s)]
        assert list(feats[s].shape) == g["%s_s%d__shape" % (tag, s)].tolist()
        got = feats[s].contiguous().cpu().numpy().reshape(-1)[::stride]
        scale = max(1.0, float(np.abs(ref).max()))
        assert report("encoder %s 1/%d (rel to max %.3g)" % (btype, s, scale), got / scale, ref / scale) <= 1e-4


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
def test_encoder_batch_of_8_odd_size_vs_oracle(hip, precision):
    """T = 8 frames, 96 x 160 (w32 = 5: ragged 32-column tiles everywhere), R-50, vs the CPU oracle."""
    from stemseg_amd.modeling.backbone import ResNetFPN
    bb = ResNetFPN("R-50-FPN").eval()
    bb.precision = precision
    sd = synth.synth_state_dict([(k, v.shape) for k, v in bb.state_dict().items()], 51, prefix="backbone.")
    bb.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(bb.state_dict()[k].shape) for k, v in sd.items()})
    x = torch.from_numpy(synth.synth_frames(8, 96, 160, seed=51).astype(np.float32)).permute(0, 3, 1, 2) - \
        torch.tensor([102.9801, 115.9465, 122.7717])[None, :, None, None]
    ref = oenc.resnet_fpn(x, {"backbone." + k: v for k, v in sd.items()}, "R-50-FPN")
    feats = bb.cuda().run_backbone(x.cuda())
    for s in (4, 8, 16, 32):
        scale = max(1.0, float(ref[s].abs().max()))
        assert report("encoder T=8 96x160 1/%d %s" % (s, precision), feats[s].cpu().numpy() / scale, ref[s].numpy() / scale) <= 1e-4


def test_encoder_full_size_480x864_two_clips_vs_oracle(hip):
    """BASELINE config 1 frame size (padded 480x864: the 8x / 16x maps are W = 108 / 54, where the flat-tile conv runs), four
    frames as TWO clips of two through one encoder pass (n_clips = 2), R-50, vs the CPU oracle."""
    from stemseg_amd.modeling.backbone import ResNetFPN
    bb = ResNetFPN("R-50-FPN").eval()
    sd = synth.synth_state_dict([(k, v.shape) for k, v in bb.state_dict().items()], 53, prefix="backbone.")
    bb.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(bb.state_dict()[k].shape) for k, v in sd.items()})
    x = torch.from_numpy(synth.synth_frames(4, 480, 864, seed=53).astype(np.float32)).permute(0, 3, 1, 2) - \
        torch.tensor([102.9801, 115.9465, 122.7717])[None, :, None, None]
    ref = oenc.resnet_fpn(x, {"backbone." + k: v for k, v in sd.items()}, "R-50-FPN")
    bb = bb.cuda()
    outs = [[torch.full((256, 2, 480 // s, 864 // s), float("nan"), device="cuda") for s in (4, 8, 16, 32)] for _ in range(2)]
    bb.run_backbone_into(x.cuda(), [hip.dense_volume(o) for clip in outs for o in clip])
    for c in range(2):
        for o, s in zip(outs[c], (4, 8, 16, 32)):
            r = ref[s][2 * c:2 * c + 2].permute(1, 0, 2, 3).numpy()
            scale = max(1.0, float(np.abs(r).max()))
            assert report("encoder 480x864 clip %d 1/%d" % (c, s), o.cpu().numpy() / scale, r / scale) <= 1e-4


# ------------------------------------------------------------------------------------------------ GN / pool / upsample / heads
@pytest.mark.parametrize("shape", [(256, 8, 4, 7), (128, 8, 30, 54), (64, 3, 5, 5)])
def test_groupnorm_stats(hip, shape):
    x = _rand(shape, 6) * 3 + 1.5
    stats = hip.groupnorm_stats(dev(x), 32).cpu().numpy().reshape(32, 2)
    xg = x.reshape(32, -1).astype(np.float64)
    assert report("gn mean", stats[:, 0], xg.mean(1)) <= 1e-5
    assert report("gn rstd", stats[:, 1], 1 / np.sqrt(xg.var(1) + 1e-5)) <= 1e-5


@pytest.mark.parametrize("case", [(64, 128, 2, 9, 40, 0, False), (64, 256, 4, 15, 27, 0, False), (32, 128, 8, 24, 64, 0, True), (32, 256, 2, 30, 54, 0, True),
                                  (16, 128, 3, 17, 70, 1, False), (16, 128, 3, 17, 70, 2, True), (16, 256, 2, 9, 37, 3, True), (16, 64, 2, 6, 9, 0, False),
                                  (256, 128, 8, 120, 216, 0, True)])
def test_conv3d_with_groupnorm_statistics_in_the_epilogue(hip, case):
    """Conv3d + the GroupNorm(32) statistics of its output in one pass (epilogue partial sums / split-K reduce partial sums +
    fixed-order finalize) vs the conv followed by the separate statistics pass: same conv output bit for bit, statistics
    to fp32 round-off, and bit-identical from run to run.  Cases: group sizes 4 and 8 (fused) and 2 (fallback), every 3x3x3
    tile shape incl. flat tiles, split-K, and the planner's row cut on the full block_4x shape."""
    Cin, Cout, T, H, W, cfg, with_scratch = case
    x = _rand((Cin, T, H, W), 21)
    w = _rand((Cout, Cin, 3, 3, 3), 22, 1.0 / np.sqrt(Cin * 27))
    b = _rand((Cout,), 23)
    buf, g = hip.alloc_padded(Cin, T, H, W)
    hip.copy_to_volume(dev(x), 0, hip.padded_interior_view(buf, g, Cin, T, H, W))
    pw = hip.pack_conv_weight(dev(w))
    scratch = torch.empty(2 * Cout * T * H * W, device="cuda") if with_scratch else None
    vin = hip.padded_halo_view(buf, g, Cin, T, H, W)
    ref_out = torch.empty(Cout, T, H, W, device="cuda")
    hip.conv3d(vin, pw, dev(b), hip.dense_volume(ref_out), 3, cfg, scratch)
    ref_stats = hip.groupnorm_stats(ref_out, 32).cpu().numpy()
    out = torch.empty(Cout, T, H, W, device="cuda")
    stats = hip.conv3d_gn(vin, pw, dev(b), hip.dense_volume(out), 3, 32, tile_cfg=cfg, splitk_scratch=scratch)
    assert torch.equal(out, ref_out)
    got = stats.cpu().numpy()
    assert report("fused GN mean %s" % (case,), got[0::2], ref_stats[0::2]) <= 2e-6
    assert report("fused GN rstd %s (rel)" % (case,), got[1::2] / ref_stats[1::2], np.ones(32)) <= 2e-6
    again = hip.conv3d_gn(vin, pw, dev(b), hip.dense_volume(out), 3, 32, tile_cfg=cfg, splitk_scratch=scratch)
    assert torch.equal(again, stats)
    xo = ref_out.cpu().numpy().astype(np.float64).reshape(32, -1)
    assert report("fused GN mean vs fp64", got[0::2], xo.mean(1)) <= 1e-5
    assert report("fused GN rstd vs fp64", got[1::2], 1 / np.sqrt(xo.var(1) + 1e-5)) <= 1e-5


@pytest.mark.parametrize("pool", [0, 1])
@pytest.mark.parametrize("shape", [(64, 8, 6, 9), (128, 4, 15, 27), (32, 5, 3, 33), (32, 3, 4, 4)])
def test_gn_relu_pool(hip, pool, shape):
    C, T, H, W = shape
    x = _rand(shape, 7) * 2 + 0.3
    gam, bet = _rand((C,), 8) * 0.2 + 1, _rand((C,), 9) * 0.1
    ref = F.relu(F.group_norm(torch.from_numpy(x)[None], 32, torch.from_numpy(gam), torch.from_numpy(bet), 1e-5))
    if pool:
        ref = F.avg_pool3d(ref, 3, stride=(2, 1, 1), padding=1)
    ref = ref[0].numpy()
    xd = dev(x)
    stats = hip.groupnorm_stats(xd, 32)
    To = ref.shape[1]
    # (a) dense destination that is a channel slice of a larger concat buffer
    cat = torch.zeros(C + 32, To, H, W, device="cuda")
    v = hip.dense_volume(cat[32:])
    hip.gn_relu_pool(xd, 32, stats, dev(gam), dev(bet), pool, v)
    assert report("gn_relu_pool(%d) dense %s" % (pool, shape), cat[32:].cpu().numpy(), ref) <= 1e-5
    assert float(cat[:32].abs().max()) == 0.0
    # (b) zero-haloed destination: interior matches, halo untouched
    buf, g = hip.alloc_padded(C, To, H, W)
    hip.gn_relu_pool(xd, 32, stats, dev(gam), dev(bet), pool, hip.padded_interior_view(buf, g, C, To, H, W))
    assert report("gn_relu_pool(%d) haloed %s" % (pool, shape), hip.padded_to_dense(buf, g, C, To, H, W).cpu().numpy(), ref) <= 1e-5
    full = buf[:C * g["cs"]].view(C, To + 2, H + 2, g["pitch"]).clone()
    full[:, 1:To + 1, 1:H + 1, 1:W + 1] = 0
    assert float(full.abs().max()) == 0.0 and float(buf[C * g["cs"]:].abs().max()) == 0.0


@pytest.mark.parametrize("scale", [(1, 2, 2), (2, 2, 2), (1, 4, 4)])
@pytest.mark.parametrize("shape", [(16, 2, 3, 5), (7, 4, 15, 27), (3, 1, 1, 1), (5, 3, 6, 8), (4, 2, 15, 54), (3, 2, 5, 2), (2, 1, 4, 6)])
def test_upsample_trilinear(hip, scale, shape):
    x = _rand(shape, 10)
    ref = F.interpolate(torch.from_numpy(x)[None], scale_factor=tuple(float(s) for s in scale), mode="trilinear", align_corners=False)[0].numpy()
    got = hip.upsample_trilinear(dev(x), *scale).cpu().numpy()
    assert report("upsample %s %s" % (scale, shape), got, ref) <= 1e-6
    # same products and sums in the same order as ATen's CPU kernel, none fused: bit-identical (what keeps the clusterer's
    # arg-max stable after --resize_embeddings)
    assert np.array_equal(got, ref), "%d of %d values differ in the last bits" % ((got != ref).sum(), got.size)
    # into a channel slice of a wider (concat) buffer, as the decoder does: same values, neighbours untouched
    C, T, H, W = shape
    cat = torch.zeros(C + 3, T * scale[0], H * scale[1], W * scale[2], device="cuda")
    hip.upsample_trilinear(dev(x), *scale, vout=hip.dense_volume(cat[3:]))
    assert np.array_equal(cat[3:].cpu().numpy(), ref) and float(cat[:3].abs().max()) == 0.0


@pytest.mark.parametrize("layout", [0, 1])
def test_copy_to_volume(hip, layout):
    C, T, H, W = 6, 3, 4, 5
    x = _rand((C, T, H, W), 11)
    src = x if layout == 0 else np.ascontiguousarray(x.transpose(1, 0, 2, 3))
    buf, g = hip.alloc_padded(C, T, H, W)
    hip.copy_to_volume(dev(src), layout, hip.padded_interior_view(buf, g, C, T, H, W))
    assert np.array_equal(hip.padded_to_dense(buf, g, C, T, H, W).cpu().numpy(), x)


def test_heads_all_activations(hip):
    Cin, T, H, W = 128, 2, 6, 8
    x = _rand((Cin, T, H, W), 12)
    w = _rand((7, Cin), 13, 0.2)
    b = _rand((7,), 14, 0.1)
    gt, gy, gx = odec.grid_vectors(H, W, T, 1.0)
    act, axis = [1, 1, 4, 0, 3, 2, 1], [1, 2, 3, 0, 0, 0, 0]
    z = (torch.from_numpy(w) @ torch.from_numpy(x).reshape(Cin, -1) + torch.from_numpy(b)[:, None]).reshape(7, T, H, W)
    G = {1: gt[:, None, None].expand(T, H, W), 2: gy[None, :, None].expand(T, H, W), 3: gx[None, None, :].expand(T, H, W)}
    ref = torch.stack([(z[0] * 0.25).tanh() + G[1], (z[1] * 0.25).tanh() + G[2], z[2] + G[3], z[3], z[4].exp() * 10,
                       z[5].sigmoid(), (z[6] * 0.25).tanh()]).numpy()
    got = hip.heads(dev(x), dev(w), dev(b), act, axis, gt.cuda(), gy.cuda(), gx.cuda()).cpu().numpy()
    assert report("heads", got / np.maximum(1, np.abs(ref)), ref / np.maximum(1, np.abs(ref))) <= 2e-5


# ------------------------------------------------------------------------------------------------ whole decoders
def _gn(c):
    return torch.nn.GroupNorm(32, c)


def _emb_head(mode, E, tanh, seed_out, T, wseed):
    from stemseg_amd.modeling.embedding_decoder import SqueezingExpandDecoder as Emb
    m = Emb(256, [256, 256, 128, 128], E, tanh, seed_out, mode, NormType=_gn, num_frames=T)
    sd = synth.synth_state_dict([(k, v.shape) for k, v in m.state_dict().items()], wseed, prefix="embedding_head.")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(m.state_dict()[k].shape) for k, v in sd.items()})
    return m.cuda().eval()


@pytest.mark.parametrize("T", [8, 16, 4, 2, 24])
def test_embedding_decoder_vs_golden_and_oracle(hip, golden, T):
    g = golden("decoder_T%d" % T)
    for name in [k for k in g.files if k.startswith("emb_") and "__" not in k]:
        E, h32, w32, ws, tanh, so = g[name + "__meta"].tolist()
        mode = str(g[name + "__mode"])
        head = _emb_head(mode, E, bool(tanh), bool(so), T, ws)
        feats = synth.synth_features(T, h32, w32, seed=ws)
        out = head([dev(f)[None] for f in feats])[0].cpu().numpy()
        assert out.shape == g[name].shape
        assert report("emb decoder T%d %s" % (T, name), out, g[name]) <= 1e-3
        # bandwidth activation fused in the heads kernel == exp(var) * 10 of inference_model.py:148
        head.fuse_bandwidth_activation = True
        out2 = head([dev(f)[None] for f in feats])[0].cpu().numpy()
        nE = odec.nb_embedding_dims(mode)
        nV = E - odec.nb_free_dims(mode)
        ref_bw = np.exp(g[name][nE:nE + nV]) * 10
        assert report("  fused bandwidth", out2[nE:nE + nV] / ref_bw, np.ones_like(ref_bw)) <= 1e-3
        assert np.array_equal(out2[:nE], out[:nE])


@pytest.mark.parametrize("variant", [("xytff", 5, True, "avg", "gn"), ("xyfff", 5, True, "max", "gn"), ("xytff", 5, True, "max", "none"), ("xyff", 4, False, "avg", "none")])
def test_embedding_decoder_wide_heads_max_pool_no_norm_vs_oracle(hip, variant):
    """Everything model_builder.py:29-33 and embedding_utils.py:4-25 admit beyond the presets: 9-channel heads (xytff + seediness:
    5 embedding + 3 variance + 1), POOL_TYPE 'max' (nn.MaxPool3d) and NORMALIZATION_LAYER 'none' (nn.Identity), vs the oracle's
    torch-CPU composition of the same modules."""
    from stemseg_amd.modeling.embedding_decoder import SqueezingExpandDecoder as Emb
    mode, E, seed_out, pool, norm = variant
    T, h32, w32 = 8, 3, 4
    Pool = torch.nn.AvgPool3d if pool == "avg" else torch.nn.MaxPool3d
    Norm = _gn if norm == "gn" else (lambda c: torch.nn.Identity())
    m = Emb(256, [256, 256, 128, 128], E, True, seed_out, mode, PoolType=Pool, NormType=Norm, num_frames=T)
    sd = synth.synth_state_dict([(k, v.shape) for k, v in m.state_dict().items()], 77, prefix="embedding_head.")
    if norm == "none":            # no normalisation: keep the activations of seven stacked convs in range
        sd = {k: (np.asarray(v) * (0.35 if k.endswith(".weight") and np.asarray(v).ndim == 5 and np.asarray(v).shape[-1] == 3 else 1.0)).astype(np.float32) for k, v in sd.items()}
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(m.state_dict()[k].shape) for k, v in sd.items()})
    m = m.cuda().eval()
    feats = synth.synth_features(T, h32, w32, seed=78)
    out = m([dev(f)[None] for f in feats])[0].cpu().numpy()
    odec.VARIANT.update(pool=pool, norm=norm)
    try:
        ref = odec.embedding_decoder(feats, {"embedding_head." + k: v for k, v in sd.items()}, mode).numpy()
    finally:
        odec.VARIANT.update(pool="avg", norm="gn")
    n_emb = odec.nb_embedding_dims(mode)
    assert out.shape == ref.shape and out.shape[0] == n_emb + (E - odec.nb_free_dims(mode)) + int(seed_out)
    sc = np.maximum(1.0, np.abs(ref))
    assert report("emb decoder %s pool=%s norm=%s (%d channels)" % (mode, pool, norm, out.shape[0]), out / sc, ref / sc) <= 1e-3


@pytest.mark.parametrize("T", [8, 16, 4, 2, 24])
def test_seediness_decoder_vs_golden(hip, golden, T):
    from stemseg_amd.modeling.seediness_decoder import SqueezingExpandDecoder as Seed
    g = golden("decoder_T%d" % T)
    _, h32, w32, ws, _, _ = g["seediness__meta"].tolist()
    m = Seed(256, [256, 256, 128, 128], NormType=_gn, num_frames=T)
    sd = synth.synth_state_dict([(k, v.shape) for k, v in m.state_dict().items()], ws, prefix="seediness_head.")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(m.state_dict()[k].shape) for k, v in sd.items()})
    m = m.cuda().eval()
    feats = synth.synth_features(T, h32, w32, seed=ws)
    a = m([dev(f)[None] for f in feats])[0].cpu().numpy()
    b = m([dev(f)[None] for f in feats])[0].cpu().numpy()
    assert report("seediness decoder T%d" % T, a, g["seediness"]) <= 1e-3
    assert np.array_equal(a, b), "decoder is not run-to-run deterministic"


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_linear_tail_folds_vs_the_step_by_step_form(hip, precision):
    """Between the last GroupNorm + ReLU of every branch and the heads' activations the reference applies only linear maps (trilinear
    up-sampling, concatenation, the bias-free 1x1x1 convs conv_16 / conv_8 / conv_4, the 1x1x1 heads: embedding_decoder.py:64-80,112-143).
    By default the narrow heads fold that whole tail into four per-level matrices at load (SqueezeExpandTrunk._linear_tail: every level adds
    its n_out-channel share at its own resolution, fuse_w[0..2] = NULL) and the wide semseg head folds conv_4 into its weights (_fold,
    fuse_w[2] = NULL).  Against the step-by-step form (both switches off) on the same inputs: fp32 round-off -- the embedding decoder's
    embedding / variance / seediness heads, the seediness decoder's sigmoid, the 41 + 1-class head on the MFMA conv -- and the intermediate
    form (conv_4 only) likewise."""
    from stemseg_amd.modeling.seediness_decoder import SqueezingExpandDecoder as Seed
    from stemseg_amd.modeling.semseg_decoder import SqueezeExpandDecoder as Sem
    T, h32, w32 = 8, 6, 10
    feats = [dev(f)[None] for f in synth.synth_features(T, h32, w32, seed=57)]
    heads = [_emb_head("xytff", 5, True, True, T, 57)]
    for cls, args, prefix in ((Seed, (256, [256, 256, 128, 128]), "seediness_head."), (Sem, (256, 41, [256, 256, 128, 128], (4, 8, 16, 32), True), "semseg_head.")):
        m = cls(*args, NormType=_gn, num_frames=T)
        sd = synth.synth_state_dict([(k, v.shape) for k, v in m.state_dict().items()], 58, prefix=prefix)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(m.state_dict()[k].shape) for k, v in sd.items()})
        heads.append(m.cuda().eval())
    for m in heads:
        m.precision = precision
        x = feats[::-1] if isinstance(m, Sem) else feats          # (the semseg head takes 4x .. 32x)
        assert m.fold_conv4 and m.fold_linear_tail
        outs = {}
        for name, lin, c4 in (("default", True, True), ("conv_4 only", False, True), ("step by step", False, False), ("default again", True, True)):
            m.fold_linear_tail, m.fold_conv4 = lin, c4
            outs[name] = m(x)[0].cpu().numpy()
        ref = outs["step by step"]
        scale = np.maximum(1.0, np.abs(ref))
        for name in ("default", "conv_4 only"):
            err = float(np.abs(outs[name] - ref).max()), float((np.abs(outs[name] - ref) / scale).max())
            print("[fold] %s %s, %s vs step by step: max |diff| %.3g (rel to max(1, |x|) %.3g)" % (type(m).__module__.split(".")[-1], precision, name, err[0], err[1]))
            assert err[1] <= 2e-5
        assert np.array_equal(outs["default"], outs["default again"])


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
def test_decoder_full_size_480x864_vs_oracle(hip, precision):
    """BASELINE config 1 shape (T=8, padded 480x864 -> 120x216 outputs): HIP decoder vs the CPU oracle, both MFMA modes."""
    T, h32, w32 = 8, 15, 27
    head = _emb_head("xyff", 4, True, False, T, 41)
    head.precision = precision
    sd = synth.synth_state_dict(odec.decoder_param_shapes("embedding_head.", mode="xyff", embedding_size=4), 41)
    feats = synth.synth_features(T, h32, w32, seed=41)
    ref = odec.embedding_decoder(feats, sd, "xyff", True).numpy()
    out = head([dev(f)[None] for f in feats])[0].cpu().numpy()
    assert out.shape == (6, 8, 120, 216)
    assert report("decoder 480x864 %s" % precision, out, ref) <= 1e-3


def test_graphed_step_equals_eager(hip):
    """ClipPipeline.capture: the hipGraph replay of a step is bit-identical to eager launches, for the captured clip and
    for other clips (different fg counts / instance counts), including the pinned read-back of the clustering record."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    config.load_preset("davis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel()
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 29))).reshape(v.shape) for k, v in sd.items()}
        new["seediness_head.conv_out.weight"] = new["seediness_head.conv_out.weight"] * 40.0
        model._model.load_state_dict(new)
        pipe = ClipPipeline(model, seediness_thresh=0.5)
        clips = [dev(synth.synth_frames(8, 96, 128, seed=s).astype(np.float32).transpose(0, 3, 1, 2) - 110.0) for s in (1, 2, 3)]
        eager = []
        for c in clips:
            o = pipe.step(c)
            eager.append((o["labels"].clone(), o["emb"].clone(), hip.read_cluster_meta(o["meta"])))
        g = pipe.capture(clips[0])
        for rep in range(3):
            for c, (lab, emb, meta) in zip(clips, eager):
                o = g.run(c)
                m = hip.read_cluster_meta(o["meta"])
                assert (m.K, m.n_points) == (meta.K, meta.n_points)
                assert torch.equal(o["labels"], lab) and torch.equal(o["emb"], emb)
        assert len(set(int(e[2].n_points) for e in eager)) > 1, "test clips should differ in fg count"
    finally:
        config.load_preset("defaults")


def test_step_batch_shares_the_encoder_pass(hip):
    """ClipPipeline.step_batch: 3 clips through ONE encoder call (n_clips = 3 in the encoder descriptor, each clip's FPN maps
    written into its own zero-haloed buffers) == the clips one by one, BIT FOR BIT: every launch of the encoder decides its tile and
    split-K factor for the planning frame count, not for the frames that happen to share the pass."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    config.load_preset("davis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel()
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 29))).reshape(v.shape) for k, v in sd.items()}
        new["seediness_head.conv_out.weight"] = new["seediness_head.conv_out.weight"] * 40.0
        model._model.load_state_dict(new)
        pipe = ClipPipeline(model, seediness_thresh=0.5)
        clips = [dev(synth.synth_frames(8, 96, 160, seed=s).astype(np.float32).transpose(0, 3, 1, 2) - 110.0) for s in (4, 5, 6)]
        single = []
        for c in clips:
            o = pipe.step(c)
            single.append({k: o[k].clone() for k in ("emb", "bw", "seed", "labels", "fg")})
        outs = pipe.step_batch(torch.cat(clips, 0), 3)
        assert len(outs) == 3
        for i, (o, r) in enumerate(zip(outs, single)):
            for k in ("emb", "bw", "seed", "fg", "labels"):
                assert torch.equal(o[k], r[k]), "batch clip %d: %s differs from the clip stepped on its own" % (i, k)
        g = pipe.capture(torch.cat(clips, 0), n_clips=3)
        outs2 = g.run(torch.cat(clips, 0))
        for o, r in zip(outs2, outs):
            assert torch.equal(o["emb"], r["emb"]) and torch.equal(o["labels"], r["labels"])
        # two lanes: independent workspaces and streams, both steps in flight at once, results identical to one lane
        pipe1 = pipe
        if os.environ.get("STEMSEG_TEST_LANE1_PREC"):        # diagnostic: lane 1 on its own model instance in another conv precision
            model1 = InferenceModel()
            model1._model.load_state_dict(new)
            model1.set_precision(os.environ["STEMSEG_TEST_LANE1_PREC"])
            pipe1 = ClipPipeline(model1, seediness_thresh=0.5)
        g1 = pipe1.capture(torch.cat(clips, 0), n_clips=3, lane=1)
        rev = torch.cat(clips[::-1], 0)
        ref_rev = [{k: v.clone() for k, v in o.items() if torch.is_tensor(v)} for o in (g.run(rev) if pipe1 is pipe else g1.run(rev))]
        torch.cuda.synchronize()
        for rep in range(3):
            a = g.run_async(torch.cat(clips, 0))
            b = g1.run_async(rev)
            g.wait()
            g1.wait()
            torch.cuda.synchronize()
            for (name, graph, inp, got, want, lane) in (("lane 0", g, torch.cat(clips, 0), a, outs, 0), ("lane 1", g1, rev, b, ref_rev, 1)):
                same = all(torch.equal(o["emb"], r["emb"]) and torch.equal(o["labels"], r["labels"]) for o, r in zip(got, want))
                if not same and pipe1 is pipe:            # localise: the encoder's outputs (the lane's zero-haloed FPN buffers) of this replay vs a lone replay
                    conc = {k: [bf.clone() for bf, _ in v] for k, v in pipe.model._pads.items() if k[-1] == lane}
                    bb = pipe.model._model.backbone
                    ws_conc = {k: v.clone() for k, v in bb._ws.items() if k[4] == lane}
                    graph.run(inp)
                    torch.cuda.synchronize()
                    ws_ref = {k: v.clone() for k, v in bb._ws.items() if k[4] == lane}
                    other = torch.cat(clips, 0) if lane == 1 else rev          # the input this lane saw before (capture warm-up / other order)
                    graph.run(other)
                    torch.cuda.synchronize()
                    ws_other = {k: v.clone() for k, v in bb._ws.items() if k[4] == lane}
                    graph.run(inp)
                    torch.cuda.synchronize()
                    import ctypes as C
                    names = ["S0", "X1", "A", "B"] + ["Cst%d" % i for i in range(4)] + ["M1_%d" % i for i in range(4)] + ["M2", "DS", "XS"] + \
                            ["L%d" % i for i in range(4)] + ["FO%d" % i for i in range(4)] + ["SK", "total"]
                    for k, v in bb._ws.items():
                        if k[4] != lane:
                            continue
                        offs = (C.c_int64 * 25)()
                        hip.check(hip.lib().stemseg_hip_encoder_plan_offsets(C.byref(bb._desc(k[0], k[1], k[2], 3)), offs))
                        order = sorted((o, n) for o, n in zip(list(offs), names) if o >= 0)
                        a32, b32 = v.view(torch.float32), ws_conc[k].view(torch.float32)
                        for (o, n), (o2, _) in zip(order[:-1], order[1:]):
                            d = (a32[o:o2] - b32[o:o2]).abs()
                            nd = int((d > 0).sum())
                            if nd:
                                idx = torch.nonzero(d > 0).flatten()
                                c32 = ws_other[k].view(torch.float32)[o:o2]
                                stale = int(((b32[o:o2] == c32) & (d > 0)).sum())
                                print("[lanes] rep %d %s encoder buffer %-5s: %d of %d floats differ (max %.3e), first at +%d, last at +%d; %d of them equal the "
                                      "values of this lane's OTHER input order (stale data)" % (rep, name, n, nd, o2 - o, float(d.max()), int(idx[0]), int(idx[-1]), stale))
                    for k, v in sorted(pipe.model._pads.items()):
                        if k[-1] == lane:
                            for lvl, ((bf, _), c) in enumerate(zip(v, conc[k])):
                                d = (bf - c).abs()
                                print("[lanes] rep %d %s slot %d FPN level %d: %d elements differ (max %.3e)" % (rep, name, k[4], lvl, int((d > 0).sum()), float(d.max())))
                assert same, "%s: two pipelines in flight changed the result (rep %d)" % (name, rep)
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ semseg head
def _semseg_head(ncls, fg, ws, inter=(128, 128, 64, 64)):
    from stemseg_amd.modeling.semseg_decoder import SqueezeExpandDecoder as Sem
    m = Sem(256, ncls, list(inter), (4, 8, 16, 32), foreground_channel=bool(fg), NormType=_gn, num_frames=8)
    sd = synth.synth_state_dict([(k, v.shape) for k, v in m.state_dict().items()], ws, prefix="semseg_head.")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(m.state_dict()[k].shape) for k, v in sd.items()})
    return m.cuda().eval()


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("name", ["sem_bin", "sem_kitti", "sem_ytvis"])
def test_semseg_decoder_vs_golden(hip, golden, name, precision):
    """2 / 3+1 channels go through the fused heads kernel, 40+1 through the 1x1x1 MFMA conv (zero-padded to 64 rows)."""
    g = golden("semseg")
    ncls, fg, h32, w32, ws = g[name + "__meta"].tolist()
    m = _semseg_head(ncls, fg, ws)
    m.precision = precision
    feats = synth.synth_features(8, h32, w32, seed=ws)
    out = m([dev(f)[None] for f in feats[::-1]])[0].cpu().numpy()          # reference order: 4x, 8x, 16x, 32x
    assert out.shape == tuple(g[name + "__shape"].tolist())
    stride = 1 if g[name].ndim == 4 else 3
    assert report("semseg %s %s" % (name, precision), out.reshape(-1)[::stride], g[name].reshape(-1)) <= 1e-3


def test_semseg_accumulate_and_masks_vs_golden(hip, golden):
    """accumulate (incl. a clip that repeats a frame) is bit-exact vs sequential numpy fp32 adds; the mask kernel on the
    GOLDEN logits: fg / logits / probs <= 1e-6, argmax identical."""
    g = golden("semseg")
    rs = np.random.RandomState(5)
    Cn, T, H, W, Fn = 5, 4, 6, 10, 6
    clips = [(rs.randn(Cn, T, H, W).astype(np.float32), sub) for sub in ([0, 1, 2, 3], [2, 3, 4, 5], [5, 5, 5, 1])]
    acc = torch.zeros(Fn, Cn, H, W, device="cuda")
    ref, cnt = np.zeros((Fn, Cn, H, W), np.float32), [0] * Fn
    from stemseg_amd.modeling.inference_model import InferenceModel
    for x, sub in clips:
        InferenceModel._accumulate_semseg(acc, cnt, dev(x), sub)
        for i, t in enumerate(sub):
            ref[t] = ref[t] + x[:, i]
    assert cnt == [1, 2, 2, 2, 1, 4] and np.array_equal(acc.cpu().numpy(), ref)
    with pytest.raises(RuntimeError):
        hip.semseg_accumulate(acc, dev(clips[0][0]), [1, 1, 2, 3])          # duplicates in one launch are refused
    for name, kinds in (("sem_kitti", ("logits", "probs", "argmax", None)), ("sem_ytvis", ("argmax",))):
        ncls, fg, h32, w32, ws = g[name + "__meta"].tolist()
        if name == "sem_kitti":
            y = g[name]
        else:                                                                # wide case: the golden keeps every 3rd logit only
            sd = synth.synth_state_dict(odec.decoder_param_shapes("semseg_head.", kind="semseg", n_classes=ncls + fg, inter=(128, 128, 64, 64)), ws)
            y = odec.semseg_decoder(synth.synth_features(8, h32, w32, seed=ws), sd).numpy()
        T2 = y.shape[1]
        mult = np.array([1 + t % 3 for t in range(T2)], np.float32)
        sums = np.ascontiguousarray((y * mult[None, :, None, None]).transpose(1, 0, 2, 3))     # [F, C, h, w]
        for kind in kinds:
            fgm, mc = hip.semseg_masks(dev(sums), dev(mult), kind)
            key = "%s_fg_%s" % (name, kind or "logits")
            assert report("semseg fg %s %s" % (name, kind), fgm.cpu().numpy(), g[key]) <= 1e-6
            if kind is None:
                assert mc is None
            elif kind == "argmax":
                srt = np.sort((sums / mult[:, None, None, None])[:, :-1], 1)
                safe = (srt[:, -1] - srt[:, -2]) > (0 if name == "sem_kitti" else 1e-4)
                assert np.array_equal(mc.cpu().numpy()[safe], g["%s_mc_argmax" % name][safe]) and safe.mean() > 0.99
            else:
                assert report("semseg mc %s %s" % (name, kind), mc.cpu().numpy(), g["%s_mc_%s" % (name, kind)]) <= 1e-6
    # 2-channel head: fg = softmax[:, 1]
    x = rs.randn(3, 2, 5, 7).astype(np.float32)
    fgm, mc = hip.semseg_masks(dev(x), dev(np.ones(3, np.float32)), "probs")
    assert mc is None and report("semseg fg binary", fgm.cpu().numpy(), torch.softmax(torch.from_numpy(x), 1)[:, 1].numpy()) <= 1e-6


def test_inference_model_with_semseg_head(hip):
    """kitti-style model (xyt embeddings with in-head seediness + 3+1-channel semseg head) over overlapping clips incl. a
    repeated-frame clip: per-clip outputs and the averaged semseg masks vs the oracle run on the same frames."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel, preprocess_frames
    from oracle import encoder as oenc
    config.load_preset("kittimots")
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = 96, 128
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel(semseg_output_type="argmax")
        sd = model._model.state_dict()
        new = {k: synth.synth_param(k, v.shape, 23) for k, v in sd.items()}
        model._model.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(sd[k].shape) for k, v in new.items()})
        model = model.cuda()
        assert model.has_semseg_head
        frames = synth.synth_frames(10, 96, 128, seed=23)
        subs = [list(range(0, 8)), list(range(2, 10)), [9, 9, 9, 9, 9, 9, 9, 3]]
        res = model([f for f in frames], subs)
        x, _ = preprocess_frames(frames)
        new = {k: np.asarray(v).reshape(tuple(sd[k].shape)) for k, v in new.items()}
        feats = {s: f.numpy() for s, f in oenc.resnet_fpn(x.cpu().numpy(), new, "R-50-FPN").items()}
        acc, cnt = None, np.zeros(10)
        for sub in subs:
            f = [np.stack([feats[s][t] for t in sub], 1) for s in (32, 16, 8, 4)]
            y = odec.semseg_decoder(f, new).numpy()
            acc = np.zeros((10,) + y.shape[:1] + y.shape[2:], np.float32) if acc is None else acc
            for i, t in enumerate(sub):
                acc[t] += y[:, i]
                cnt[t] += 1
        mean = acc / cnt[:, None, None, None].astype(np.float32)
        fg_ref, mc_ref = odec.semseg_masks(mean, "argmax")
        assert report("semseg model fg", res["fg_masks"].cpu().numpy(), fg_ref.numpy()) <= 1e-3
        srt = np.sort(mean[:, :-1], 1)
        safe = (srt[:, -1] - srt[:, -2]) > 1e-3
        assert np.array_equal(res["multiclass_masks"].cpu().numpy()[safe], mc_ref.numpy()[safe])
        assert len(res["embeddings"]) == 3 and res["embeddings"][2].subseq_frames == [3, 9]
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ pre-processing (8f #4)
def test_preprocess_frames_vs_golden_and_oracle(hip, golden):
    from oracle import pipeline as opipe
    from stemseg_amd import config
    from stemseg_amd.modeling import inference_model as im
    g = golden("misc")
    try:
        config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = 64, 96
        out, hw = im.preprocess_frames(g["preproc__in"][None])
        assert tuple(out.shape[1:]) == g["preproc__out"].shape
        assert report("preprocess (golden, 50x70 -> 64x96)", out[0].cpu().numpy(), g["preproc__out"]) <= 1e-4
        # 720p -> DAVIS 480p network input (down-scaling), 360p -> YT-VIS (up-scaling), unit scale + RGB flip variant
        for (h0, w0), (mn, mx), unit, flip in (((720, 1280), (480, 854), False, False), ((360, 640), (640, 1196), False, False),
                                              ((100, 75), (96, 160), True, True)):
            config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = mn, mx
            config.cfg.INPUT.NORMALIZE_TO_UNIT_SCALE, config.cfg.INPUT.BGR_INPUT = unit, not flip
            config.cfg.INPUT.IMAGE_STD = [0.229, 0.224, 0.225] if unit else [1.0, 1.0, 1.0]
            config.cfg.INPUT.IMAGE_MEAN = [0.485, 0.456, 0.406] if unit else [102.9801, 115.9465, 122.7717]
            fr = synth.synth_frames(3, h0, w0, seed=h0)
            out, hw = im.preprocess_frames(fr)
            ref, hw_ref = opipe.preprocess_frames(fr, mn, mx, config.cfg.INPUT.IMAGE_MEAN, config.cfg.INPUT.IMAGE_STD, unit, flip)
            assert hw == hw_ref and tuple(out.shape) == tuple(ref.shape)
            assert report("preprocess %dx%d -> %s" % (h0, w0, tuple(ref.shape[-2:])), out.cpu().numpy(), ref.numpy()) <= 1e-4
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ mask materialisation (8f #3)
def test_mask_materialisation_vs_golden(hip, golden):
    """MaskMaterializer vs the PNGs written by the reference's DavisOutputGenerator: identical except (at most) pixels whose
    soft value sits within 1e-6 of the 0.5 threshold."""
    from oracle import masks as omask
    from stemseg_amd import config
    from stemseg_amd.inference.output_utils import MaskMaterializer, instances_to_keep
    g = golden("masks")
    try:
        for name in g["__names"].tolist():
            h, w, ih, iw, mn, mx, nf, max_tracks = g[name + "__dims"].tolist()
            config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = mn, mx
            maps = g[name + "__maps"]
            life = dict(zip(g[name + "__lifetime_keys"].tolist(), g[name + "__lifetime_vals"].tolist()))
            idxes, labels = [], []
            for t in range(nf):
                ys, xs = np.nonzero(maps[t])
                idxes.append((dev(ys), dev(xs)))
                labels.append(dev(maps[t][ys, xs]))
            keep, out = MaskMaterializer(-1).process_sequence((ih, iw), idxes, labels, life, (h, w), 4.0, max_tracks)
            assert keep == g[name + "__keep"].tolist() == instances_to_keep(life, -1, max_tracks)
            out, ref = out.cpu().numpy(), g[name + "__condensed"]
            bad = out != ref
            for t in np.unique(np.nonzero(bad)[0]).tolist():
                soft = omask.soft_masks(maps[t], keep, (ih, iw), mn, mx).numpy()
                assert (np.abs(soft - 0.5).min(0)[bad[t]] < 1e-6).all(), "mask mismatch away from the threshold (%s frame %d)" % (name, t)
            print("[parity] masks %-6s %d frames %dx%d -> %dx%d: %d / %d pixels differ (threshold ties)" % (name, nf, h, w, ih, iw, bad.sum(), bad.size))
            assert bad.mean() < 1e-3
    finally:
        config.load_preset("defaults")


def test_mask_materialisation_full_size_vs_oracle(hip):
    """DAVIS 480p: 120x216 label maps -> 480x854 masks (identity resize after the crop) and a 720p original (up-scaling)."""
    from oracle import masks as omask
    from stemseg_amd import config
    from stemseg_amd.inference.output_utils import MaskMaterializer
    rs = np.random.RandomState(11)
    try:
        for (ih, iw), (mn, mx) in (((480, 854), (480, 854)), ((720, 1280), (480, 854))):
            config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = mn, mx
            maps = np.zeros((2, 120, 216), np.int64)
            for k in range(1, 13):
                y, x, hh, ww = rs.randint(0, 100), rs.randint(0, 190), rs.randint(4, 30), rs.randint(4, 40)
                maps[:, y:y + hh, x:x + ww] = k
            maps[1] = np.roll(maps[1], 3, axis=1)
            life = {k: int(rs.randint(0, 5)) for k in range(1, 13)}
            idxes = [(dev(np.nonzero(m)[0]), dev(np.nonzero(m)[1])) for m in maps]
            labels = [dev(m[np.nonzero(m)]) for m in maps]
            keep, out = MaskMaterializer(-1).process_sequence((ih, iw), idxes, labels, life, (120, 216), 4.0, 10)
            ref = omask.condensed_masks(maps, keep, (ih, iw), mn, mx).numpy()
            bad = out.cpu().numpy() != ref
            print("[parity] masks full size -> %dx%d: %d / %d pixels differ" % (ih, iw, bad.sum(), bad.size))
            for t in np.unique(np.nonzero(bad)[0]).tolist():
                soft = omask.soft_masks(maps[t], keep, (ih, iw), mn, mx).numpy()
                assert (np.abs(soft - 0.5).min(0)[bad[t]] < 1e-6).all()
            assert bad.mean() < 1e-3
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ fg mask / gather
def test_fg_mask_accumulate(hip):
    rs = np.random.RandomState(3)
    planes = [rs.uniform(0, 1, (13, 17)).astype(np.float32) for _ in range(3)]
    acc = torch.empty(13, 17, device="cuda")
    for i, p in enumerate(planes):
        hip.seediness_accumulate(acc, dev(p), i == 0)
    m = hip.fg_mask(acc, 3.0, 0.5).cpu().numpy()
    ref = ((0.0 + torch.from_numpy(planes[0]) + torch.from_numpy(planes[1]) + torch.from_numpy(planes[2])) / 3.0 > 0.5).numpy()
    assert np.array_equal(m.astype(bool), ref)


@pytest.mark.parametrize("case", [(4, 20, 28, 5), (3, 7, 5, 0), (8, 33, 65, 12)])
def test_fg_gather_vs_oracle(hip, case):
    T, H, W, K = case
    emb, bw, sd, fg = synth.synth_cluster_case(T, H, W, K, seed=K)
    e, b, s, counts = opipe.gather_fg(emb, bw, sd, fg)
    ge, gb, gs, vox, offs = hip.fg_gather(dev(emb), dev(bw), dev(sd), dev(fg))
    offs = offs.cpu().numpy()
    n = int(offs[-1])
    assert n == e.shape[0] and np.array_equal(np.diff(offs), counts)
    assert np.array_equal(ge[:n].cpu().numpy(), e) and np.array_equal(gb[:n].cpu().numpy(), b) and np.array_equal(gs[:n].cpu().numpy(), s[:, 0])
    assert np.array_equal(vox[:n].cpu().numpy(), np.flatnonzero(fg.reshape(-1)))
    # all-foreground and all-background masks
    for fill in (0, 1):
        f2 = np.full_like(fg, fill)
        _, _, _, vox2, offs2 = hip.fg_gather(dev(emb), dev(bw), dev(sd), dev(f2))
        assert offs2.cpu().tolist() == [fill * H * W * t for t in range(T + 1)]


# ------------------------------------------------------------------------------------------------ clustering
def _hip_cluster(case_emb, case_bw, case_seed, params, want=True):
    from stemseg_amd.inference.clusterers import SequentialClustering
    nfree = int(params[3])
    cl = SequentialClustering(0.5, 0.3, float(params[0]), nfree, [float(v) for v in params[4:4 + nfree]], "cuda:0",
                              max_instances=int(params[1]))
    labels, meta = cl(dev(case_emb), bandwidths=dev(case_bw), seediness=dev(case_seed), cluster_label_start=int(params[2]),
                      return_label_masks=want, return_probs=want)
    return labels.cpu().numpy(), meta


def _oracle_cluster(c, n, probs=False):
    p = c[n + "__params"]
    nfree = int(p[3])
    return sequential_clustering(c[n + "__emb"], c[n + "__bw"], c[n + "__seed"], label_start=int(p[2]), min_seediness=p[0],
                                 max_instances=int(p[1]), free_dim_stds=p[4:4 + nfree], return_masks=True, return_probs=probs)


def test_cluster_vs_golden_exact(hip, golden):
    c = golden("cluster")
    for n in [str(x) for x in c["__names"] if str(x) != "adversarial_cloud"]:
        labels, meta = _hip_cluster(c[n + "__emb"], c[n + "__bw"], c[n + "__seed"], c[n + "__params"])
        E = c[n + "__emb"].shape[1]
        assert labels.dtype == np.int64
        assert meta["instance_labels"] == c[n + "__instance_labels"].tolist(), n
        bad = np.flatnonzero(labels != c[n + "__labels"])
        assert bad.size == 0, "%s: %d label mismatches, first at %s" % (n, bad.size, bad[:5])
        assert np.array_equal(np.array(meta["instance_centers"], np.float32).reshape(-1, E), c[n + "__centers"]), n
        assert report("stds " + n, np.array(meta["instance_stds"], np.float32).reshape(-1, E), c[n + "__stds"]) <= 1e-6
        masks = np.stack([m.numpy() for m in meta["instance_masks"]]) if meta["instance_masks"] else np.zeros((0, labels.shape[0]), bool)
        assert np.array_equal(masks, c[n + "__masks"]), n


def test_cluster_adversarial_band(hip, golden):
    """Points within a few ulp of a threshold may legitimately differ (CPU libm vs GPU expf/sqrt);
    everything else -- and the instance list -- must match exactly, and probabilities agree to 1e-6."""
    c = golden("cluster")
    n = "adversarial_cloud"
    labels, meta = _hip_cluster(c[n + "__emb"], c[n + "__bw"], c[n + "__seed"], c[n + "__params"])
    ref_labels, ref_meta = _oracle_cluster(c, n, probs=True)
    assert meta["instance_labels"] == ref_meta["instance_labels"]
    P = np.stack(ref_meta["instance_probs"])
    G = np.stack([p.numpy() for p in meta["instance_probs"]])
    assert report("cluster probs", G, P) <= 1e-6
    near = (np.abs(P - 0.5) < 2e-6).any(0) | (np.abs(P - 0.3) < 2e-6).any(0)
    bad = np.flatnonzero(labels != ref_labels)
    print("[parity] adversarial cloud: %d mismatches, %d points in the threshold band" % (bad.size, int(near.sum())))
    assert np.all(near[bad])


@pytest.mark.parametrize("K", [0, 1, 10, 25])
def test_cluster_full_size_vs_oracle(hip, K):
    """N ~ 2e5 points (T=8, 120x216, BASELINE config 1 shape): exact labels on margin data, deterministic."""
    emb, bw, sd, fg = synth.synth_cluster_case(8, 120, 216, K, seed=100 + K, bg_fraction=0.5)
    e, b, s, _ = opipe.gather_fg(emb, bw, sd, fg)
    ref, ref_meta = sequential_clustering(e, b, s, label_start=3, free_dim_stds=[0.3, 0.3])
    params = np.array([0.8, 20, 3, 2, 0.3, 0.3])
    l1, m1 = _hip_cluster(e, b, s[:, None], params, want=False)
    l2, _ = _hip_cluster(e, b, s[:, None], params, want=False)
    assert m1["instance_labels"] == ref_meta["instance_labels"]
    assert np.array_equal(l1, l2)
    bad = np.flatnonzero(l1 != ref)
    assert bad.size == 0, "K=%d N=%d: %d mismatches" % (K, e.shape[0], bad.size)


def test_cluster_ytvis_full_resolution_1M_points(hip):
    """BASELINE config 2 (YouTube-VIS 360p -> padded 384x640, --resize_embeddings): the head outputs are up-sampled x4 and
    clustered at full resolution: ~1e6 foreground points in one clip (5x the DAVIS case).  Labels exact vs the oracle, deterministic."""
    emb, bw, sd, fg = synth.synth_cluster_case(8, 384, 640, 12, seed=77, bg_fraction=0.12)
    e, b, s, _ = opipe.gather_fg(emb, bw, sd, fg)
    assert e.shape[0] > 900_000
    ref, ref_meta = sequential_clustering(e, b, s, label_start=1, free_dim_stds=[0.3, 0.3])
    params = np.array([0.8, 20, 1, 2, 0.3, 0.3])
    l1, m1 = _hip_cluster(e, b, s[:, None], params, want=False)
    l2, _ = _hip_cluster(e, b, s[:, None], params, want=False)
    assert m1["instance_labels"] == ref_meta["instance_labels"] and len(m1["instance_labels"]) >= 12
    assert np.array_equal(l1, l2)
    assert np.array_equal(l1, ref), "%d mismatches of %d" % ((l1 != ref).sum(), l1.size)


def test_overlap_counts_and_relabel(hip):
    rs = np.random.RandomState(5)
    la = rs.randint(-1, 6, 5000).astype(np.int64)
    lb = np.where(rs.uniform(size=5000) < 0.2, -1, rs.randint(6, 10, 5000)).astype(np.int64)
    ids_a, ids_b = [1, 2, 3, 4, 5], [6, 7, 8, 9]
    from stemseg_amd.inference.online_chainer import HipChainerOps
    ops = HipChainerOps()
    inter, ca, cb = ops.overlap_counts(dev(la), dev(lb), ids_a, ids_b)
    assert np.array_equal(inter, np.array([[np.sum((la == a) & (lb == b)) for b in ids_b] for a in ids_a]))
    assert np.array_equal(ca, [np.sum(la == a) for a in ids_a]) and np.array_equal(cb, [np.sum(lb == b) for b in ids_b])
    t = dev(lb.copy())
    ops.relabel(t, {6: 2, 8: 1})
    ref = lb.copy()
    ref[lb == 6] = 2
    ref[lb == 8] = 1
    assert np.array_equal(t.cpu().numpy(), ref)


def test_overlap_counts_ids_beyond_500_and_large_tables(hip):
    """Label VALUES may be arbitrarily large (track ids only grow, online_chainer.py:43-49) and the id sets arbitrarily big:
    the LDS table serves K1 x K2 up to 12k cells, global atomics beyond; both exact vs numpy."""
    from stemseg_amd.inference.online_chainer import HipChainerOps
    from tests.oracle_ops import OracleChainerOps
    ops, ref = HipChainerOps(), OracleChainerOps()
    rs = np.random.RandomState(9)
    for ids_a, ids_b, n in (([3, 501, 777, 1203], [1204, 1210, 4000], 20000),                       # few ids, large values
                            (list(range(5, 305)), list(range(1000, 1300)), 200000)):                 # 300 x 300: global-atomic path
        la = np.where(rs.uniform(size=n) < 0.2, -1, rs.choice(ids_a, n)).astype(np.int64)
        lb = np.where(rs.uniform(size=n) < 0.3, -1, rs.choice(ids_b, n)).astype(np.int64)
        assert ops.present_ids([dev(la)]) == ref.present_ids([torch.from_numpy(la)])
        assert ops.present_ids([dev(la[: n // 2]), dev(la[n // 2:])], cap=max(ids_a) + 1) == sorted(set(la[la > 0].tolist()))
        assert ops.max_label([dev(la), dev(lb)]) == max(la.max(), lb.max()) and ops.max_label([dev(np.full(7, -1, np.int64))]) == -1
        got = ops.overlap_counts(dev(la), dev(lb), ids_a, ids_b)
        exp = ref.overlap_counts(torch.from_numpy(la), torch.from_numpy(lb), ids_a, ids_b)
        assert all(np.array_equal(x, y) for x, y in zip(got, exp))
        assert ops.present_ids([dev(la)], with_outlier=True) == (sorted(set(la[la > 0].tolist())), True)
        assert ops.present_ids([dev(np.abs(la))], with_outlier=True)[1] is False
        perm_a, perm_b = ids_a[::-1], ids_b[1:] + ids_b[:1]                                           # any id order: rows / columns follow it
        got = ops.overlap_counts(dev(la), dev(lb), perm_a, perm_b)
        exp = ref.overlap_counts(torch.from_numpy(la), torch.from_numpy(lb), perm_a, perm_b)
        assert all(np.array_equal(x, y) for x, y in zip(got, exp))
    (ids,), (neg,) = ops.label_sets([[dev(np.array([3, -1, 7], np.int64)), dev(np.zeros(0, np.int64))]], 10)
    assert ids == [3, 7] and neg is True and ops.label_sets([[dev(np.array([3, 7], np.int64))]], 10) == ([[3, 7]], [False])
    with pytest.raises(AssertionError):
        ops.present_ids([dev(np.array([5, 900], np.int64))], cap=100)                                # a wrong bound is caught


def test_cluster_batch_equals_separate_calls(hip, golden):
    """stemseg_hip_cluster_batch: several independent point sets through ONE sequence of launches (grid.y = set) -- labels and
    the whole clustering record bit-identical to separate stemseg_hip_cluster calls; sets of very different sizes (one of them
    larger than a launch's one-point-per-thread capacity), an empty set, n_points on the device."""
    rs = np.random.RandomState(17)
    params = hip.make_cluster_params(0.5, 0.3, 0.6, 20, [0.3, 0.3])
    sets = []
    for n, k in ((3000, 5), (207360, 20), (1, 1), (70000, 9), (300000, 12), (0, 0), (17, 2), (40000, 25), (9000, 3)):
        centers = rs.uniform(-1, 1, (max(k, 1), 4)).astype(np.float32)
        which = rs.randint(0, max(k, 1), n)
        emb = (centers[which] + 0.03 * rs.standard_normal((n, 4))).astype(np.float32)
        bw = (20 + rs.uniform(0, 5, (n, 2))).astype(np.float32)
        seed = rs.uniform(0, 1, n).astype(np.float32)
        n_dev = dev(np.array([max(n - 3, 0)], np.int64)) if n > 100 else None
        sets.append((dev(emb), dev(bw), dev(seed), n_dev))
    sep = []
    for e, b, s_, nd in sets:
        if e.shape[0] == 0:
            sep.append(None)
            continue
        labels, meta, _, _ = hip.cluster(e, b, s_, params, 7, nd)
        sep.append((labels.clone(), meta.clone()))
    got = hip.cluster_batch(sets, params, 7)
    torch.cuda.synchronize()
    assert len(got) == len(sets)
    for i, ((labels, meta), ref) in enumerate(zip(got, sep)):
        if ref is None:
            assert hip.read_cluster_meta(meta).K == 0
            continue
        n = sets[i][0].shape[0] if sets[i][3] is None else int(sets[i][3].item())
        assert torch.equal(labels[:n], ref[0][:n]) and torch.equal(meta, ref[1]), "set %d differs" % i
        assert hip.read_cluster_meta(meta).K >= (1 if n > 1000 else 0)


def test_nonfinite_flags_op(hip):
    x = torch.randn(7, 1000, device="cuda")
    assert int(hip.nonfinite_flags(x).sum()) == 0
    x[3, 500] = float("inf")
    x[6, 999] = float("nan")
    f = hip.nonfinite_flags(x).cpu().numpy()
    n, per = x.numel(), -(-x.numel() // hip.NONFINITE_FLAGS)
    assert f.sum() == 2 and f[(3 * 1000 + 500) // per] == 1 and f[(n - 1) // per] == 1
    x[3, 500], x[6, 999] = 0.0, 0.0
    assert int(hip.nonfinite_flags(x, hip.nonfinite_flags(x)).sum()) == 0                 # flags are rewritten, not accumulated
    st = hip.overflow_status([x[:4], x[4:], torch.full((5,), float("-inf"), device="cuda")])    # adjacent views -> one run
    assert tuple(st.shape) == (2, hip.NONFINITE_FLAGS) and int(st[0].sum()) == 0 and int(st[1].sum()) >= 1


def test_overflow_guard_reruns_the_clip_in_bf16x6(hip):
    """Frames scaled far beyond pixel range drive the first f16x3 convolutions past |activation| = 2.6e5: the head outputs come back
    non-finite (every ReLU / pool on the way keeps NaN), the clustering read-back raises instead of returning labels, and
    ``ClipPipeline.step_checked`` re-runs the clip in bf16x6 (fp32's exponent range) -- same result as a plain bf16x6 run -- and
    restores the model's mode."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    config.load_preset("davis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel()
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 29))).reshape(v.shape) for k, v in sd.items()}
        new["seediness_head.conv_out.weight"] = new["seediness_head.conv_out.weight"] * 40.0
        model._model.load_state_dict(new)
        model.set_precision("f16x3")
        pipe = ClipPipeline(model, seediness_thresh=0.5)
        clip = dev(synth.synth_frames(8, 96, 160, seed=4).astype(np.float32).transpose(0, 3, 1, 2) - 110.0)
        ok = pipe.step(clip)
        assert int(ok["status"].sum()) == 0 and hip.read_cluster_meta(ok["meta"], ok["status"]).K >= 0
        big = clip * 3.0e4
        bad = pipe.step(big)
        assert int(bad["status"].sum()) > 0 and not bool(torch.isfinite(bad["emb"]).all())
        with pytest.raises(hip.NonFiniteError):
            hip.read_cluster_meta(bad["meta"], bad["status"])
        out, meta = pipe.step_checked(big)
        assert model._model.backbone.precision == "f16x3" and model._model.embedding_head.precision == "f16x3"
        assert bool(torch.isfinite(out["emb"]).all()) and int(out["status"].sum()) == 0
        got = {k: out[k].clone() for k in ("emb", "bw", "seed", "labels")}
        model.set_precision("bf16x6")
        ref = pipe.step(big)
        assert all(torch.equal(got[k], ref[k]) for k in got) and hip.read_cluster_meta(ref["meta"], ref["status"]).K == meta.K
        model.set_precision("f16x3")
        # the reference-API flow: InferenceModel.forward re-runs the sequence itself
        res = model(big, [list(range(8))])
        assert all(bool(torch.isfinite(getattr(res["embeddings"][0], k)).all()) for k in ("embeddings", "bandwidths", "seediness"))
        assert model._model.backbone.precision == "f16x3"
        model.overflow_fallback = None
        with pytest.raises(hip.NonFiniteError):
            model(big, [list(range(8))])
    finally:
        config.load_preset("defaults")


@pytest.mark.parametrize("shape", [(2, 64, 96), (3, 66, 98), (1, 34, 258), (5, 480, 864)])
def test_stem_conv_vs_fp64(hip, shape):
    """The stem alone (stemseg_hip_stem_conv: 7x7 stride 2 pad 3 + folded-BN bias + ReLU on v_mfma_f32_32x32x2_f32) against an
    fp64 convolution, on sizes whose last tiles hang over the right / bottom edges; bit-identical run to run."""
    import torch.nn.functional as F
    T, H, W = shape
    rs = np.random.RandomState(T * 1000 + H)
    frames = (rs.randint(0, 256, (T, 3, H, W)).astype(np.float32) - np.array([102.9801, 115.9465, 122.7717], np.float32)[None, :, None, None])
    w = (rs.standard_normal((64, 3, 7, 7)) * (2.0 / 147) ** 0.5).astype(np.float32)
    b = rs.standard_normal(64).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(frames).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2, padding=3)).permute(1, 0, 2, 3).numpy()
    got = hip.stem_conv(dev(frames), dev(w), dev(b))
    got2 = hip.stem_conv(dev(frames), dev(w), dev(b))
    assert tuple(got.shape) == (64, T, H // 2, W // 2) and torch.equal(got, got2)
    err = np.abs(got.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("[stem] %s max rel err vs fp64 %.3e" % (shape, err))
    assert err <= 2e-6


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 96, 160), (3, 70, 134)])
def test_stem_as_4x4_conv_over_space_to_depth_vs_fp64(hip, shape):
    """The f16x3 mode's stem: space-to-depth (S[(p*2+q)*3+c][Y][X] = in[c][2Y+p][2X+q], two halo positions before, one after) followed by a
    stride-1 1x4x4 convolution on the split-staged MFMA kernel with W2[co][(p,q,c)][a][b] = w[co][c][2a+p-1][2b+q-1] -- here the
    transform in torch and the convolution through stemseg_hip_conv3d (kernel (1, 4, 4), 12 input channels = one 16-channel chunk),
    against the fp64 7x7 stride-2 convolution and next to the exact fp32-MFMA stem: the same fp32-level error (22-bit operands)."""
    import torch.nn.functional as F
    T, H, W = shape
    rs = np.random.RandomState(T * 1000 + H + 7)
    frames = (rs.randint(0, 256, (T, 3, H, W)).astype(np.float32) - np.array([102.9801, 115.9465, 122.7717], np.float32)[None, :, None, None])
    w = (rs.standard_normal((64, 3, 7, 7)) * (2.0 / 147) ** 0.5).astype(np.float32)
    b = rs.standard_normal(64).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(frames).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2, padding=3)).permute(1, 0, 2, 3).numpy()
    H2, W2 = H // 2, W // 2
    x = torch.from_numpy(frames)
    s2d = x.reshape(T, 3, H2, 2, W2, 2).permute(3, 5, 1, 0, 2, 4).reshape(12, T, H2, W2)          # [(p, q, c)][T][Y][X]
    pitch = (W2 + 3 + 3) // 4 * 4
    buf = torch.zeros(12, T, H2 + 3, pitch)
    buf[:, :, 2:2 + H2, 2:2 + W2] = s2d
    buf = buf.cuda()
    w8 = torch.zeros(64, 3, 8, 8)
    w8[:, :, 1:, 1:] = torch.from_numpy(w)
    w2 = w8.reshape(64, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(64, 12, 1, 4, 4).contiguous().cuda()
    vin = hip.Volume(buf.data_ptr(), T * (H2 + 3) * pitch, (H2 + 3) * pitch, pitch, 12, T, H2 + 3, W2 + 3, buf.numel())
    out = torch.empty(64, T, H2, W2, device="cuda")
    hip.conv3d(vin, hip.pack_conv_weight_any(w2, "f16x3"), dev(b), hip.dense_volume(out), (1, 4, 4), 0, None, dict(relu=1, precision="f16x3"))
    torch.cuda.synchronize()
    exact = hip.stem_conv(dev(frames), dev(w), dev(b)).cpu().numpy()
    e_new = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
    e_old = np.abs(exact - ref).max() / np.abs(ref).max()
    print("[stem s2d] %s max rel err vs fp64: f16x3 4x4 form %.3e, exact fp32-MFMA stem %.3e" % (shape, e_new, e_old))
    assert e_new <= 2e-6


def test_chainer_exact_cost_ties_on_gpu_vs_golden(hip, golden):
    """tests/golden/chainer_ties.npz through the HIP chainer: exact Hungarian cost ties resolved in the reference's id enumeration
    order (online_chainer.reference_id_order) -- device-side id sets now also report whether the outlier id occurs."""
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    fg, dicts, exp = synth.tie_sequence_case(golden("chainer_ties"), dev)
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cuda:0"), 1.0)
    synth.check_long_sequence(ch.process(torch.from_numpy(fg), dicts), exp)


def test_chainer_track_ids_beyond_500_on_gpu_vs_golden(hip, golden):
    """The 48-clip sequence whose track ids the reference drives to 527 (tests/golden/chainer_long.npz): tracks, counts,
    lifetimes and per-clip instance lists identical through the HIP chainer."""
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    fg, dicts, exp = synth.long_sequence_case(golden("chainer_long"), dev)
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cuda:0"), 1.0)
    assert synth.check_long_sequence(ch.process(torch.from_numpy(fg), dicts), exp) > 500


# ------------------------------------------------------------------------------------------------ chainer / model end to end
@pytest.mark.parametrize("tag", ["seq20_ov4", "seq14_ov6", "seq8_single"])
def test_chainer_on_gpu_vs_golden(hip, golden, tag):
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    g = golden("chainer")
    emb, bw, sd, fg = g[tag + "__emb"], g[tag + "__bw"], g[tag + "__sd"], g[tag + "__fg"]
    clips = g[tag + "__subseqs"].tolist()
    dicts = [dict(frames=list(fr), embeddings=dev(emb[:, fr]), bandwidths=dev(bw[:, fr]), seediness=dev(sd[:, fr])) for fr in clips]
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cuda:0"), 1.0)
    (track, counts, life), mask_idxes, clip_labels, _, meta = ch.process(torch.from_numpy(fg), dicts)
    for t, l in enumerate(track):
        assert np.array_equal(l.numpy(), g["%s_track_%02d" % (tag, t)]), (tag, t)
        ys, xs = np.nonzero(fg[t])
        assert np.array_equal(mask_idxes[t][0].numpy(), ys) and np.array_equal(mask_idxes[t][1].numpy(), xs)
    assert sorted(counts.items()) == [tuple(r) for r in g[tag + "__pt_counts"].tolist()]
    assert sorted(life.items()) == [tuple(r) for r in g[tag + "__lifetimes"].tolist()]
    for i in range(len(clips)):
        assert meta[i]["instance_labels"] == g["%s_clip%d_instance_labels" % (tag, i)].tolist()


def test_chainer_resize_path_on_gpu(hip, golden):
    from stemseg_amd.inference.online_chainer import OnlineChainer
    g = golden("chainer")
    e = dev(g["resize__in"])
    sub = {"embeddings": e, "seediness": e[:1].contiguous(), "bandwidths": e[:2].contiguous()}
    OnlineChainer(None, 4.0).resize_tensors(sub)
    assert report("resize x4", sub["embeddings"].cpu().numpy(), g["resize__emb"]) <= 1e-6


def test_inference_model_vs_golden(hip, golden):
    """frames -> batched encoder -> both HIP decoders -> fused bandwidth activation, per clip, incl. the
    short-video dedup case; then the cross-clip seediness-averaged fg mask (model_davis golden, R-50)."""
    from stemseg_amd import config
    from stemseg_amd.inference.main import fg_masks_from_seediness
    from stemseg_amd.modeling.inference_model import InferenceModel
    g = golden("model_davis")
    config.load_preset("davis")
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = 96, 128
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    model = InferenceModel()
    sd = model._model.state_dict()
    new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 21))).reshape(v.shape) for k, v in sd.items()}
    model._model.load_state_dict(new)
    model = model.cuda()
    for tag, nframes in (("seq12", 12), ("seq5", 5)):
        frames = synth.synth_frames(nframes, 96, 128, seed=21)
        res = model([f for f in frames], g[tag + "__subseqs"].tolist())
        for i, e in enumerate(res["embeddings"]):
            assert list(e.subseq_frames) == g["%s_c%d_frames" % (tag, i)].tolist()
            assert report("%s clip%d emb" % (tag, i), e.embeddings.cpu().numpy(), g["%s_c%d_emb" % (tag, i)]) <= 1e-3
            rb = g["%s_c%d_bw" % (tag, i)]
            assert report("%s clip%d bw (rel)" % (tag, i), e.bandwidths.cpu().numpy() / rb, np.ones_like(rb)) <= 1e-3
            assert report("%s clip%d seed" % (tag, i), e.seediness.cpu().numpy(), g["%s_c%d_seed" % (tag, i)]) <= 1e-3
        # fg mask kernel on the GOLDEN seediness (so the comparison is exact, not tolerance-limited)
        entries = [(g["%s_c%d_frames" % (tag, i)].tolist(), None, None, dev(g["%s_c%d_seed" % (tag, i)])) for i in range(len(res["embeddings"]))]
        fg = fg_masks_from_seediness(entries, float(g[tag + "__fg_thr"])).cpu().numpy()
        assert np.array_equal(fg, g[tag + "__fg"])
    config.load_preset("defaults")


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
def test_config0_vs_reference_cpu_path(hip, golden, precision):
    """BASELINE configs[0] -- one synthetic 8 x 256 x 448 clip, random-init ResNet-50 -- through the whole HIP path (uint8 frames
    -> pre-processing -> encoder -> decoders -> fg mask -> gather -> clustering -> chainer) against what the REFERENCE itself
    computed on CPU for the same frames and weights (tests/golden/config0.npz): maps <= 1e-3, foreground mask and instance
    labels compared point by point."""
    from stemseg_amd import config
    from stemseg_amd.inference.main import TrackGenerator
    from stemseg_amd.modeling.inference_model import InferenceModel
    from tests.test_oracle_vs_golden import config0_inputs
    g = golden("config0")
    sd, frames, thr, min_seed = config0_inputs(g)
    config.load_preset("davis")
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = 256, 448
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    config.cfg.CLUSTERING.MIN_SEEDINESS_PROB = min_seed
    try:
        model = InferenceModel()
        msd = model._model.state_dict()
        model._model.load_state_dict({k: torch.from_numpy(np.asarray(sd[k])).reshape(msd[k].shape) for k in msd})
        model = model.cuda()
        model.set_precision(precision)
        exact = precision in ("f32", "bf16x6", "f16x3")        # the split modes with fp32-level error are held to the fp32 standard
        tg = TrackGenerator(model, "davis", seediness_thresh=thr, frame_overlap=4)
        embeddings, fg, _ = tg.do_inference([f for f in frames])
        e = embeddings[0]
        assert tuple(e.embeddings.shape) == tuple(g["shape"].tolist())
        assert report("config0 emb vs reference", e.embeddings.cpu().numpy().reshape(-1)[::5], g["emb"]) <= 1e-3
        assert report("config0 seediness vs reference", e.seediness.cpu().numpy().reshape(-1)[::5], g["seed"]) <= 1e-3
        assert report("config0 bandwidth (rel) vs reference", e.bandwidths.cpu().numpy().reshape(-1)[::5] / g["bw"], np.ones_like(g["bw"])) <= 1e-3
        ref_fg = np.unpackbits(g["fg_bits"])[:int(np.prod(g["fg_shape"]))].reshape(g["fg_shape"]).astype(bool)
        got_fg = fg.cpu().numpy().astype(bool)
        (track, counts, life), mask_idxes, _, _, meta = tg.do_clustering(embeddings, fg)
        ref_lab = np.full(ref_fg.size, -2, np.int64)
        ref_lab[np.flatnonzero(ref_fg.reshape(-1))] = g["labels"].astype(np.int64)
        got_lab = np.full(ref_fg.size, -2, np.int64)
        got_lab[np.flatnonzero(got_fg.reshape(-1))] = torch.cat([t.cpu() for t in track]).numpy()
        both = (ref_fg & got_fg).reshape(-1)
        agree = float((ref_lab[both] == got_lab[both]).mean())
        print("[parity] config0 vs reference: fg %d vs %d (%d pixels differ), %d instances vs %d, labels identical on %.4f of the common fg"
              % (got_fg.sum(), ref_fg.sum(), (got_fg != ref_fg).sum(), len(meta[0]["instance_labels"]), len(g["instance_labels"]), agree))
        print("[labels] config0 %s: fg differs in %d pixels, label agreement %.5f, instance list %s"
              % (precision, (got_fg != ref_fg).sum(), agree, "identical" if meta[0]["instance_labels"] == g["instance_labels"].tolist() else "DIFFERENT"))
        if exact:
            # exact-or-in-band: a foreground pixel may differ from the reference only where the seediness sits within
            # 1e-4 of the threshold (two fp32 pipelines' maps agree to ~3e-5 here), and every label on the common foreground is identical
            flips = np.flatnonzero((got_fg != ref_fg).reshape(-1))
            seed_map = e.seediness.cpu().numpy().reshape(-1)
            assert flips.size <= 4 and (np.abs(seed_map[flips] - thr) <= 1e-4).all(), "fg pixels differ away from the threshold band: %s" % np.abs(seed_map[flips] - thr)
            assert agree == 1.0, "labels differ on the common foreground"
        else:
            assert (got_fg != ref_fg).sum() <= 60 and agree >= 0.85
        assert meta[0]["instance_labels"] == g["instance_labels"].tolist()
    finally:
        config.load_preset("defaults")


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
def test_ytvis_flow_vs_reference(hip, golden, precision):
    """BASELINE configs[2] flow (reduced size) vs the REFERENCE's own CPU result (tests/golden/model_ytvis.npz): YouTube-VIS preset
    -- 7-channel embedding head with in-head seediness, 40+1-channel semseg head (inter [256]*4, wide head on the MFMA conv),
    --resize_embeddings: semseg logits x4, averaged over two overlapping clips, fg = sigmoid > 0.5, class argmax; the chainer
    resizes embeddings / bandwidths / seediness x4 (trilinear kernel) and clusters ~80 k points per sequence at full resolution."""
    from stemseg_amd import config
    from stemseg_amd.inference.main import TrackGenerator
    from stemseg_amd.modeling.inference_model import InferenceModel
    g = golden("model_ytvis")
    config.load_preset("ytvis")
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = 96, 128
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    config.cfg.CLUSTERING.MIN_SEEDINESS_PROB = float(g["min_seed"])
    try:
        model = InferenceModel(semseg_output_type="argmax", resize_scale=4.0)
        msd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 81))).reshape(v.shape) for k, v in msd.items()}
        new["embedding_head.conv_seediness.weight"] = new["embedding_head.conv_seediness.weight"] * 6.0
        model._model.load_state_dict(new)
        model = model.cuda()
        model.set_precision(precision)
        exact = precision in ("f32", "bf16x6", "f16x3")        # the split modes with fp32-level error are held to the fp32 standard
        frames = synth.synth_frames(12, 96, 128, seed=81)
        tg = TrackGenerator(model, "ytvis", resize_scale=4.0, frame_overlap=4)
        out = model([f for f in frames], g["subseqs"].tolist())
        assert report("ytvis semseg fg prob vs reference", out["fg_masks"].cpu().numpy().reshape(-1)[::3], g["fg_probs"]) <= 1e-3
        mc = out["multiclass_masks"].cpu().numpy()
        print("[parity] ytvis class argmax: %.5f identical" % (mc == g["multiclass"]).mean())
        assert (mc == g["multiclass"]).mean() > 0.999
        for i, e in enumerate(out["embeddings"]):
            assert report("ytvis clip %d emb vs reference" % i, e.embeddings.cpu().numpy().reshape(-1)[::3], g["c%d_emb" % i]) <= 1e-3
            assert report("ytvis clip %d seediness vs reference" % i, e.seediness.cpu().numpy().reshape(-1)[::3], g["c%d_seed" % i]) <= 1e-3
        embeddings, fg, _ = tg.do_inference([f for f in frames])
        ref_fg = np.unpackbits(g["fg_bits"])[:int(np.prod(g["fg_shape"]))].reshape(g["fg_shape"]).astype(bool)
        got_fg = fg.cpu().numpy().astype(bool)
        (track, counts, life), _, _, _, meta = tg.do_clustering(embeddings, fg)
        ref_lab = np.full(ref_fg.size, -2, np.int64)
        ref_lab[np.flatnonzero(ref_fg.reshape(-1))] = g["labels"].astype(np.int64)
        got_lab = np.full(ref_fg.size, -2, np.int64)
        got_lab[np.flatnonzero(got_fg.reshape(-1))] = torch.cat([t.cpu() for t in track]).numpy()
        both = (ref_fg & got_fg).reshape(-1)
        agree = float((ref_lab[both] == got_lab[both]).mean())
        print("[parity] ytvis flow vs reference: fg %d vs %d (%d pixels differ), labels identical on %.4f of the common fg, tracks %s"
              % (got_fg.sum(), ref_fg.sum(), (got_fg != ref_fg).sum(), agree, sorted(counts.items())[:6]))
        # (the fixture keeps the seediness sigmoid out of saturation: a plateau of exact 1.0 values, once resized x4, leaves the
        #  round's arg-max to last-bit differences between any two fp32 resamplers -- measured 3.6 % label differences with a x30
        #  gain, against the reference AND against the CPU oracle alike -- and a different, equally good seed moves the boundary)
        print("[labels] ytvis %s: fg differs in %d pixels, label agreement %.5f" % (precision, (got_fg != ref_fg).sum(), agree))
        assert (got_fg != ref_fg).mean() < 1e-3 and agree >= (0.999 if exact else 0.99)
        for i in range(2):
            assert meta[i]["instance_labels"] == g["c%d_instance_labels" % i].tolist()
        # the ORACLE chain on the SAME head outputs (CPU x4 trilinear resize, CPU clustering and stitching) must agree too
        from stemseg_amd.inference.clusterers import SequentialClustering
        from stemseg_amd.inference.online_chainer import OnlineChainer
        from tests.oracle_ops import OracleChainerOps
        embeddings2 = model([f for f in frames], g["subseqs"].tolist())["embeddings"]
        dicts = [dict(frames=list(e.subseq_frames), embeddings=e.embeddings.cpu(), bandwidths=e.bandwidths.cpu(), seediness=e.seediness.cpu())
                 for e in embeddings2]
        ref_chain = OnlineChainer(SequentialClustering(0.5, 0.3, float(g["min_seed"]), 2, [0.3, 0.3], "cpu"), 4.0, ops=OracleChainerOps())
        (rtrack, rcounts, _), _, _, _, _ = ref_chain.process(fg.cpu(), dicts)
        same = float(np.mean([float((a.cpu() == b).float().mean()) for a, b in zip(track, rtrack) if b.numel()]))
        print("[parity] ytvis flow vs oracle chain on the same head outputs: %.5f of labels identical" % same)
        assert same >= 0.999        # (same head outputs on both sides: holds in either precision)
    finally:
        config.load_preset("defaults")


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
def test_kitti_flow_vs_reference(hip, golden, precision):
    """KITTI-MOTS preset ('xyt' embeddings: the time coordinate is an embedding dimension, no free dims; in-head seediness; 3+1
    channel semseg head through the fused heads kernel) at a reduced wide-aspect size, 14 frames as three overlapping clips, vs
    the REFERENCE's own CPU result (tests/golden/model_kitti.npz): semseg fg / class probabilities, embeddings, stitched tracks."""
    from stemseg_amd import config
    from stemseg_amd.inference.main import TrackGenerator
    from stemseg_amd.modeling.inference_model import InferenceModel
    g = golden("model_kitti")
    config.load_preset("kittimots")
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = 96, 320
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    config.cfg.CLUSTERING.MIN_SEEDINESS_PROB = float(g["min_seed"])
    try:
        model = InferenceModel(semseg_output_type="probs")
        msd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 91))).reshape(v.shape) for k, v in msd.items()}
        new["embedding_head.conv_seediness.weight"] = new["embedding_head.conv_seediness.weight"] * 6.0
        model._model.load_state_dict(new)
        model = model.cuda()
        model.set_precision(precision)
        exact = precision in ("f32", "bf16x6", "f16x3")        # the split modes with fp32-level error are held to the fp32 standard
        frames = synth.synth_frames(14, 60, 190, seed=91)
        tg = TrackGenerator(model, "kittimots", frame_overlap=4)
        out = model([f for f in frames], g["subseqs"].tolist())
        assert report("kitti semseg fg prob vs reference", out["fg_masks"].cpu().numpy().reshape(-1)[::3], g["fg_probs"]) <= 1e-3
        assert report("kitti class probs vs reference", out["multiclass_masks"].cpu().numpy().reshape(-1)[::5], g["class_probs"]) <= 1e-3
        for i, e in enumerate(out["embeddings"]):
            assert report("kitti clip %d emb vs reference" % i, e.embeddings.cpu().numpy().reshape(-1)[::3], g["c%d_emb" % i]) <= 1e-3
        embeddings, fg, _ = tg.do_inference([f for f in frames])
        ref_fg = np.unpackbits(g["fg_bits"])[:int(np.prod(g["fg_shape"]))].reshape(g["fg_shape"]).astype(bool)
        got_fg = fg.cpu().numpy().astype(bool)
        (track, counts, life), _, _, _, meta = tg.do_clustering(embeddings, fg)
        ref_lab = np.full(ref_fg.size, -2, np.int64)
        ref_lab[np.flatnonzero(ref_fg.reshape(-1))] = g["labels"].astype(np.int64)
        got_lab = np.full(ref_fg.size, -2, np.int64)
        got_lab[np.flatnonzero(got_fg.reshape(-1))] = torch.cat([t.cpu() for t in track]).numpy()
        both = (ref_fg & got_fg).reshape(-1)
        agree = float((ref_lab[both] == got_lab[both]).mean())
        print("[parity] kitti flow vs reference: fg %d vs %d (%d pixels differ), labels identical on %.4f of the common fg, tracks %s"
              % (got_fg.sum(), ref_fg.sum(), (got_fg != ref_fg).sum(), agree, sorted(counts.items())[:8]))
        print("[labels] kitti %s: fg differs in %d pixels, label agreement %.5f" % (precision, (got_fg != ref_fg).sum(), agree))
        assert (got_fg != ref_fg).mean() < 1e-3 and agree >= (0.999 if exact else 0.99)
        assert not exact or sorted(counts.items()) == [tuple(r) for r in g["pt_counts"].tolist()] or (got_fg != ref_fg).any()
        for i in range(3):
            assert meta[i]["instance_labels"] == g["c%d_instance_labels" % i].tolist()
    finally:
        config.load_preset("defaults")


def test_sequence_end_to_end_tracks_and_masks(hip):
    """A 14-frame sequence through the whole device path -- TrackGenerator (pre-processing, encoder, decoders, fg mask from the
    clip-averaged seediness, gather, clustering, Hungarian stitching) and MaskMaterializer -- against the ORACLE chain (CPU
    clustering / stitching twin, oracle mask resampling) run on the same head outputs: track ids per point, point counts,
    lifetimes and final masks must be identical.  The sharded driver (run_sequence_sharded, single process) must agree too."""
    from oracle import masks as omask
    from stemseg_amd import config
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.main import TrackGenerator, fg_masks_from_seediness
    from stemseg_amd.inference.online_chainer import OnlineChainer
    from stemseg_amd.inference.output_utils import MaskMaterializer
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import run_sequence_sharded
    from tests.oracle_ops import OracleChainerOps
    config.load_preset("davis")
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = 96, 128
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel()
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 61))).reshape(v.shape) for k, v in sd.items()}
        new["seediness_head.conv_out.weight"] = new["seediness_head.conv_out.weight"] * 12.0
        model._model.load_state_dict(new)
        model = model.cuda()
        frames = synth.synth_frames(14, 90, 120, seed=61)                   # resized to 96x128 by the pre-processing kernel
        # thresholds from the data (random-init weights give an arbitrary seediness range): half of the pixels foreground,
        # seeds from the upper quartile
        from stemseg_amd.modeling.inference_model import preprocess_frames
        probe = model.embed_frames(preprocess_frames(frames[:8])[0])[2].flatten().float()
        thr = float(probe.median())
        config.cfg.CLUSTERING.MIN_SEEDINESS_PROB = float(probe.quantile(0.75))
        tg = TrackGenerator(model, "davis", seediness_thresh=thr, frame_overlap=4)
        embeddings, fg, _ = tg.do_inference([f for f in frames])
        (track, counts, life), mask_idxes, _, _, meta = tg.do_clustering(embeddings, fg)
        assert len(track) == 14 and fg.shape == (14, 24, 32)
        # oracle chain on the same head outputs
        dicts = [dict(frames=list(e.subseq_frames), embeddings=e.embeddings.cpu(), bandwidths=e.bandwidths.cpu(), seediness=e.seediness.cpu())
                 for e in embeddings]
        c = config.cfg.CLUSTERING
        ref_chain = OnlineChainer(SequentialClustering(c.PRIMARY_PROB_THRESHOLD, c.SECONDARY_PROB_THRESHOLD, c.MIN_SEEDINESS_PROB, 2,
                                                       config.cfg.TRAINING.LOSSES.EMBEDDING.FREE_DIM_STDS, "cpu"), 1.0, ops=OracleChainerOps())
        (rtrack, rcounts, rlife), ridx, _, _, rmeta = ref_chain.process(fg.cpu(), dicts)
        n_inst = 0
        for t in range(14):
            assert torch.equal(track[t].cpu(), rtrack[t]), "frame %d: track labels differ" % t
            n_inst = max(n_inst, int(track[t].max().item()) if track[t].numel() else 0)
        assert dict(counts) == dict(rcounts) and dict(life) == dict(rlife) and n_inst >= 2 and sum(counts.values()) > 1000
        assert [m["instance_labels"] for m in meta] == [m["instance_labels"] for m in rmeta]
        # sharded driver, one process: identical tracks
        clips_seen = {tuple(e.subseq_frames): e for e in embeddings}
        (strack, scounts, _), _, _, _, _ = run_sequence_sharded(
            14, lambda fr: (lambda e: (e.embeddings, e.bandwidths, e.seediness))(clips_seen[tuple(sorted(set(fr)))]), tg.chainer, "davis",
            frame_overlap=4, seediness_thresh=thr)
        assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(strack, track)) and dict(scounts) == dict(counts)
        # masks at the original 90x120 size
        keep, masks = MaskMaterializer(-1).process_sequence((90, 120), mask_idxes, track, life, (24, 32), 4.0, 10)
        maps = np.zeros((14, 24, 32), np.int64)
        for t in range(14):
            maps[t][mask_idxes[t][0].cpu().numpy(), mask_idxes[t][1].cpu().numpy()] = track[t].cpu().numpy()
        ref_masks = omask.condensed_masks(maps, omask.instances_to_keep(dict(life), -1, 10), (90, 120), 96, 128).numpy()
        assert keep == omask.instances_to_keep(dict(life), -1, 10)
        bad = masks.cpu().numpy() != ref_masks
        print("[parity] sequence end to end: %d tracks kept, %d fg points, masks differ in %d / %d pixels" % (len(keep), sum(counts.values()), bad.sum(), bad.size))
        assert bad.mean() < 1e-4
    finally:
        config.load_preset("defaults")


def test_embed_many_batches_and_lanes_match_per_clip_embedding(hip):
    """ClipPipeline.embed_many (the sharded sequence driver's embedding: several clips per encoder pass, full batches as hipGraph
    replays alternating over two lanes, the remainder eagerly) returns, clip for clip, what embedding each clip on its own gives
    BIT FOR BIT -- the encoder plans every launch for a fixed frame count (ResNetFPN.plan_frames), so the K-partition of its split-K
    launches does not depend on the batch -- in the caller's clip order, and identically when repeated."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    config.load_preset("davis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel()
        names = [(k, v.shape) for k, v in model._model.state_dict().items()]
        sd = synth.synth_state_dict(names, 3)
        model._model.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(model._model.state_dict()[k].shape) for k, v in sd.items()})
        pipe = ClipPipeline(model)
        frames = (torch.from_numpy(synth.synth_frames(24, 64, 96, seed=3).astype(np.float32)).permute(0, 3, 1, 2) - 110.0).cuda().contiguous()
        clips = [list(range(s, s + 8)) for s in (0, 4, 8, 12, 16)]
        ref = [torch.cat(pipe.embed(frames[c].contiguous()), 0).clone() for c in clips]
        got = pipe.embed_many(frames, clips, batch=2, lanes=2)                       # overlapping windows: shared frames encoded once
        again = pipe.embed_many(frames, clips, batch=2, lanes=2)
        eager = pipe.embed_many(frames, clips, batch=2, lanes=2, use_graph=False)
        stacked = pipe.embed_many(frames, clips, batch=2, lanes=2, share_overlap=False)     # every clip's frames stacked per pass
        three = pipe.embed_many(frames, clips, batch=3, lanes=1)                     # 3 windows per pass (20 frames) + 2 eagerly
        torch.cuda.synchronize()
        for i in range(len(clips)):
            for name, other in (("windows x2 (graph)", got), ("again", again), ("eager", eager), ("stacked", stacked), ("windows x3", three)):
                assert torch.equal(other[i], ref[i]), "clip %d: %s differs from the clip embedded on its own" % (i, name)
        # a sequence whose length leaves an OFF-STRIDE tail clip (22 frames: windows 0, 4, 8, 12 + the tail 14..21, what
        # get_subsequence_frames cuts when (F - T) % (T - overlap) != 0): the on-stride prefix still shares the trunk
        clips2 = [list(range(s0, s0 + 8)) for s0 in (0, 4, 8, 12)] + [list(range(14, 22))]
        ref2 = [torch.cat(pipe.embed(frames[c].contiguous()), 0).clone() for c in clips2]
        got2 = pipe.embed_many(frames, clips2, batch=2, lanes=2)
        torch.cuda.synchronize()
        for i in range(len(clips2)):
            assert torch.equal(got2[i], ref2[i]), "tail-clip sequence, clip %d" % i
    finally:
        config.load_preset("defaults")


@pytest.mark.parametrize("preset", ["ytvis", "kittimots"])
def test_clip_pipeline_step_on_semseg_presets(hip, preset):
    """ClipPipeline.step on the presets with a semseg head (the bench's unit of work for BASELINE configs[2] / [4]): third decoder
    -> class logits (x resize_scale) -> fg = fg probability > 0.5 (semseg_fg_clip == accumulate + get_semseg_masks + threshold for
    one independent clip); YouTube-VIS: --resize_embeddings, head outputs x4 and clustering at full resolution.  Every stage vs
    the per-stage entry points, labels vs the oracle's clusterer on the same maps (exact)."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    config.load_preset(preset)
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        r = 4 if preset == "ytvis" else 1
        model = InferenceModel(resize_scale=float(r))
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 61))).reshape(v.shape) for k, v in sd.items()}
        new["embedding_head.conv_seediness.weight"] = new["embedding_head.conv_seediness.weight"] * 25.0
        model._model.load_state_dict(new)
        pipe = ClipPipeline(model)
        T, H, W = 8, 96, 160
        frames = dev(synth.synth_frames(T, H, W, seed=61).astype(np.float32).transpose(0, 3, 1, 2) - 110.0)
        out = pipe.step(frames)
        torch.cuda.synchronize()
        # stage by stage through the per-stage entry points
        emb, bw, seed = model.embed_frames(frames)
        logits = model.semseg_logits_clip(T, H, W, emb.device)
        assert torch.equal(out["semseg_logits"], logits)
        acc = torch.zeros((T, logits.shape[0]) + tuple(logits.shape[2:]), device="cuda")
        hip.semseg_accumulate(acc, logits.contiguous(), list(range(T)))
        fg_p, _ = hip.semseg_masks(acc, torch.ones(T, device="cuda"), None)
        fg_ref = torch.stack([hip.fg_mask(p_.contiguous(), 1.0, 0.5) for p_ in fg_p], 0)
        assert torch.equal(out["fg"], fg_ref) and float(fg_ref.float().mean()) > 0.02
        if r != 1:
            emb, bw, seed = [hip.upsample_trilinear(x.contiguous(), 1, r, r) for x in (emb, bw, seed)]
        assert tuple(out["emb"].shape[-2:]) == (H // 4 * r, W // 4 * r)
        assert torch.equal(out["emb"], emb) and torch.equal(out["bw"], bw) and torch.equal(out["seed"], seed)
        # clustering vs the oracle on the same maps
        c = config.cfg.CLUSTERING
        nfree = 2 if preset == "ytvis" else 0
        e_, b_, s_, cnt = opipe.gather_fg(emb.cpu().numpy(), bw.cpu().numpy(), seed.cpu().numpy(), fg_ref.cpu().numpy())
        ref, ref_meta = sequential_clustering(e_, b_, s_, label_start=1, min_seediness=c.MIN_SEEDINESS_PROB,
                                              free_dim_stds=[0.3, 0.3][:nfree], return_probs=True)
        n = int(out["frame_offsets"].cpu()[-1])
        meta = hip.read_cluster_meta(out["meta"])
        got = out["labels"][:n].cpu().numpy()
        assert n == ref.shape[0] and meta.K == len(ref_meta["instance_labels"])
        bad = np.flatnonzero(got != ref)
        print("[parity] step %s: %d fg points, K = %d, %d labels differ" % (preset, n, meta.K, bad.size))
        if bad.size:
            P = np.stack(ref_meta["instance_probs"])
            near = (np.abs(P - 0.5) < 2e-6).any(0) | (np.abs(P - 0.3) < 2e-6).any(0)
            assert near[bad].all()
        # the batched form (two clips through one encoder pass) gives the same per clip
        outs = pipe.step_batch(torch.cat([frames, frames.flip(0)], 0), 2)
        assert torch.equal(outs[0]["fg"], out["fg"]) and torch.equal(outs[0]["emb"], out["emb"]) and torch.equal(outs[0]["labels"], out["labels"])
    finally:
        config.load_preset("defaults")


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("size", [(96, 160), (256, 448)])
def test_clip_pipeline_end_to_end_vs_oracle(hip, precision, size):
    """One clip through ClipPipeline.step (the bench's unit of work) vs the oracle pipeline -- at a reduced size and at
    BASELINE configs[0] (one synthetic 8 x 256 x 448 clip, random-init ResNet-50: the reference's own CPU-runnable case):
    float outputs <= 1e-3; labels identical wherever the oracle's own decision has margin."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    config.load_preset("davis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    model = InferenceModel()
    names = [(k, v.shape) for k, v in model._model.state_dict().items()]
    sd = synth.synth_state_dict(names, 33)
    sd["seediness_head.conv_out.weight"] = sd["seediness_head.conv_out.weight"] * 30      # spread seediness over (0, 1)
    model._model.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(model._model.state_dict()[k].shape) for k, v in sd.items()})
    pipe = ClipPipeline(model)
    model.set_precision(precision)
    frames = torch.from_numpy(synth.synth_frames(8, size[0], size[1], seed=33).astype(np.float32)).permute(0, 3, 1, 2) - \
        torch.tensor(config.cfg.INPUT.IMAGE_MEAN)[None, :, None, None]
    out = pipe.step(frames.cuda())
    ref = opipe.embed_and_cluster_clip(frames, sd, "R-50-FPN", "xyff", 4, True, free_dim_stds=[0.3, 0.3], return_probs=True)
    assert report("pipeline emb", out["emb"].cpu().numpy(), ref["emb"].numpy()) <= 1e-3
    assert report("pipeline seed", out["seed"].cpu().numpy(), ref["seed"].numpy()) <= 1e-3
    assert report("pipeline bw (rel)", (out["bw"].cpu() / ref["bw"]).numpy(), np.ones(ref["bw"].shape)) <= 1e-3
    offs = out["frame_offsets"].cpu().numpy()
    print("[parity] pipeline: N=%d fg points, oracle N=%d, K=%d" % (offs[-1], ref["labels"].shape[0], len(ref["meta"]["instance_labels"])))
    # clustering on the ORACLE's embeddings through the HIP clusterer must reproduce the oracle's labels
    o2 = pipe.cluster(ref["emb"].cuda().contiguous(), ref["bw"].cuda().contiguous(), ref["seed"].cuda().contiguous())
    n = int(o2["frame_offsets"].cpu()[-1])
    assert n == ref["labels"].shape[0]
    bad = np.flatnonzero(o2["labels"][:n].cpu().numpy() != ref["labels"])
    print("[parity] pipeline labels: %d / %d differ" % (bad.size, n))
    if bad.size:          # identical, or inside the threshold band of the oracle's own probabilities
        P = np.stack(ref["meta"]["instance_probs"])
        near = (np.abs(P - 0.5) < 2e-6).any(0) | (np.abs(P - 0.3) < 2e-6).any(0)
        assert near[bad].all()
    config.load_preset("defaults")
