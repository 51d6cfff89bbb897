"""The split-staged convolution modes.  bf16x6 (STEMSEG_PRECISION_BF16X6): every fp32 operand split EXACTLY into three bf16
terms, six products on the bf16 matrix cores; f16x3 (STEMSEG_PRECISION_F16X3): operands scaled by powers of two, split into two
fp16 terms, three products on the fp16 matrix cores; fp32 accumulation in both.  The claim under test is "fp32-level results":
against an fp64 convolution of the same inputs the split kernel's error must be of the order of the exact-fp32-MFMA kernel's
own error (both are dominated by the rounding of the fp32 accumulation), on every kernel class, tile shape, split-K, ragged /
unaligned shapes and fused epilogue -- every test below runs once per mode."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


SP = "bf16x6"


@pytest.fixture(params=["bf16x6", "f16x3"], autouse=True)
def split_mode(request):
    global SP
    SP = request.param
    yield SP


@pytest.fixture(scope="module")
def hip():
    from stemseg_amd import hip as h
    h.require_gpu()
    return h


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _rand(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def _ref64(x, w, b, kt):
    y = F.conv3d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), None if b is None else torch.from_numpy(b).double(),
                 padding=(kt // 2, w.shape[3] // 2, w.shape[4] // 2))[0]
    return y.numpy()


def _haloed(hip, x, kt):
    """[C,T,H,W] -> zero-haloed device volume for a (kt,3,3) conv (16-B aligned rows)."""
    Cn, T, H, W = x.shape
    if kt == 3:
        buf, g = hip.alloc_padded(Cn, T, H, W)
        hip.copy_to_volume(dev(x), 0, hip.padded_interior_view(buf, g, Cn, T, H, W))
        return buf, hip.padded_halo_view(buf, g, Cn, T, H, W)
    pitch = (W + 2 + 3) // 4 * 4
    buf = torch.zeros(Cn, T, H + 2, pitch, device="cuda")
    buf[:, :, 1:H + 1, 1:W + 1] = dev(x)
    return buf, hip.Volume(buf.data_ptr(), T * (H + 2) * pitch, (H + 2) * pitch, pitch, Cn, T, H + 2, W + 2, buf.numel())


def _both(hip, vin, w, b, out_shape, k, cfg, scratch_floats=0, flat=False):
    res = {}
    for prec in ("f32", SP):
        out = torch.full(out_shape, float("nan"), device="cuda")
        vout = hip.flat_volume(out) if flat else hip.dense_volume(out)
        scratch = torch.full((scratch_floats,), float("nan"), device="cuda") if scratch_floats else None
        hip.conv3d(vin, hip.pack_conv_weight_any(dev(w), prec), None if b is None else dev(b), vout, k, cfg, scratch, dict(precision=prec))
        torch.cuda.synchronize()
        res[prec] = out.cpu().numpy().astype(np.float64)
    return res


def _check(name, res, ref):
    e32 = float(np.abs(res["f32"] - ref).max())
    e6 = float(np.abs(res[SP] - ref).max())
    scale = float(np.abs(ref).max())
    print("[%s] %-46s max|err| vs fp64: fp32-MFMA %.3e, split %.3e (ratio %.2f; max|ref| %.3g); split vs fp32-MFMA %.3e"
          % (SP, name, e32, e6, e6 / max(e32, 1e-30), scale, float(np.abs(res[SP] - res["f32"]).max())))
    assert np.isfinite(res[SP]).all()
    assert e6 <= max(3.0 * e32, 4e-7 * scale), "%s is not at the fp32 error level" % SP
    return e32, e6


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["k3", "k2", "k1"])
def test_bf16x6_every_tile_shape_vs_fp64(hip, kind, cfg):
    rs = 71 + cfg
    if kind == "k1":
        if cfg == 3:
            pytest.skip("1x1 has two tile shapes")
        Cin, Cout, V = 256, 128, 4 * 30 * 54
        x, w, b = _rand((Cin, V), rs), _rand((Cout, Cin, 1, 1, 1), rs + 1, 1.0 / np.sqrt(Cin)), _rand((Cout,), rs + 2)
        ref = w.reshape(Cout, Cin).astype(np.float64) @ x.astype(np.float64) + b.astype(np.float64)[:, None]
        xd = dev(x)                                                  # (a Volume holds a raw pointer: keep the tensor alive)
        res = _both(hip, hip.flat_volume(xd), w, b, (Cout, V), 1, cfg, flat=True)
    else:
        kt = 3 if kind == "k3" else 1
        Cin, Cout, T, H, W = (64, 128, 3, 21, 40) if kind == "k3" else (64, 128, 2, 17, 70)
        x, w, b = _rand((Cin, T, H, W), rs), _rand((Cout, Cin, kt, 3, 3), rs + 1, 1.0 / np.sqrt(Cin * 9 * kt)), _rand((Cout,), rs + 2)
        ref = _ref64(x, w, b, kt)
        buf, vin = _haloed(hip, x, kt)
        res = _both(hip, vin, w, b, (Cout, T, H, W), (kt, 3, 3), cfg)
    _check("%s cfg%d" % (kind, cfg), res, ref)


@pytest.mark.parametrize("case", [("k3", 8, 64, 3, 5, 37), ("k3", 256, 128, 2, 9, 40), ("k3", 12, 32, 1, 2, 3), ("k3", 16, 160, 4, 17, 70),
                                  ("k2", 8, 64, 2, 9, 37), ("k2", 256, 256, 2, 6, 27), ("k2", 64, 64, 3, 17, 40), ("k2", 128, 128, 8, 30, 54)])
def test_bf16x6_ragged_shapes_and_splitk(hip, case):
    """Odd extents (tiles that hang over every edge, Cin not a multiple of the chunk, Cout below the tile), with and without
    split-K (fixed-order reduce: run-to-run identical)."""
    kind, Cin, Cout, T, H, W = case
    kt = 3 if kind == "k3" else 1
    x, w, b = _rand((Cin, T, H, W), 5), _rand((Cout, Cin, kt, 3, 3), 6, 1.0 / np.sqrt(Cin * 9 * kt)), _rand((Cout,), 7)
    ref = _ref64(x, w, b, kt)
    buf, vin = _haloed(hip, x, kt)
    _check("%s %s" % (kind, case[1:]), _both(hip, vin, w, b, (Cout, T, H, W), (kt, 3, 3), 0), ref)
    r1 = _both(hip, vin, w, b, (Cout, T, H, W), (kt, 3, 3), 0, scratch_floats=16 * Cout * T * H * W)
    r2 = _both(hip, vin, w, b, (Cout, T, H, W), (kt, 3, 3), 0, scratch_floats=16 * Cout * T * H * W)
    _check("%s %s split-K" % (kind, case[1:]), r1, ref)
    assert np.array_equal(r1[SP], r2[SP])


def test_bf16x6_unaligned_input_rows(hip):
    """A dense (un-haloed) source volume whose rows are not 16-B aligned takes the scalar staging path."""
    Cin, Cout, T, H, W = 16, 64, 2, 7, 13
    x, w, b = _rand((Cin, T + 2, H + 2, W + 2), 11), _rand((Cout, Cin, 3, 3, 3), 12, 0.05), _rand((Cout,), 13)
    xd = dev(x)
    ref = F.conv3d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), torch.from_numpy(b).double())[0].numpy()
    _check("k3 unaligned rows", _both(hip, hip.dense_volume(xd), w, b, (Cout, T, H, W), 3, 0), ref)


@pytest.mark.parametrize("shape", [(64, 256, 2, 9, 37), (1024, 256, 2, 6, 27), (256, 1024, 3, 15, 27)])
def test_bf16x6_1x1_decode_residual_relu(hip, shape):
    """The encoder's 1x1 convs: flat [C][V] input, output decoded into a zero-haloed volume, residual + ReLU in the epilogue."""
    Cin, Cout, T, H, W = shape
    V = T * H * W
    x, w, b = _rand((Cin, V), 21), _rand((Cout, Cin, 1, 1, 1), 22, 1.0 / np.sqrt(Cin)), _rand((Cout,), 23)
    r = _rand((Cout, V), 24)
    ref = np.maximum(w.reshape(Cout, Cin).astype(np.float64) @ x.astype(np.float64) + b.astype(np.float64)[:, None] + r, 0.0)
    xd, rd = dev(x), dev(r)
    res = {}
    for prec in ("f32", SP):
        pitch = (W + 2 + 3) // 4 * 4
        buf = torch.zeros(Cout, T, H + 2, pitch, device="cuda")
        vout = hip.Volume(buf.data_ptr() + 4 * (pitch + 1), T * (H + 2) * pitch, (H + 2) * pitch, pitch, Cout, T, H, W, buf.numel() - (pitch + 1))
        hip.conv3d(hip.flat_volume(xd), hip.pack_conv_weight_any(dev(w), prec), dev(b), vout, 1, 0, None,
                   dict(relu=1, residual=rd, res_strides=(V, H * W, W), decode=(H, W), precision=prec))
        torch.cuda.synchronize()
        res[prec] = buf[:, :, 1:H + 1, 1:W + 1].reshape(Cout, V).cpu().numpy().astype(np.float64)
        assert float(buf[:, :, 0].abs().max()) == 0.0 and float(buf[:, :, :, 0].abs().max()) == 0.0      # the halo stays zero
    _check("k1 decode+res+relu %s" % (shape,), res, ref)


@pytest.mark.parametrize("case", [(64, 128, 2, 9, 40), (32, 256, 8, 24, 64), (256, 128, 8, 60, 108)])
def test_bf16x6_conv_with_groupnorm_statistics(hip, case):
    """conv3d_gn in x6 mode: statistics of the x6 output from the conv epilogue (plain and split-K) vs numpy on the result."""
    Cin, Cout, T, H, W = case
    x, w, b = _rand((Cin, T, H, W), 31), _rand((Cout, Cin, 3, 3, 3), 32, 1.0 / np.sqrt(Cin * 27)), _rand((Cout,), 33)
    buf, vin = _haloed(hip, x, 3)
    for scratch_floats in (0, 8 * Cout * T * H * W):
        out = torch.full((Cout, T, H, W), float("nan"), device="cuda")
        scratch = torch.empty(scratch_floats, device="cuda") if scratch_floats else None
        stats = hip.conv3d_gn(vin, hip.pack_conv_weight_any(dev(w), SP), dev(b), hip.dense_volume(out), 3, 32, 1e-5, 0, scratch, SP)
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype(np.float64).reshape(32, -1)
        mean, var = o.mean(1), o.var(1)
        st = stats.cpu().numpy().reshape(32, 2)
        assert np.abs(st[:, 0] - mean).max() <= 2e-6 * max(1.0, np.abs(mean).max())
        assert np.abs(st[:, 1] - 1.0 / np.sqrt(var + 1e-5)).max() <= 1e-4 * (1.0 / np.sqrt(var + 1e-5)).max()
    ref = _ref64(x, w, b, 3)
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-5


def test_bf16x6_block4x_shape_is_fp32_accurate_and_deterministic(hip):
    """The dominant launch (256 -> 128 channels, T=8, 120x216): x6 vs fp64 next to the fp32-MFMA kernel, twice."""
    Cin, Cout, T, H, W = 256, 128, 8, 120, 216
    x, w, b = _rand((Cin, T, H, W), 41), _rand((Cout, Cin, 3, 3, 3), 42, 1.0 / np.sqrt(Cin * 27)), _rand((Cout,), 43)
    ref = _ref64(x[:, :3, :40], w, b, 3)[:, 1]                      # fp64 on a slab (t = 1 of the first three planes, 40 rows)
    buf, vin = _haloed(hip, x, 3)
    r1 = _both(hip, vin, w, b, (Cout, T, H, W), 3, 0, scratch_floats=4 * Cout * T * H * W)
    r2 = _both(hip, vin, w, b, (Cout, T, H, W), 3, 0, scratch_floats=4 * Cout * T * H * W)
    assert np.array_equal(r1[SP], r2[SP])
    sl = {k: v[:, 1, :39] for k, v in r1.items()}
    _check("block_4x 256->128 T=8 120x216", sl, ref[:, :39])


@pytest.mark.parametrize("xscale,wscale", [(1.0, 1.0), (1e-2, 1.0), (1e-4, 1.0), (100.0, 1.0), (2e4, 1.0), (1.0, 1e-3), (1.0, 50.0), (1e-3, 1e-3), (3e3, 1e-2)])
def test_split_modes_keep_fp32_level_across_operand_magnitudes(hip, xscale, wscale):
    """The f16x3 mode scales its operands by powers of two to sit inside fp16's exponent range: activations of 1e-4 ... 2e4 times
    unit scale (pixel-scale inputs of an un-normalised backbone included) and weights of 1e-3 ... 50 times He scale must all stay
    at the fp32-MFMA error level (bf16x6 has fp32's range and passes trivially)."""
    Cin, Cout, T, H, W = 64, 128, 2, 17, 40
    x = _rand((Cin, T, H, W), 51, xscale)
    w, b = _rand((Cout, Cin, 3, 3, 3), 52, wscale / np.sqrt(Cin * 27)), _rand((Cout,), 53, xscale * wscale)
    ref = _ref64(x, w, b, 3)
    buf, vin = _haloed(hip, x, 3)
    _check("k3 x*%g w*%g" % (xscale, wscale), _both(hip, vin, w, b, (Cout, T, H, W), 3, 0), ref)


def test_f16x3_overflowing_activation_is_not_silent(hip):
    """|activation| >= 2.6e5 leaves the fp16 range of the scaled operand: the affected outputs must come back non-finite, never
    as plausible numbers."""
    Cin, Cout, V = 64, 128, 512
    x, w = _rand((Cin, V), 61), _rand((Cout, Cin, 1, 1, 1), 62, 1.0 / 8)
    x[3, 100] = 3.0e5
    xd = dev(x)
    out = torch.zeros(Cout, V, device="cuda")
    hip.conv3d(hip.flat_volume(xd), hip.pack_conv_weight_any(dev(w), "f16x3"), None, hip.flat_volume(out), 1, 0, None, dict(precision="f16x3"))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert not np.isfinite(o[:, 100]).any()
    assert np.isfinite(np.delete(o, 100, axis=1)).all()


@pytest.mark.parametrize("kind", ["k3", "k2", "k1"])
def test_split_modes_keep_fp32_level_with_per_channel_dynamic_range(hip, kind):
    """Intra-layer dynamic range: FrozenBN folded with eps = 0 (make_layers.py:43,51-63) multiplies every OUTPUT channel by its
    own gamma / sqrt(var) -- per-channel weight magnitudes log-uniform over 2^-20 ... 1 here.  Every channel is held to the
    fp32-MFMA error level RELATIVE TO ITS OWN output scale (a per-layer weight scale fails this: the small channels' low fp16
    terms go subnormal)."""
    rs = np.random.RandomState(91)
    if kind == "k1":
        Cin, Cout, V = 256, 256, 3 * 30 * 54
        ch = (2.0 ** rs.uniform(-20, 0, size=(Cout, 1, 1, 1, 1))).astype(np.float32)
        x, w, b = np.maximum(_rand((Cin, V), 92, 30.0), 0), _rand((Cout, Cin, 1, 1, 1), 93, 1.0 / np.sqrt(Cin)) * ch, None
        ref = w.reshape(Cout, Cin).astype(np.float64) @ x.astype(np.float64)
        xd = dev(x)
        res = _both(hip, hip.flat_volume(xd), w, b, (Cout, V), 1, 0, flat=True)
    else:
        kt = 3 if kind == "k3" else 1
        Cin, Cout, T, H, W = (64, 128, 3, 21, 40) if kind == "k3" else (64, 128, 2, 17, 70)
        ch = (2.0 ** rs.uniform(-20, 0, size=(Cout, 1, 1, 1, 1))).astype(np.float32)
        x, w, b = np.maximum(_rand((Cin, T, H, W), 92, 30.0), 0), _rand((Cout, Cin, kt, 3, 3), 93, 1.0 / np.sqrt(Cin * 9 * kt)) * ch, None
        ref = _ref64(x, w, b, kt)
        buf, vin = _haloed(hip, x, kt)
        res = _both(hip, vin, w, b, (Cout, T, H, W), (kt, 3, 3), 0)
    Cout = ref.shape[0]
    scale = np.abs(ref.reshape(Cout, -1)).max(1)                        # per output channel
    e32 = (np.abs(res["f32"] - ref).reshape(Cout, -1).max(1) / scale)
    e6 = (np.abs(res[SP] - ref).reshape(Cout, -1).max(1) / scale)
    print("[%s] %s per-channel dynamic range 2^-20..1: worst relative error fp32-MFMA %.3e, split %.3e" % (SP, kind, e32.max(), e6.max()))
    assert np.isfinite(res[SP]).all()
    assert e6.max() <= max(3.0 * e32.max(), 4e-7), "%s loses precision on small-magnitude output channels" % SP


def test_split_modes_encoder_with_checkpoint_like_frozen_bn_statistics(hip):
    """R-101-FPN with FrozenBN statistics like a trained checkpoint's: running_var log-uniform over 1e-6 ... 1e2, gamma over
    0.05 ... 3 (folded: per-output-channel weight scales spanning ~5 orders of magnitude inside every layer).  The split modes must
    reproduce the fp32-input MFMA encoder at ITS error level against the fp32 CPU oracle."""
    from stemseg_amd import config
    from stemseg_amd.modeling.backbone import ResNetFPN
    from oracle import encoder as oenc
    from tests import synth
    config.load_preset("davis")
    bb = ResNetFPN("R-101-FPN").cuda()
    rs = np.random.RandomState(5)
    sd = {k: np.asarray(synth.synth_param(k, v.shape, 17)).reshape(v.shape).astype(np.float32) for k, v in bb.state_dict().items()}
    # consistent statistics, as training leaves them: a channel's running_var IS the variance of its conv output, so the conv rows
    # are scaled by sqrt(var) (var log-uniform over 1e-6 ... 1e2) and the folded weight keeps gamma / sqrt(var) * sqrt(var) = gamma
    # times the unit-variance row -- gamma log-uniform over 1e-4 ... 3 (dead and strong channels side by side in every layer)
    for k in list(sd):
        if not k.endswith("running_var"):
            continue
        bn = k[:-len(".running_var")]
        conv = bn.replace("bn1", "conv1").replace("bn2", "conv2").replace("bn3", "conv3")
        if bn.endswith("downsample.1"):
            conv = bn[:-1] + "0"
        var = (10.0 ** rs.uniform(-6, 2, size=sd[k].shape)).astype(np.float32)
        sd[k] = var
        sd[conv + ".weight"] = sd[conv + ".weight"] * np.sqrt(var)[:, None, None, None]
        top = 0.5 if bn.endswith("bn3") else 3.0
        sd[bn + ".weight"] = (10.0 ** rs.uniform(-4, np.log10(top), size=var.shape)).astype(np.float32)
        sd[bn + ".running_mean"] = (sd[bn + ".running_mean"] * np.sqrt(var)).astype(np.float32)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    bb.load_state_dict(sd)
    frames = torch.from_numpy(synth.synth_frames(2, 64, 96, seed=3).astype(np.float32)).permute(0, 3, 1, 2) - 110.0
    ref_d = oenc.resnet_fpn(frames, {k: v.numpy() for k, v in sd.items()}, "R-101-FPN", prefix="")
    ref = [ref_d[s_].numpy() for s_ in (4, 8, 16, 32)]
    outs = {}
    for prec in ("f32", SP):
        bb.precision = prec
        outs[prec] = [o.cpu().numpy() for o in bb.forward(frames.cuda())]
    for lvl in range(4):
        r = np.asarray(ref[lvl])
        scale = float(np.abs(r).max())
        e32 = float(np.abs(outs["f32"][lvl] - r).max()) / scale
        e6 = float(np.abs(outs[SP][lvl] - r).max()) / scale
        print("[%s] R-101 checkpoint-like BN, FPN level %d: rel err fp32-MFMA %.3e, split %.3e (max|ref| %.3g)" % (SP, lvl, e32, e6, scale))
        assert np.isfinite(outs[SP][lvl]).all()
        assert e6 <= max(3.0 * e32, 2e-6)


@pytest.mark.parametrize("form", ["relu", "relu+residual", "relu+splitk"])
def test_f16x3_overflow_survives_the_fused_relu(hip, form):
    """An activation beyond the f16x3 range must come back NON-FINITE through the fused ReLU / residual / split-K reduce epilogues
    too (the ReLU keeps NaN like torch's; fmaxf(NaN, 0) would have returned a plausible 0)."""
    if SP != "f16x3":
        pytest.skip("range limit of the f16x3 mode")
    Cin, Cout, T, H, W = 64, 128, 2, 9, 40
    V = T * H * W
    x, w = _rand((Cin, V), 61), _rand((Cout, Cin, 1, 1, 1), 62, 1.0 / 8)
    x[3, 100] = 3.0e5
    xd = dev(x)
    out = torch.zeros(Cout, V, device="cuda")
    epi = dict(precision="f16x3", relu=1)
    if "residual" in form:
        rd = dev(_rand((Cout, V), 63))
        epi.update(residual=rd, res_strides=(V, 0, 0))
    scratch = torch.zeros(16 * Cout * V, device="cuda") if "splitk" in form else None
    hip.conv3d(hip.flat_volume(xd), hip.pack_conv_weight_any(dev(w), "f16x3"), dev(_rand((Cout,), 64)), hip.flat_volume(out), 1, 0, scratch, epi)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert not np.isfinite(o[:, 100]).any(), "the overflow was flushed to a finite value by the %s epilogue" % form
    assert np.isfinite(np.delete(o, 100, axis=1)).all()


@pytest.mark.parametrize("case", [("k2", 256, 256, 8, 30, 54, "plain"), ("k2", 128, 128, 3, 60, 108, "relu+res"), ("k2", 256, 128, 2, 120, 216, "splitk"),
                                  ("k2", 64, 128, 5, 17, 23, "plain")])
def test_flat_split_tiles_vs_fp64(hip, case):
    """Flat (ragged-width) form of the big split-staged 2-D tile (tile_cfg 5; f16x3 only): 512 flat positions of the zero-haloed
    [T][H + 2][pitch] run per workgroup, across the frames, so that maps whose width / height waste a
    32-column x 16-row tile (54 x 30: 21 %) compute the halo columns instead.  Same operands, same products: fp32-level error vs fp64,
    bit-identical run to run, identical summation to the 2-D tile wherever no split-K is involved."""
    if SP != "f16x3":
        pytest.skip("flat tiles exist for the default mode")
    kind, Cin, Cout, T, H, W, form = case
    kt = 3 if kind == "k3" else 1
    x, w, b = _rand((Cin, T, H, W), 5), _rand((Cout, Cin, kt, 3, 3), 6, 1.0 / np.sqrt(Cin * 9 * kt)), _rand((Cout,), 7)
    ref = _ref64(x, w, b, kt)
    buf, vin = _haloed(hip, x, kt)
    pw = hip.pack_conv_weight_any(dev(w), SP)
    outs = {}
    for cfg in (5, 1):
        out = torch.full((Cout, T, H, W), float("nan"), device="cuda")
        epi = dict(precision=SP)
        r = None
        if form == "relu+res":
            r = _rand((Cout, T, H, W), 8)
            rd = dev(r)
            epi.update(relu=1, residual=rd, res_strides=(T * H * W, H * W, W))
        scratch = torch.full((8 * Cout * T * H * W,), float("nan"), device="cuda") if form == "splitk" else None
        if form == "gn":
            stats = hip.conv3d_gn(vin, pw, dev(b), hip.dense_volume(out), 3, 32, 1e-5, cfg, None, SP)
            o64 = None
        else:
            hip.conv3d(vin, pw, dev(b), hip.dense_volume(out), (kt, 3, 3), cfg, scratch, epi)
        torch.cuda.synchronize()
        outs[cfg] = out.cpu().numpy().astype(np.float64)
        if form == "gn":
            o = outs[cfg].reshape(32, -1)
            st = stats.cpu().numpy().reshape(32, 2)
            assert np.abs(st[:, 0] - o.mean(1)).max() <= 2e-6 * max(1.0, np.abs(o.mean(1)).max())
            assert np.abs(st[:, 1] - 1.0 / np.sqrt(o.var(1) + 1e-5)).max() <= 1e-4 * (1.0 / np.sqrt(o.var(1) + 1e-5)).max()
        if cfg == 5:                                         # run-to-run identical
            out2 = torch.full((Cout, T, H, W), float("nan"), device="cuda")
            if form == "gn":
                hip.conv3d_gn(vin, pw, dev(b), hip.dense_volume(out2), 3, 32, 1e-5, cfg, None, SP)
            else:
                hip.conv3d(vin, pw, dev(b), hip.dense_volume(out2), (kt, 3, 3), cfg, scratch, epi)
            torch.cuda.synchronize()
            assert torch.equal(out, out2)
    want = ref if form != "relu+res" else np.maximum(ref + r, 0.0)
    e_flat, e_2d = np.abs(outs[5] - want).max(), np.abs(outs[1] - want).max()
    print("[f16x3] flat %s: max|err| vs fp64 flat %.3e, 2-D tile %.3e (max|ref| %.3g)" % (case, e_flat, e_2d, np.abs(want).max()))
    assert np.isfinite(outs[5]).all() and e_flat <= max(3.0 * e_2d, 4e-7 * np.abs(want).max())
    if form in ("plain", "relu+res", "gn"):
        assert np.array_equal(outs[5], outs[1]), "same chunk order, same products: the flat tile must reproduce the 2-D tile bit for bit"


@pytest.mark.parametrize("case", [("k3", 16, 128, 3, 40, 48, "plain"), ("k3", 8, 128, 2, 120, 216, "gn"), ("k3", 12, 256, 2, 37, 50, "relu+res"),
                                  ("k2", 32, 128, 3, 120, 216, "plain"), ("k2", 16, 256, 2, 23, 29, "relu+res"), ("k3", 8, 128, 4, 20, 24, "splitk")])
def test_block_tiles_vs_fp64(hip, case):
    """Block tiles (tile_cfg 6; f16x3 only): a workgroup owns 20 rows x 24 columns of the map as 5 x 3 MFMA column blocks of 4 rows x 8 columns
    (15 blocks: the eighth wave carries one instead of two), so that a 120 x 216 map is tiled with no junk position.  Same operands, same k
    order per output as the 16-row x 32-column tile: bit-identical to it (ragged maps, residual / ReLU, fused GroupNorm statistics included),
    fp32-level error vs fp64, identical run to run."""
    if SP != "f16x3":
        pytest.skip("block tiles exist for the default mode")
    kind, Cin, Cout, T, H, W, form = case
    kt = 3 if kind == "k3" else 1
    x, w, b = _rand((Cin, T, H, W), 15), _rand((Cout, Cin, kt, 3, 3), 16, 1.0 / np.sqrt(Cin * 9 * kt)), _rand((Cout,), 17)
    ref = _ref64(x, w, b, kt)
    buf, vin = _haloed(hip, x, kt)
    pw = hip.pack_conv_weight_any(dev(w), SP)
    outs, stats = {}, {}
    r = _rand((Cout, T, H, W), 18) if form == "relu+res" else None
    for cfg in (6, 1, 6):
        out = torch.full((Cout, T, H, W), float("nan"), device="cuda")
        epi = dict(precision=SP)
        if form == "relu+res":
            rd = dev(r)
            epi.update(relu=1, residual=rd, res_strides=(T * H * W, H * W, W))
        scratch = torch.full((8 * Cout * T * H * W,), float("nan"), device="cuda") if form == "splitk" else None
        if form == "gn":
            st = hip.conv3d_gn(vin, pw, dev(b), hip.dense_volume(out), 3, 32, 1e-5, cfg, None, SP)
            stats.setdefault(cfg, []).append(st.cpu().numpy())
        else:
            hip.conv3d(vin, pw, dev(b), hip.dense_volume(out), (kt, 3, 3), cfg, scratch, epi)
        torch.cuda.synchronize()
        outs.setdefault(cfg, []).append(out.cpu().numpy().astype(np.float64))
    want = ref if form != "relu+res" else np.maximum(ref + r, 0.0)
    e_blk, e_2d = np.abs(outs[6][0] - want).max(), np.abs(outs[1][0] - want).max()
    print("[f16x3] block tile %s: max|err| vs fp64 %.3e, 16 x 32 tile %.3e (max|ref| %.3g)" % (case, e_blk, e_2d, np.abs(want).max()))
    assert np.isfinite(outs[6][0]).all() and e_blk <= max(3.0 * e_2d, 4e-7 * np.abs(want).max())
    assert np.array_equal(outs[6][0], outs[6][1]), "run to run"
    if kind == "k3" and form in ("plain", "relu+res"):
        # the medium tile's block form (tile_cfg 7: 4 rows x 56 columns as 1 x 7 blocks, the eighth block missing): the same bits again
        out7 = torch.full((Cout, T, H, W), float("nan"), device="cuda")
        epi7 = dict(precision=SP)
        if form == "relu+res":
            epi7.update(relu=1, residual=dev(r), res_strides=(T * H * W, H * W, W))
        hip.conv3d(vin, pw, dev(b), hip.dense_volume(out7), (kt, 3, 3), 7, None, epi7)
        torch.cuda.synchronize()
        assert np.array_equal(out7.cpu().numpy().astype(np.float64), outs[1][0]), "4 x 56 block tile vs the 16 x 32 tile"
    if form != "splitk":
        assert np.array_equal(outs[6][0], outs[1][0]), "same chunk order, same products: the block tile must reproduce the 16 x 32 tile bit for bit"
    if form == "gn":
        o = outs[6][0].reshape(32, -1)
        assert np.array_equal(stats[6][0], stats[6][1])
        assert np.abs(stats[6][0].reshape(32, 2)[:, 0] - o.mean(1)).max() <= 2e-6 * max(1.0, np.abs(o.mean(1)).max())
