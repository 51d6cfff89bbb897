"""The arithmetic of the f16x3 convolution mode (csrc/conv_igemm.hip: split3 / pack_conv_weight_f16x3_kernel / the mm() terms), restated in
numpy and held against fp64 -- CPU only, no library needed.  It pins the scheme the kernel implements: every output channel's weights
scaled by S[co] = 2^(13 - floor(log2 max|w[co]|)), activations by 2^-2, two fp16 terms each, the low activation term stored as lo * 2^11 against
hi_w * 2^-11, three products, one exact rescale of the sum.  The GPU tests (tests/test_gpu_bf16x6.py) measure the kernel itself."""
import numpy as np
import pytest

ACT_SCALE = 0.25            # STEMSEG_F16X3_ACT_SCALE


def f16(a):
    return a.astype(np.float16).astype(np.float32)


def weight_scale(w):
    """[M, K] -> [M, 1]: one power of two per output channel (row)."""
    bits = np.abs(w).max(1).astype(np.float32).view(np.uint32)
    e = ((bits >> 23) & 0xff).astype(np.int64)
    s = np.where((e == 0) | (e == 0xff), 1.0, 2.0 ** (13 - (e - 127).astype(np.float64)))
    return s.astype(np.float32)[:, None]


def f16x3_matmul(w, x):
    """[M,K] x [K,N] the way the kernel forms it; products and sums in fp64 so that only the SPLIT's error is measured (the kernel
    accumulates in fp32 like the fp32-input MFMA kernel it is compared with)."""
    S = weight_scale(w)
    ws = w * S
    wh = f16(ws)
    wl = f16(ws - wh)
    whs = f16(wh * np.float32(2.0 ** -11))                  # the third packed weight plane
    xs = x * np.float32(ACT_SCALE)
    xh = f16(xs)
    xl = f16((xs - xh) * np.float32(2048.0))
    d = lambda a, b: a.astype(np.float64) @ b.astype(np.float64)
    return (d(wl, xh) + d(whs, xl) + d(wh, xh)) / (S.astype(np.float64) * ACT_SCALE), (wh, wl, whs, xh, xl)


@pytest.mark.parametrize("K", [256, 6912])
@pytest.mark.parametrize("wstd", [1e-4, 0.02, 5.0])
@pytest.mark.parametrize("xscale", [1e-4, 1e-2, 1.0, 100.0, 1e4, 5e4])
def test_split_error_is_below_fp32_summation_spread(K, wstd, xscale):
    rng = np.random.default_rng(K + int(1e6 * wstd) + int(xscale))
    w = (rng.standard_normal((32, K)) * wstd).astype(np.float32)
    x = (np.maximum(rng.standard_normal((K, 64)), 0) * xscale).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(ref).max()
    y, planes = f16x3_matmul(w, x)
    assert all(np.isfinite(p).all() for p in planes)
    err = np.abs(y - ref).max() / scale
    err32 = np.abs((w @ x).astype(np.float64) - ref).max() / scale         # one fp32 summation order (BLAS)
    # 22 significand bits per operand: <= 2^-21 of the output scale, and in practice below what fp32 accumulation itself adds
    assert err <= 5e-7, (err, err32)
    if xscale >= 1e-2:
        assert err <= 2.0e-7, (err, err32)


def test_the_pair_keeps_22_bits_over_the_documented_range():
    tiny = np.float32(2.0 ** -14)                              # smallest normal fp16
    for a in (2.5e-4, 1.0, 4094.0, 2.6e5):
        xs = np.float32(a) * np.float32(ACT_SCALE) * np.float32(1.0 + 2.0 ** -12)       # (an inexact value: the low term is non-zero)
        xh = f16(np.array([xs]))[0]
        xl = f16(np.array([(xs - xh) * np.float32(2048.0)]))[0]
        assert np.isfinite(xh) and abs(xh) >= tiny
        assert abs((xh + xl / 2048.0) - xs) <= 2.0 ** -21 * abs(xs)
    w = np.array([[0.03, -0.02, 0.03 * 2.0 ** -15]], np.float32) * np.float32(1.0 + 2.0 ** -12)
    S = weight_scale(w)
    wh = f16(w * S)
    assert (np.abs(wh) >= tiny).all() and (np.abs(f16(wh * np.float32(2.0 ** -11))) >= tiny).all()
    assert 2.0 ** 13 <= np.abs(w * S).max() < 2.0 ** 14


def test_per_channel_scale_keeps_fp32_level_error_under_folded_bn_statistics():
    """FrozenBN folded with eps = 0 (make_layers.py:51-63) multiplies every output channel by gamma / sqrt(var): per-channel weight
    magnitudes log-uniform over 2^-20 ... 1.  With one scale per output channel every channel keeps the 22-bit pair; with one scale
    per LAYER (round 3) the small channels' low terms go subnormal and their error grows by orders of magnitude."""
    rng = np.random.default_rng(11)
    M, K, N = 64, 1024, 48
    ch = 2.0 ** rng.uniform(-20, 0, size=(M, 1))
    w = (rng.standard_normal((M, K)) * 0.05 * ch).astype(np.float32)
    x = (np.maximum(rng.standard_normal((K, N)), 0) * 30.0).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    row_scale = np.abs(ref).max(1, keepdims=True)
    y, planes = f16x3_matmul(w, x)
    assert all(np.isfinite(p).all() for p in planes)
    err = (np.abs(y - ref) / row_scale).max()                  # relative to each CHANNEL's own output scale
    assert err <= 2.0e-7, err
    # the per-layer form, for the record: same split with one scale for the whole matrix
    S = weight_scale(w).max() * 0 + weight_scale(w.reshape(1, -1))[0, 0]
    ws = w * S
    wh = f16(ws); wl = f16(ws - wh); whs = f16(wh * np.float32(2.0 ** -11))
    xs = x * np.float32(ACT_SCALE); xh = f16(xs); xl = f16((xs - xh) * np.float32(2048.0))
    d = lambda a, b: a.astype(np.float64) @ b.astype(np.float64)
    y_layer = (d(wl, xh) + d(whs, xl) + d(wh, xh)) / (float(S) * ACT_SCALE)
    err_layer = (np.abs(y_layer - ref) / row_scale).max()
    assert err_layer > 20 * err, (err_layer, err)


def test_out_of_range_activation_becomes_infinite_not_wrong():
    x = np.array([[3.0e5]], np.float32)
    _, (wh, wl, whs, xh, xl) = f16x3_matmul(np.array([[0.5]], np.float32), x)
    assert np.isinf(xh).all()
    ok, _ = f16x3_matmul(np.array([[0.5]], np.float32), np.array([[2.5e5]], np.float32))
    assert np.isfinite(ok).all() and abs(ok[0, 0] - 1.25e5) <= 1.25e5 * 2.0 ** -21
