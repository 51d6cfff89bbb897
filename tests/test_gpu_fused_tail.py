"""The fused bottleneck tail (csrc/bottleneck_fused.hip; StemsegEncoderDesc.fuse_tail): conv3 (+ bias + identity + ReLU) of block b and
conv1 (+ bias + ReLU) of block b + 1 in ONE back-to-back kernel, conv2 handing its output over as fp16 operand planes.
Reference: /root/reference/stemseg/modeling/backbone/resnet.py:262-282 of two consecutive blocks.

Stages 1-2 run the 32-column form, stage 3 (MID = 256) the one-wave-per-SIMD form: both issue v_mfma_f32_32x32x16_f16 on the same operands,
with the same split arithmetic and the same k order per accumulator as the separate launches => the encoder's four FPN maps must be
BIT-IDENTICAL with the fusion on and off wherever the separate launches run without split-K (the bench shape at its planning frame count).
The earlier stage-3 kernel, the 16-column form (fuse_tail bit 3; one v_mfma_f32_16x16x32_f16 sums a 32-channel chunk where the others issue
two 16-deep groups), is fp32 round-off apart, ONE result whatever the batch.  Everything within 1e-4 of the CPU oracle."""
import numpy as np
import pytest
import torch

from tests import synth
from oracle import encoder as oenc

pytestmark = pytest.mark.gpu
MEAN = torch.tensor([102.9801, 115.9465, 122.7717])[None, :, None, None]


@pytest.fixture(scope="module")
def hip():
    from stemseg_amd import hip as h
    h.require_gpu()
    return h


def _backbone(name, seed):
    from stemseg_amd.modeling.backbone import ResNetFPN
    bb = ResNetFPN(name).eval()
    sd = synth.synth_state_dict([(k, v.shape) for k, v in bb.state_dict().items()], seed, prefix="backbone.")
    bb.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(bb.state_dict()[k].shape) for k, v in sd.items()})
    return bb.cuda(), sd


def _run(hip, bb, x, fuse, precision="f16x3"):
    bb.fuse_tail, bb.precision = fuse, precision
    T, _, H, W = x.shape
    outs = [torch.full((256, T, H // s, W // s), float("nan"), device="cuda") for s in (4, 8, 16, 32)]
    hip.profile_enable(True)
    hip.profile_read()
    bb.run_backbone_into(x, [hip.dense_volume(o) for o in outs])
    prof = hip.profile_read()
    hip.profile_enable(False)
    return outs, (prof.get(19, (0, 0, 0))[2])


R1 = 7 | 16          # fuse_tail bits 3-4 = 2: stage 3 on the one-wave-per-SIMD form (fused_tail_r1_kernel; also the default)
W16 = 7 | 8          # ... = 1: stage 3 on the 16-column form (fused_tail16_kernel)


def test_fused_tail_vs_the_three_launch_blocks_at_the_bench_shape(hip):
    """480 x 864, R-101, 8 frames under the 32-frame plan (what a bench step runs).  Stages 1-2 fused (mask 3): 2 + 3 launches; all three stages
    (mask 7, the default): 2 + 3 + 22 launches -- every FPN map torch.equal to the un-fused encoder's; the SAME bits whether the 8 frames pass
    alone or as the second clip of a 16-frame pass (batch invariance of the fused path).  The 16-column form of stage 3: fp32 round-off apart."""
    bb, _ = _backbone("R-101-FPN", 71)
    x = (torch.from_numpy(synth.synth_frames(8, 480, 864, seed=71).astype(np.float32)).permute(0, 3, 1, 2) - MEAN).cuda()
    ref, n0 = _run(hip, bb, x, False)
    got3, n3 = _run(hip, bb, x, 3)
    got, n1 = _run(hip, bb, x, True)
    got_r1, n2 = _run(hip, bb, x, R1)
    got16, n16 = _run(hip, bb, x, W16)
    assert n0 == 0 and n3 == 2 + 3 and n1 == n2 == n16 == 2 + 3 + 22, (n0, n3, n1, n2, n16)
    for r, g3, g, g1, g16, s in zip(ref, got3, got, got_r1, got16, (4, 8, 16, 32)):
        assert torch.isfinite(g).all()
        assert torch.equal(r, g3), "mask 3, 1/%d: %d of %d values differ, max %g" % (s, int((r != g3).sum()), r.numel(), float((r - g3).abs().max()))
        assert torch.equal(r, g), "mask 7, 1/%d: %d of %d values differ, max %g" % (s, int((r != g).sum()), r.numel(), float((r - g).abs().max()))
        assert torch.equal(g, g1)
        scale = max(1.0, float(r.abs().max()))
        err = float((r - g16).abs().max()) / scale
        print("[fused] 1/%d: stage-3 16-column form vs separate launches: max |diff| / scale = %.3g" % (s, err))
        assert err <= 1e-5
    # a second pass of other frames through the same workspaces (no state left behind in the operand planes), alone and behind another clip
    x2 = (torch.from_numpy(synth.synth_frames(8, 480, 864, seed=72).astype(np.float32)).permute(0, 3, 1, 2) - MEAN).cuda()
    for mask, first in ((True, got), (W16, got16)):
        alone, _ = _run(hip, bb, x2, mask)
        both, _ = _run(hip, bb, torch.cat([x, x2], 0), mask)
        for a, b, g in zip(alone, both, first):
            assert torch.equal(a, b[:, 8:]) and torch.equal(g, b[:, :8])
    assert bb.check_workspaces()[0] == 0


@pytest.mark.parametrize("shape", [(3, 96, 160), (5, 128, 224), (2, 480, 864)])
def test_fused_tail_vs_oracle_on_ragged_sizes(hip, shape):
    """Frame counts / map sizes whose position count is not a multiple of the 256-position tile, R-50, vs the CPU oracle (and vs the
    un-fused path to fp32 round-off: at these sizes the separate launches may split K, i.e. sum in another order)."""
    T, H, W = shape
    bb, sd = _backbone("R-50-FPN", 73)
    x = torch.from_numpy(synth.synth_frames(T, H, W, seed=73).astype(np.float32)).permute(0, 3, 1, 2) - MEAN
    ref = oenc.resnet_fpn(x, {"backbone." + k: v for k, v in sd.items()}, "R-50-FPN")
    got, n1 = _run(hip, bb, x.cuda(), True)
    unf, _ = _run(hip, bb, x.cuda(), False)
    print("[fused] %s: %d fused launches" % (shape, n1))
    for g, u, s in zip(got, unf, (4, 8, 16, 32)):
        r = ref[s].permute(1, 0, 2, 3).numpy()
        scale = max(1.0, float(np.abs(r).max()))
        assert float(np.abs(g.cpu().numpy() - r).max()) / scale <= 1e-4
        assert float((g - u).abs().max()) / scale <= 2e-5
    assert bb.check_workspaces()[0] == 0


def test_fused_tail_only_in_f16x3(hip):
    bb, _ = _backbone("R-50-FPN", 74)
    x = (torch.from_numpy(synth.synth_frames(2, 96, 160, seed=74).astype(np.float32)).permute(0, 3, 1, 2) - MEAN).cuda()
    for prec in ("bf16x6", "f32"):
        _, n = _run(hip, bb, x, True, prec)
        assert n == 0


@pytest.mark.parametrize("shape", [(3, 96, 160), (5, 128, 224), (1, 480, 864)])
def test_one_wave_per_simd_form_on_ragged_sizes(hip, shape):
    """Position counts that are not a multiple of the 128-position workgroup (columns past V are computed and dropped by the buffer
    descriptors' range check): vs the CPU oracle, and no write outside the workspace slices."""
    T, H, W = shape
    bb, sd = _backbone("R-50-FPN", 73)
    x = torch.from_numpy(synth.synth_frames(T, H, W, seed=73).astype(np.float32)).permute(0, 3, 1, 2) - MEAN
    ref = oenc.resnet_fpn(x, {"backbone." + k: v for k, v in sd.items()}, "R-50-FPN")
    got, n1 = _run(hip, bb, x.cuda(), R1)
    w16, _ = _run(hip, bb, x.cuda(), W16)
    for g, u, s in zip(got, w16, (4, 8, 16, 32)):
        r = ref[s].permute(1, 0, 2, 3).numpy()
        scale = max(1.0, float(np.abs(r).max()))
        assert float(np.abs(g.cpu().numpy() - r).max()) / scale <= 1e-4
        assert float((g - u).abs().max()) / scale <= 2e-5
    assert bb.check_workspaces()[0] == 0
