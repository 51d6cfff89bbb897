"""The fused bottleneck tail (csrc/bottleneck_fused.hip; StemsegEncoderDesc.fuse_tail): conv3 (+ bias + identity + ReLU) of block b and
conv1 (+ bias + ReLU) of block b + 1 in ONE back-to-back kernel, conv2 handing its output over as fp16 operand planes.
Reference: /root/reference/stemseg/modeling/backbone/resnet.py:262-282 of two consecutive blocks.

Same operands, same split arithmetic, same k order per accumulator as the separate launches => the encoder's four FPN maps must be
BIT-IDENTICAL with the fusion on and off wherever the separate launches run without split-K (the bench shape at its planning frame
count), and within fp32 round-off of the oracle everywhere."""
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


def test_fused_tail_is_bit_identical_to_the_three_launch_blocks_at_the_bench_shape(hip):
    """480 x 864, R-101, 8 frames under the 32-frame plan (what a bench step runs): 2 + 3 + 22 fused launches (stages 1-3; stage 4 is
    not fused), every FPN map torch.equal to the un-fused encoder's."""
    bb, _ = _backbone("R-101-FPN", 71)
    x = (torch.from_numpy(synth.synth_frames(8, 480, 864, seed=71).astype(np.float32)).permute(0, 3, 1, 2) - MEAN).cuda()
    ref, n0 = _run(hip, bb, x, False)
    got, n1 = _run(hip, bb, x, True)
    assert n0 == 0 and n1 == 2 + 3 + 22, (n0, n1)
    for r, g, s in zip(ref, got, (4, 8, 16, 32)):
        assert torch.isfinite(g).all()
        assert torch.equal(r, g), "1/%d: %d of %d values differ, max %g" % (s, int((r != g).sum()), r.numel(), float((r - g).abs().max()))
    # a second pass of other frames through the same workspaces: still identical (no state left behind in the operand planes)
    x2 = (torch.from_numpy(synth.synth_frames(8, 480, 864, seed=72).astype(np.float32)).permute(0, 3, 1, 2) - MEAN).cuda()
    got2, _ = _run(hip, bb, x2, True)
    ref2, _ = _run(hip, bb, x2, False)
    assert all(torch.equal(a, b) for a, b in zip(ref2, got2))
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
