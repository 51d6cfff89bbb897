"""2-D encoder: ResNet-50/101 (FrozenBN, stride in the first 1x1) + FPN on the HIP implicit-GEMM kernels.

Counterpart of the reference's stemseg/modeling/backbone/* (ResNet resnet.py:49-113, Bottleneck :194-282,
BaseStem :285-304, FPN fpn.py:8-69, FrozenBatchNorm2d make_layers.py:37-63) and of
TrainingModel.run_backbone (model_builder.py:154-169).  State-dict keys follow the reference
(``body.stem.conv1.weight``, ``body.layerL.B.{conv1,bn1,...}``, ``fpn.fpn_{inner,layer}K.{weight,bias}``) so real
checkpoints load; the ``nn.Conv2d`` / ``FrozenBatchNorm2d`` objects are parameter holders only.

Execution (``stemseg_hip_encoder_forward``, csrc/encoder.hip): all T frames of a clip form the T axis of one
[C][T][H][W] volume, every FrozenBN (eps = 0, make_layers.py:43) is folded into its convolution once at load time, and
bias / ReLU / residual add run in the conv epilogue.  The four FPN maps come out channel-major ([256][T][h][w]) -- or
are written straight into the decoders' zero-haloed input buffers (``run_backbone_into``).
"""
import ctypes as C
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import hip
from ..utils.global_registry import GlobalRegistry

STAGE_BLOCKS = {"R-50-FPN": (3, 4, 6, 3), "R-101-FPN": (3, 4, 23, 3)}


class FrozenBatchNorm2d(nn.Module):
    """Per-channel affine with fixed statistics (make_layers.py:37-63); holds buffers, folded away at load time."""

    def __init__(self, n, epsilon=0.0):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.epsilon = epsilon

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.epsilon).rsqrt()
        return scale, self.bias - self.running_mean * scale


def _conv(cin, cout, k, stride=1, bias=False):
    m = nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=bias)
    nn.init.kaiming_uniform_(m.weight, a=1)
    if bias:
        nn.init.constant_(m.bias, 0)
    return m


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride), FrozenBatchNorm2d(cout))
        self.conv1, self.bn1 = _conv(cin, mid, 1, stride), FrozenBatchNorm2d(mid)      # STRIDE_IN_1X1 (defaults.yaml:55)
        self.conv2, self.bn2 = _conv(mid, mid, 3), FrozenBatchNorm2d(mid)
        self.conv3, self.bn3 = _conv(mid, cout, 1), FrozenBatchNorm2d(cout)
        self.stride = stride


class _Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        nn.init.kaiming_uniform_(self.conv1.weight, a=1)
        self.bn1 = FrozenBatchNorm2d(64)


class _Body(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.stem = _Stem()
        cin = 64
        for li, n in enumerate(blocks, 1):
            mid, cout = 64 * 2 ** (li - 1), 256 * 2 ** (li - 1)
            layer = []
            for bi in range(n):
                layer.append(_Bottleneck(cin, mid, cout, 2 if (bi == 0 and li > 1) else 1))
                cin = cout
            setattr(self, "layer%d" % li, nn.Sequential(*layer))


class _FPN(nn.Module):
    def __init__(self, out_channels=256):
        super().__init__()
        for k in (1, 2, 3, 4):
            setattr(self, "fpn_inner%d" % k, _conv(256 * 2 ** (k - 1), out_channels, 1, bias=True))
            setattr(self, "fpn_layer%d" % k, _conv(out_channels, out_channels, 3, bias=True))


class ResNetFPN(nn.Module):
    """``forward([N,3,H,W]) -> tuple of 4 maps [N,256,H/s,W/s], highest resolution first`` (fpn.py:67-69)."""

    def __init__(self, backbone_type="R-101-FPN", out_channels=256):
        super().__init__()
        if backbone_type not in STAGE_BLOCKS:
            raise KeyError(backbone_type)       # "X-101-FPN" is registered but has no stage spec (resnet.py:352-355)
        self.stage_blocks = STAGE_BLOCKS[backbone_type]
        self.body = _Body(self.stage_blocks)
        self.fpn = _FPN(out_channels)
        self.out_channels, self.is_3d = out_channels, False
        self._packed, self._ws, self._ws_desc = {}, {}, {}     # precision -> (parameter signature, packed weights); workspaces (+ their descriptors) by shape / lane
        self.precision = hip.DEFAULT_PRECISION    # hip.PRECISIONS: "f16x3" | "bf16x6" | "f32"
        self.lane = 0             # selects one of several independent workspaces (one per in-flight step / stream)
        # Every convolution of a pass decides its tile shape and split-K factor as if the pass held this many frames
        # (StemsegEncoderDesc.plan_frames): a frame's maps are then bit-identical whether it passes alone, in a batch of clips or in
        # the union of overlapping windows -- the K-partition of a split-K launch is a summation order, and the clusterer downstream is
        # a chain of hard thresholds (clusterers.py:106-146).  32 = four 8-frame clips per pass, the shape bench.py and
        # ClipPipeline.embed_many run; a job whose ranks must agree bit for bit keeps ONE value on all of them.  0: plan every pass on
        # its own frame count (fastest for a lone small pass; results then depend on the batch).
        self.plan_frames = 32
        # f16x3: conv3 (+ identity + ReLU) of a bottleneck block and conv1 of the next in one back-to-back kernel (StemsegEncoderDesc.fuse_tail):
        # bit-identical to the separate launches wherever those run un-split; True / 7: stages 1-3, a bit mask (1, 2, 4) picks stages,
        # False / 0: three launches per block everywhere (A/B, tests)
        self.fuse_tail = True
        self.stem_s2d = True             # f16x3 mode: the stem as space-to-depth + 4x4 conv on the split-staged MFMA kernel (False: the exact fp32-MFMA stem; A/B)

    # ---- FrozenBN folding: w' = w * scale[:, None, None, None], b' = shift (exact: eps == 0) -------------------
    def _signature(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    @staticmethod
    def _fold(conv, bn):
        s, b = bn.scale_shift()
        return (conv.weight * s.reshape(-1, 1, 1, 1)).detach().float(), b.detach().float()

    def blocks(self):
        for li in (1, 2, 3, 4):
            for blk in getattr(self.body, "layer%d" % li):
                yield blk

    @torch.no_grad()
    def folded_state(self):
        """{name: (weight [Cout,Cin,kh,kw], bias [Cout])} with every FrozenBN folded in; plain tensor math, no conv."""
        f = OrderedDict()
        f["stem"] = self._fold(self.body.stem.conv1, self.body.stem.bn1)
        for i, blk in enumerate(self.blocks()):
            f["b%d.conv1" % i] = self._fold(blk.conv1, blk.bn1)
            f["b%d.conv2" % i] = self._fold(blk.conv2, blk.bn2)
            f["b%d.conv3" % i] = self._fold(blk.conv3, blk.bn3)
            if blk.downsample is not None:
                f["b%d.down" % i] = self._fold(blk.downsample[0], blk.downsample[1])
        for k in (1, 2, 3, 4):
            for kind in ("inner", "layer"):
                m = getattr(self.fpn, "fpn_%s%d" % (kind, k))
                f["fpn_%s%d" % (kind, k)] = (m.weight.detach().float(), m.bias.detach().float())
        return f

    def _pack(self):
        sig = (self._signature(), bool(self.stem_s2d))
        hit = self._packed.get(self.precision)
        if hit is not None and hit[0] == sig:      # (one packing per precision: an overflow re-run in bf16x6 keeps the f16x3 one)
            return hit[1]
        hip.require_gpu()
        f = self.folded_state()
        keep = []                           # device tensors referenced by raw pointers below
        w = hip.EncoderWeights()

        def dev(t):
            t = t.contiguous().cuda()
            keep.append(t)
            return t

        def packed(name):
            wt, b = f[name]
            pw, pb = hip.pack_conv_weight_any(dev(wt), self.precision), dev(b)
            keep.append(pw)
            return pw.data_ptr(), pb.data_ptr()
        sw, sb = f["stem"]
        w.stem_w = dev(sw.reshape(64, 147).t()).data_ptr()
        w.stem_b = dev(sb).data_ptr()
        if self.precision == "f16x3" and self.stem_s2d:
            # the stem as a stride-1 4x4 convolution over the space-to-depth image (csrc/encoder.hip, stem_s2d_kernel):
            # W2[co][(p*2+q)*3+c][a][b] = w[co][c][2a+p-1][2b+q-1], zero where an index is -1
            w8 = torch.zeros(64, 3, 8, 8, dtype=torch.float32, device=sw.device)
            w8[:, :, 1:, 1:] = sw.reshape(64, 3, 7, 7)
            w2 = w8.reshape(64, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(64, 12, 1, 4, 4)      # [co][p][q][c][a][b]
            pw2 = hip.pack_conv_weight_any(dev(w2), "f16x3")
            keep.append(pw2)
            w.stem_w_s2d = pw2.data_ptr()
        for i, blk in enumerate(self.blocks()):
            w.conv1_w[i], w.conv1_b[i] = packed("b%d.conv1" % i)
            w.conv2_w[i], w.conv2_b[i] = packed("b%d.conv2" % i)
            w.conv3_w[i], w.conv3_b[i] = packed("b%d.conv3" % i)
            if blk.downsample is not None:
                w.down_w[i], w.down_b[i] = packed("b%d.down" % i)
        for k in (1, 2, 3, 4):
            w.fpn_inner_w[k - 1], w.fpn_inner_b[k - 1] = packed("fpn_inner%d" % k)
            w.fpn_layer_w[k - 1], w.fpn_layer_b[k - 1] = packed("fpn_layer%d" % k)
        self._packed[self.precision] = (sig, (w, keep))
        return w, keep

    def _desc(self, T, H, W, n_clips=1, clip_frames=0, clip_stride=0):
        d = hip.EncoderDesc()
        d.struct_bytes = C.sizeof(hip.EncoderDesc)
        for i, n in enumerate(self.stage_blocks):
            d.blocks[i] = n
        d.T, d.H, d.W, d.out_channels = T, H, W, self.out_channels
        d.precision = hip.PRECISIONS[self.precision]
        d.n_clips, d.clip_frames, d.clip_stride = n_clips, clip_frames, clip_stride
        d.plan_frames = int(self.plan_frames)
        d.fuse_tail = 7 if self.fuse_tail is True else int(self.fuse_tail)      # (bit s - 1: stage s)
        return d

    @torch.no_grad()
    def run_backbone_into(self, frames, out_volumes, window=None):
        """frames: float32 [T,3,H,W] on the device; out_volumes: 4 ``hip.Volume`` (4x, 8x, 16x, 32x), each
        [256][T][H/s][W/s] -- e.g. the interiors of the decoders' zero-haloed inputs.  With 4 * n volumes the T frames are n
        consecutive clips of T / n frames sharing one encoder pass; volumes 4c .. 4c+3 receive clip c.  ``window`` =
        (clip_frames, clip_stride): the n clips are overlapping windows of the pass (shared frames go through the trunk once)."""
        hip.require_gpu()
        frames = frames.contiguous().float()
        T, _, H, W = frames.shape
        w, _keep = self._pack()
        n = len(out_volumes) // 4
        assert len(out_volumes) % 4 == 0
        if window is None:
            assert T % n == 0
            d = self._desc(T, H, W, n)
        else:
            assert (n - 1) * window[1] + window[0] == T, "windows do not cover the pass"
            d = self._desc(T, H, W, n, int(window[0]), int(window[1]))
        key = (T, H, W, frames.device.index, self.lane, None if window is None else (n, int(window[0]), int(window[1])), int(self.plan_frames))
        ws = self._ws.get(key)
        if ws is None:
            nbytes = hip.lib().stemseg_hip_encoder_workspace_bytes(C.byref(d))
            if nbytes == 0:
                raise RuntimeError("encoder: " + hip.lib().stemseg_hip_last_error().decode())
            ws = torch.empty(nbytes, dtype=torch.uint8, device=frames.device)
            hip.check(hip.lib().stemseg_hip_encoder_init_workspace(C.byref(d), hip.ptr(ws), nbytes, hip.stream()))
            self._ws[key] = ws
            self._ws_desc[key] = d
        vols = (hip.Volume * len(out_volumes))(*out_volumes)
        hip.check(hip.lib().stemseg_hip_encoder_forward(C.byref(d), C.byref(w), hip.ptr(frames), vols, hip.ptr(ws), ws.numel(), hip.stream()))

    def check_workspaces(self):
        """Debug check (synchronises): clobbered guard words over all cached workspaces -- 0 unless some kernel wrote outside its
        slice (stemseg_hip_encoder_check_workspace).  -> (n_bad, [(workspace key, first bad float offset)])"""
        bad, where = 0, []
        for key, ws in self._ws.items():
            n, first = C.c_int32(0), C.c_int64(-1)
            with torch.cuda.device(ws.device):
                hip.check(hip.lib().stemseg_hip_encoder_check_workspace(C.byref(self._ws_desc[key]), hip.ptr(ws), ws.numel(), C.byref(n), C.byref(first), hip.stream()))
            if n.value:
                bad += n.value
                where.append((key, first.value))
        return bad, where

    @torch.no_grad()
    def forward_channel_major(self, frames):
        """-> 4 dense tensors [256, T, H/s, W/s] (s = 4, 8, 16, 32)."""
        T, _, H, W = frames.shape
        outs = [torch.empty(self.out_channels, T, H // s, W // s, dtype=torch.float32, device=frames.device) for s in (4, 8, 16, 32)]
        self.run_backbone_into(frames, [hip.dense_volume(o) for o in outs])
        return outs

    @torch.no_grad()
    def forward(self, x):
        return tuple(o.permute(1, 0, 2, 3) for o in self.forward_channel_major(x))

    @torch.no_grad()
    def run_backbone(self, frames):
        """frames [T,3,H,W] -> OrderedDict {4,8,16,32: [T,256,H/s,W/s]} (model_builder.py:154-169)."""
        return OrderedDict(zip((4, 8, 16, 32), self.forward(frames)))


def build_resnet_fpn_backbone(cfg):
    return ResNetFPN(cfg.MODEL.BACKBONE.TYPE, cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS)


BACKBONE_REGISTRY = GlobalRegistry.get("Backbone")
BACKBONE_REGISTRY.add("R-50-FPN", build_resnet_fpn_backbone)
BACKBONE_REGISTRY.add("R-101-FPN", build_resnet_fpn_backbone)
BACKBONE_REGISTRY.add("X-101-FPN", build_resnet_fpn_backbone)
