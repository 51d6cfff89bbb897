"""2-D encoder: ResNet-50/101 (FrozenBN, stride in the first 1x1) + FPN, batched over the clip's frames.

Counterpart of the reference's stemseg/modeling/backbone/* (ResNet resnet.py:49-113, Bottleneck :194-282,
BaseStem :285-304, FPN fpn.py:8-69, FrozenBatchNorm2d make_layers.py:37-63) and of
TrainingModel.run_backbone (model_builder.py:154-169).  State-dict keys follow the reference
(``body.stem.conv1.weight``, ``body.layerL.B.{conv1,bn1,...}``, ``fpn.fpn_{inner,layer}K.{weight,bias}``) so real
checkpoints load.

Round-1 status: the convolutions of this stage still run on stock
PyTorch-ROCm ops (MIOpen); what is already MI355X-specific is the schedule -- all T frames of a clip go through
as ONE batch (the reference runs eight batch-1 passes, inference_model.py:99-102) and every FrozenBN (eps = 0,
make_layers.py:43) is folded into its convolution's weight/bias once at load time, so no BN kernels run.
Replacing these convs with the implicit-GEMM MFMA kernel of csrc/conv_igemm.hip is the next step.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.global_registry import GlobalRegistry

STAGE_BLOCKS = {"R-50-FPN": (3, 4, 6, 3), "R-101-FPN": (3, 4, 23, 3)}


class FrozenBatchNorm2d(nn.Module):
    """Per-channel affine with fixed statistics (make_layers.py:37-63); holds buffers, folded away at run time."""

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

    def forward(self, x):
        s, b = self.scale_shift()
        return x * s.reshape(1, -1, 1, 1) + b.reshape(1, -1, 1, 1)


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
    """``forward([N,3,H,W]) -> tuple of 4 maps, highest resolution first`` (fpn.py:67-69)."""

    def __init__(self, backbone_type="R-101-FPN", out_channels=256):
        super().__init__()
        if backbone_type not in STAGE_BLOCKS:
            raise KeyError(backbone_type)       # "X-101-FPN" is registered but has no stage spec (resnet.py:352-355)
        self.body = _Body(STAGE_BLOCKS[backbone_type])
        self.fpn = _FPN(out_channels)
        self.out_channels, self.is_3d = out_channels, False
        self._folded, self._sig = None, None
        self.channels_last = True

    # ---- FrozenBN folding: w' = w * s[:,None,None,None], b' = shift  (exact because eps == 0 changes nothing) ----
    def _signature(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    @staticmethod
    def _fold(conv, bn):
        s, b = bn.scale_shift()
        return (conv.weight * s.reshape(-1, 1, 1, 1)).detach(), b.detach()

    def _prepare(self):
        sig = self._signature()
        if self._sig == sig:
            return self._folded
        f = {}
        mf = torch.channels_last if self.channels_last else torch.contiguous_format

        def put(name, w, b):
            f[name] = (w.contiguous(memory_format=mf), None if b is None else b.contiguous())
        put("stem", *self._fold(self.body.stem.conv1, self.body.stem.bn1))
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self.body, "layer%d" % li)):
                p = "l%d.%d." % (li, bi)
                put(p + "1", *self._fold(blk.conv1, blk.bn1))
                put(p + "2", *self._fold(blk.conv2, blk.bn2))
                put(p + "3", *self._fold(blk.conv3, blk.bn3))
                if blk.downsample is not None:
                    put(p + "d", *self._fold(blk.downsample[0], blk.downsample[1]))
        for k in (1, 2, 3, 4):
            for kind in ("inner", "layer"):
                m = getattr(self.fpn, "fpn_%s%d" % (kind, k))
                put("fpn_%s%d" % (kind, k), m.weight.detach(), m.bias.detach())
        self._folded, self._sig = f, sig
        return f

    @torch.no_grad()
    def forward(self, x):
        f = self._prepare()
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        w, b = f["stem"]
        x = F.relu_(F.conv2d(x, w, b, stride=2, padding=3))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        feats = []
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self.body, "layer%d" % li)):
                p = "l%d.%d." % (li, bi)
                idt = x
                out = F.relu_(F.conv2d(x, *f[p + "1"], stride=blk.stride))
                out = F.relu_(F.conv2d(out, *f[p + "2"], padding=1))
                out = F.conv2d(out, *f[p + "3"])
                if blk.downsample is not None:
                    idt = F.conv2d(x, *f[p + "d"], stride=blk.stride)
                x = F.relu_(out.add_(idt))
            feats.append(x)
        last = F.conv2d(feats[3], *f["fpn_inner4"])
        results = [F.conv2d(last, *f["fpn_layer4"], padding=1)]
        for k in (3, 2, 1):
            top = F.interpolate(last, scale_factor=2, mode="bilinear", align_corners=False)
            last = F.conv2d(feats[k - 1], *f["fpn_inner%d" % k]).add_(top)
            results.insert(0, F.conv2d(last, *f["fpn_layer%d" % k], padding=1))
        return tuple(results)

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
