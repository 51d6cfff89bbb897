"""Oracle: ResNet-50/101 + FPN 2-D encoder on CPU (plain torch fp32 ops, FrozenBN un-folded).

Follows /root/reference/stemseg/modeling/backbone/resnet.py:49-113,194-304, fpn.py:47-69,
make_layers.py:37-63 (FrozenBatchNorm2d, eps = 0.0) and model_builder.py:154-169 (scale keys).
TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

STAGE_BLOCKS = {"R-50-FPN": (3, 4, 6, 3), "R-101-FPN": (3, 4, 23, 3)}   # resnet.py:38-46


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))


def _frozen_bn(x, sd, p):
    # make_layers.py:51-63 : scale = w * rsqrt(var + 0) ; bias = b - mean * scale
    scale = _t(sd[p + ".weight"]) * _t(sd[p + ".running_var"]).rsqrt()
    bias = _t(sd[p + ".bias"]) - _t(sd[p + ".running_mean"]) * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def _bottleneck(x, sd, p, stride, has_down):
    # resnet.py:263-282 with STRIDE_IN_1X1 (defaults.yaml:55): stride sits on conv1 and the shortcut
    idt = x
    out = F.relu(_frozen_bn(F.conv2d(x, _t(sd[p + ".conv1.weight"]), stride=stride), sd, p + ".bn1"))
    out = F.relu(_frozen_bn(F.conv2d(out, _t(sd[p + ".conv2.weight"]), padding=1), sd, p + ".bn2"))
    out = _frozen_bn(F.conv2d(out, _t(sd[p + ".conv3.weight"])), sd, p + ".bn3")
    if has_down:
        idt = _frozen_bn(F.conv2d(x, _t(sd[p + ".downsample.0.weight"]), stride=stride), sd, p + ".downsample.1")
    return F.relu(out + idt)


@torch.no_grad()
def resnet_fpn(images, sd, backbone_type="R-101-FPN", prefix="backbone."):
    """images [N,3,H,W] (BGR, mean-subtracted) -> dict {4,8,16,32: [N,256,H/s,W/s]}."""
    x = _t(images).float()
    b = prefix + "body."
    x = F.conv2d(x, _t(sd[b + "stem.conv1.weight"]), stride=2, padding=3)           # resnet.py:292-304
    x = F.relu(_frozen_bn(x, sd, b + "stem.bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = []
    for li, nblocks in enumerate(STAGE_BLOCKS[backbone_type], 1):
        for bi in range(nblocks):
            stride = 2 if (bi == 0 and li > 1) else 1
            x = _bottleneck(x, sd, b + "layer%d.%d" % (li, bi), stride, bi == 0)
        feats.append(x)
    f = prefix + "fpn."
    # fpn.py:55-69 top-down: inner 1x1, bilinear x2 of the coarser map, add, 3x3 output conv
    last = F.conv2d(feats[3], _t(sd[f + "fpn_inner4.weight"]), _t(sd[f + "fpn_inner4.bias"]))
    outs = {32: F.conv2d(last, _t(sd[f + "fpn_layer4.weight"]), _t(sd[f + "fpn_layer4.bias"]), padding=1)}
    for k, scale in ((3, 16), (2, 8), (1, 4)):
        top = F.interpolate(last, scale_factor=2, mode="bilinear", align_corners=False)
        lat = F.conv2d(feats[k - 1], _t(sd[f + "fpn_inner%d.weight" % k]), _t(sd[f + "fpn_inner%d.bias" % k]))
        last = lat + top
        outs[scale] = F.conv2d(last, _t(sd[f + "fpn_layer%d.weight" % k]), _t(sd[f + "fpn_layer%d.bias" % k]), padding=1)
    return outs


def backbone_param_shapes(backbone_type="R-101-FPN", prefix="backbone."):
    """(key, shape) list in the reference's state-dict naming."""
    out = []

    def bn(p, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            out.append((p + "." + s, (c,)))
    b = prefix + "body."
    out.append((b + "stem.conv1.weight", (64, 3, 7, 7)))
    bn(b + "stem.bn1", 64)
    cin = 64
    for li, nblocks in enumerate(STAGE_BLOCKS[backbone_type], 1):
        mid, cout = 64 * 2 ** (li - 1), 256 * 2 ** (li - 1)
        for bi in range(nblocks):
            p = b + "layer%d.%d" % (li, bi)
            if bi == 0:
                out.append((p + ".downsample.0.weight", (cout, cin, 1, 1)))
                bn(p + ".downsample.1", cout)
            out.append((p + ".conv1.weight", (mid, cin, 1, 1)))
            bn(p + ".bn1", mid)
            out.append((p + ".conv2.weight", (mid, mid, 3, 3)))
            bn(p + ".bn2", mid)
            out.append((p + ".conv3.weight", (cout, mid, 1, 1)))
            bn(p + ".bn3", cout)
            cin = cout
    f = prefix + "fpn."
    for k in (1, 2, 3, 4):
        out.append((f + "fpn_inner%d.weight" % k, (256, 256 * 2 ** (k - 1), 1, 1)))
        out.append((f + "fpn_inner%d.bias" % k, (256,)))
        out.append((f + "fpn_layer%d.weight" % k, (256, 256, 3, 3)))
        out.append((f + "fpn_layer%d.bias" % k, (256,)))
    return out
