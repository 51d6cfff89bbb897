#!/usr/bin/env python3
"""Debug: encoder stage outputs (workspace segments S0, X1, Cst0..3) vs the torch CPU oracle, for the checkpoint-like BN statistics."""
import ctypes as C, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
from stemseg_amd import config, hip
from stemseg_amd.modeling.backbone import ResNetFPN
from oracle import encoder as oenc
from tests import synth
BT = os.environ.get("BT", "R-101-FPN")
config.load_preset("davis")
bb = ResNetFPN(BT).cuda()
rs = np.random.RandomState(5)
sd = {k: np.asarray(synth.synth_param(k, v.shape, 17)).reshape(v.shape).astype(np.float32) for k, v in bb.state_dict().items()}
if os.environ.get("PLAIN") != "1":
    for k in list(sd):
        if not k.endswith("running_var"): continue
        bn = k[:-len(".running_var")]
        conv = bn.replace("bn1", "conv1").replace("bn2", "conv2").replace("bn3", "conv3")
        if bn.endswith("downsample.1"): conv = bn[:-1] + "0"
        var = (10.0 ** rs.uniform(-6, 2, size=sd[k].shape)).astype(np.float32)
        sd[k] = var
        sd[conv + ".weight"] = sd[conv + ".weight"] * np.sqrt(var)[:, None, None, None]
        top = 0.5 if bn.endswith("bn3") else 3.0
        sd[bn + ".weight"] = (10.0 ** rs.uniform(-4, np.log10(top), size=var.shape)).astype(np.float32)
        sd[bn + ".running_mean"] = (sd[bn + ".running_mean"] * np.sqrt(var)).astype(np.float32)
bb.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
frames = torch.from_numpy(synth.synth_frames(2, 64, 96, seed=3).astype(np.float32)).permute(0, 3, 1, 2) - 110.0
# torch stages
x = F.conv2d(frames, torch.from_numpy(sd["body.stem.conv1.weight"]), stride=2, padding=3)
s0 = F.relu(oenc._frozen_bn(x, sd, "body.stem.bn1"))
x1 = F.max_pool2d(s0, 3, 2, 1)
stages, x = [], x1
for li, nb in enumerate(oenc.STAGE_BLOCKS[BT], 1):
    for bi in range(nb):
        x = oenc._bottleneck(x, sd, "body.layer%d.%d" % (li, bi), 2 if (bi == 0 and li > 1) else 1, bi == 0)
    stages.append(x)
for prec in os.environ.get("PRECS", "f32,f16x3").split(","):
    bb.precision = prec
    outs = bb.forward(frames.cuda())
    torch.cuda.synchronize()
    (key, ws), = bb._ws.items()
    offs = (C.c_int64 * 25)()
    hip.check(hip.lib().stemseg_hip_encoder_plan_offsets(C.byref(bb._desc(key[0], key[1], key[2], 1)), offs))
    names = ["S0", "X1", "A", "B"] + ["Cst%d" % i for i in range(4)] + ["M1_%d" % i for i in range(4)] + ["M2", "DS", "XS"] + ["L%d" % i for i in range(4)] + ["FO%d" % i for i in range(4)] + ["SK", "total"]
    o = dict(zip(names, list(offs)))
    w32 = ws.view(torch.float32)
    def seg(name, Cn, T, h, w):
        return w32[o[name]:o[name] + Cn * T * h * w].view(Cn, T, h, w).permute(1, 0, 2, 3).cpu()
    T = 2
    for name, ref in (("S0", s0), ("X1", x1)):
        got = seg(name, ref.shape[1], T, ref.shape[2], ref.shape[3])
        print(prec, name, "max|ref| %.4g  rel err %.3e" % (float(ref.abs().max()), float((got - ref).abs().max() / ref.abs().max())))
    for i, ref in enumerate(stages):
        print(prec, "Cst%d offset" % i, o["Cst%d" % i], "shape", tuple(ref.shape), "max|ref| %.4g" % float(ref.abs().max()))
    refd = oenc.resnet_fpn(frames, sd, BT, prefix="")
    for lvl, s_ in enumerate((4, 8, 16, 32)):
        r = refd[s_]
        print(prec, "FPN %d rel err %.3e  max|got| %.4g max|ref| %.4g nonfinite %d" % (s_, float((outs[lvl].cpu() - r).abs().max() / r.abs().max()), float(outs[lvl].abs().max()), float(r.abs().max()), int((~torch.isfinite(outs[lvl])).sum())))
