"""ctypes binding of libstemseg_hip.so (include/stemseg_hip.h).

PyTorch appears here only as plumbing: it owns device memory (``tensor.data_ptr()``) and the HIP
stream (``torch.cuda.current_stream().cuda_stream``).  Every numeric operation of the hot path runs in
the hand-written gfx950 kernels behind the C-ABI.  There is NO fallback: if the shared library is
missing, or no GPU is visible, the ops raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEMSEG_HIP_LIB") or os.path.join(_HERE, "lib", "libstemseg_hip.so")      # (override: A/B builds of the library, tools/)

MAX_INSTANCES = 64
MAX_EMB_DIMS = 8
ABI_VERSION = 10


class Volume(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("c_stride", C.c_int64), ("t_stride", C.c_int64), ("y_stride", C.c_int64),
                ("C", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("limit", C.c_int64)]


class DecoderDesc(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("in_channels", C.c_int32), ("inter", C.c_int32 * 4),
                ("T", C.c_int32), ("H4", C.c_int32), ("W4", C.c_int32), ("gn_groups", C.c_int32), ("gn_eps", C.c_float),
                ("pool", C.c_int32 * 3), ("t_scale", C.c_int32 * 3), ("n_out", C.c_int32),
                ("act", C.c_int32 * (2 * MAX_EMB_DIMS)), ("grid_axis", C.c_int32 * (2 * MAX_EMB_DIMS)),
                ("input_layout", C.c_int32), ("concurrency", C.c_int32), ("detached", C.c_int32), ("precision", C.c_int32),
                ("n_clips", C.c_int32), ("feat_clip_stride", C.c_int64 * 4), ("out_clip_stride", C.c_int64)]


class DecoderWeights(C.Structure):
    _fields_ = [("conv_w", C.c_void_p * 7), ("conv_b", C.c_void_p * 7), ("gn_w", C.c_void_p * 7), ("gn_b", C.c_void_p * 7),
                ("fuse_w", C.c_void_p * 3), ("head_w", C.c_void_p), ("head_b", C.c_void_p),
                ("grid_t", C.c_void_p), ("grid_y", C.c_void_p), ("grid_x", C.c_void_p)]


MAX_ENCODER_BLOCKS = 40


class ConvEpilogue(C.Structure):
    _fields_ = [("relu", C.c_int32), ("residual", C.c_void_p), ("res_c_stride", C.c_int64), ("res_t_stride", C.c_int64),
                ("res_y_stride", C.c_int64), ("decode_H", C.c_int32), ("decode_W", C.c_int32), ("precision", C.c_int32),
                ("frames", C.c_int32), ("plan_frames", C.c_int32), ("plan_scratch_floats", C.c_int64)]


class EncoderDesc(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("blocks", C.c_int32 * 4), ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("out_channels", C.c_int32), ("precision", C.c_int32), ("n_clips", C.c_int32), ("clip_frames", C.c_int32),
                ("clip_stride", C.c_int32), ("plan_frames", C.c_int32), ("fuse_tail", C.c_int32)]


_BLK = C.c_void_p * MAX_ENCODER_BLOCKS


class EncoderWeights(C.Structure):
    _fields_ = [("stem_w", C.c_void_p), ("stem_b", C.c_void_p), ("stem_w_s2d", C.c_void_p),
                ("conv1_w", _BLK), ("conv1_b", _BLK), ("conv2_w", _BLK), ("conv2_b", _BLK), ("conv3_w", _BLK), ("conv3_b", _BLK),
                ("down_w", _BLK), ("down_b", _BLK),
                ("fpn_inner_w", C.c_void_p * 4), ("fpn_inner_b", C.c_void_p * 4), ("fpn_layer_w", C.c_void_p * 4), ("fpn_layer_b", C.c_void_p * 4)]


class ClusterParams(C.Structure):
    _fields_ = [("primary_prob_thresh", C.c_float), ("secondary_prob_thresh", C.c_float), ("min_seediness_prob", C.c_float),
                ("max_instances", C.c_int32), ("n_free_dims", C.c_int32), ("free_dim_bandwidths", C.c_float * MAX_EMB_DIMS)]


class ClusterMeta(C.Structure):
    _fields_ = [("K", C.c_int32), ("exhausted", C.c_int32), ("n_points", C.c_int64), ("n_unassigned_last", C.c_int64),
                ("centers", (C.c_float * MAX_EMB_DIMS) * MAX_INSTANCES), ("bandwidths", (C.c_float * MAX_EMB_DIMS) * MAX_INSTANCES),
                ("seed_prob", C.c_float * MAX_INSTANCES)]


class ClusterItem(C.Structure):
    _fields_ = [("emb", C.c_void_p), ("bw", C.c_void_p), ("seed", C.c_void_p), ("n_max", C.c_int64), ("n_points_dev", C.c_void_p),
                ("label_start", C.c_int64), ("labels", C.c_void_p), ("meta_dev", C.c_void_p), ("opt_masks", C.c_void_p), ("opt_probs", C.c_void_p),
                ("workspace", C.c_void_p), ("ws_bytes", C.c_size_t)]


# name -> (restype, argtypes); mirrors include/stemseg_hip.h one to one (tests check the export list)
_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "stemseg_hip_version": (C.c_int, []),
    "stemseg_hip_last_error": (C.c_char_p, []),
    "stemseg_hip_device_count": (C.c_int, []),
    "stemseg_hip_profile_enable": (C.c_int, [_I32]),
    "stemseg_hip_profile_read": (C.c_int, [C.POINTER(C.c_double), _I32]),
    "stemseg_hip_padded_geometry": (C.c_int, [_I32, _I32, _I32, _I32, C.POINTER(_I64)]),
    "stemseg_hip_pack_conv_weight": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "stemseg_hip_encoder_plan_offsets": (C.c_int, [_P, _P]),
    "stemseg_hip_packed_weight_bytes_prec": (C.c_int64, [_I32, _I32, _I32, _I32]),
    "stemseg_hip_pack_conv_weight_prec": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "stemseg_hip_conv3d": (C.c_int, [C.POINTER(Volume), _P, _P, C.POINTER(Volume), _I32, _I32, _I32, _I32, _P, _I64, C.POINTER(ConvEpilogue), _P]),
    "stemseg_hip_conv3d_gn_scratch_doubles": (C.c_int64, [_I32, _I32]),
    "stemseg_hip_conv3d_gn": (C.c_int, [C.POINTER(Volume), _P, _P, C.POINTER(Volume), _I32, _I32, _I32, _I32, _P, _I64, _I32, _I32, _F, _P, _P, _P]),
    "stemseg_hip_stem_conv": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "stemseg_hip_encoder_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderDesc)]),
    "stemseg_hip_encoder_init_workspace": (C.c_int, [C.POINTER(EncoderDesc), _P, C.c_size_t, _P]),
    "stemseg_hip_encoder_check_workspace": (C.c_int, [C.POINTER(EncoderDesc), _P, C.c_size_t, C.POINTER(_I32), C.POINTER(_I64), _P]),
    "stemseg_hip_encoder_forward": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), _P, C.POINTER(Volume), _P, C.c_size_t, _P]),
    "stemseg_hip_groupnorm_stats": (C.c_int, [_P, _I32, _I64, _I32, _F, _P, _P, _P]),
    "stemseg_hip_gn_relu_pool": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _I32, C.POINTER(Volume), _P]),
    "stemseg_hip_upsample_trilinear": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, C.POINTER(Volume), _P]),
    "stemseg_hip_copy_to_volume": (C.c_int, [_P, _I32, C.POINTER(Volume), _P]),
    "stemseg_hip_heads": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _I32, C.POINTER(_I32), C.POINTER(_I32), _P, _P, _P, _P, _P]),
    "stemseg_hip_nonfinite_flags": (C.c_int, [_P, _I64, _P, _I32, _P]),
    "stemseg_hip_decoder_workspace_bytes": (C.c_size_t, [C.POINTER(DecoderDesc)]),
    "stemseg_hip_decoder_init_workspace": (C.c_int, [C.POINTER(DecoderDesc), _P, C.c_size_t, _P]),
    "stemseg_hip_decoder_check_workspace": (C.c_int, [C.POINTER(DecoderDesc), _P, C.c_size_t, C.POINTER(_I32), C.POINTER(_I64), _P]),
    "stemseg_hip_decoder_forward": (C.c_int, [C.POINTER(DecoderDesc), C.POINTER(DecoderWeights), C.POINTER(_P), _P, _P, C.c_size_t, _P]),
    "stemseg_hip_decoder_join": (C.c_int, [_I32, _P]),
    "stemseg_hip_seediness_accumulate": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "stemseg_hip_fg_mask": (C.c_int, [_P, _F, _F, _P, _I64, _P]),
    "stemseg_hip_fg_mask_frames": (C.c_int, [_P, _P, _F, _P, _I32, _I64, _P]),
    "stemseg_hip_fg_gather": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "stemseg_hip_cluster_workspace_bytes": (C.c_size_t, [_I64]),
    "stemseg_hip_cluster": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _I32, C.POINTER(ClusterParams), _I64, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "stemseg_hip_cluster_batch": (C.c_int, [C.POINTER(ClusterItem), _I32, _I32, _I32, C.POINTER(ClusterParams), _P]),
    "stemseg_hip_overlap_counts": (C.c_int, [_P, _P, _I64, _P, _I32, _P, _I32, _I32, _I32, _P, _P, _P, _P]),
    "stemseg_hip_label_presence": (C.c_int, [_P, _I64, _P, _I32, _P, _I32, _P]),
    "stemseg_hip_relabel": (C.c_int, [_P, _I64, _P, _I32, _P]),
    "stemseg_hip_fg_compact": (C.c_int, [_P, _I32, _I64, _P, _P, _P, _P]),
    "stemseg_hip_labels_to_codes": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _I64, _P]),
    "stemseg_hip_pair_tables": (C.c_int, [_P, _P, _P, _I32, _I64, _I32, _P, _P]),
    "stemseg_hip_codes_to_labels": (C.c_int, [_P, _P, _P, _I32, _I64, _P, _I64, _I32, _P, _P]),
    "stemseg_hip_semseg_accumulate": (C.c_int, [_P, _P, _I32, _I32, _I64, C.POINTER(C.c_int32), _I32, _P]),
    "stemseg_hip_semseg_masks": (C.c_int, [_P, _P, _I32, _I32, _I64, _I32, _P, _P, _P]),
    "stemseg_hip_semseg_fg_clip": (C.c_int, [_P, _I32, _I32, _I64, _F, _P, _P, _P]),
    "stemseg_hip_preprocess_frames": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, C.POINTER(C.c_float), C.POINTER(C.c_float), _I32, _I32, _P, _P]),
    "stemseg_hip_scatter_instance_index": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _P, _I32, _I32, _P]),
    "stemseg_hip_resample_instance_masks": (C.c_int, [_P, _I32, _I32, C.c_float, _I32, _I32, _I32, _I32, _P, _P]),
}

SEMSEG_OUTPUT_TYPES = {None: 0, "none": 0, "logits": 1, "probs": 2, "argmax": 3}

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libstemseg_hip.so not found at %s -- build it with `python stem-seg_amd/build.py` "
                               "(hipcc --offload-arch=gfx950); the hot path has no CPU fallback" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        v = l.stemseg_hip_version()
        if v != ABI_VERSION:
            raise RuntimeError("libstemseg_hip.so ABI version %d != binding %d" % (v, ABI_VERSION))
        _lib = l
    return _lib


def require_gpu():
    l = lib()
    n = l.stemseg_hip_device_count()
    if n <= 0 or not torch.cuda.is_available():
        raise RuntimeError("stemseg_amd: no MI355X visible (hip device count %d, torch.cuda.is_available()=%s): %s"
                           % (n, torch.cuda.is_available(), l.stemseg_hip_last_error().decode()))
    return n


def check(rc):
    if rc != 0:
        raise RuntimeError("libstemseg_hip error %d: %s" % (rc, lib().stemseg_hip_last_error().decode()))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None):
    if t is None:
        return None
    assert t.is_cuda, "device tensor required"
    assert t.is_contiguous(), "contiguous tensor required"
    # kernels are enqueued on the CURRENT device's current stream (``stream()``): a tensor of another device would be touched from
    # the wrong device / an unordered stream -- wrap the call in ``torch.cuda.device(t.device)``
    assert t.device.index == torch.cuda.current_device(), "tensor on %s but the current device is cuda:%d" % (t.device, torch.cuda.current_device())
    if dtype is not None:
        assert t.dtype == dtype, "expected %s, got %s" % (dtype, t.dtype)
    return C.c_void_p(t.data_ptr())


def profile_enable(on):
    check(lib().stemseg_hip_profile_enable(int(on)))


# profiler tags: convolutions by tile class (work = FLOP) and the streaming kernels (work = algorithmic bytes)
PROFILE_CONV_TAGS = {"conv3x3x3": (9, 8, 4, 2), "conv1x3x3": (36, 28, 27, 24, 22), "conv1x1x1": (18, 17, 14, 16, 12, 19)}      # (19: the fused bottleneck tail, both of its 1x1 GEMMs)
PROFILE_HBM_TAGS = {40: "upsample_trilinear", 41: "gn_stats (partial + finalize)", 42: "gn_relu (apply)", 43: "gn_relu_pool (apply + AvgPool3d)",
                    44: "heads", 45: "fg_gather (count + scan + scatter)", 46: "cluster (all rounds + final)", 47: "stem_conv7x7",
                    48: "maxpool3x3s2", 49: "subsample2", 50: "upsample2x_add (FPN top-down)"}


def profile_read(n_tags=64):
    """{tag: (ms, work, launches)} for the tagged launches since the last read (synchronises)."""
    buf = (C.c_double * (3 * n_tags))()
    check(lib().stemseg_hip_profile_read(buf, n_tags))
    return {t: (buf[3 * t], buf[3 * t + 1], int(buf[3 * t + 2])) for t in range(n_tags) if buf[3 * t + 2] > 0}


# ------------------------------------------------------------------------------------------------ volumes
def padded_geometry(Cn, T, H, W):
    out = (C.c_int64 * 5)()
    check(lib().stemseg_hip_padded_geometry(Cn, T, H, W, out))
    return dict(pitch=out[0], ts=out[1], cs=out[2], total=out[3], interior=out[4])


def dense_volume(t):
    """t: contiguous [C,T,H,W] float32 cuda tensor."""
    Cn, T, H, W = t.shape
    return Volume(t.data_ptr(), T * H * W, H * W, W, Cn, T, H, W, t.numel())


def flat_volume(t):
    """t: contiguous [C, V] -> one-row volume for the 1x1x1 conv."""
    Cn, V = t.shape[0], t[0].numel()
    return Volume(t.data_ptr(), V, 0, 0, Cn, 1, 1, V, t.numel())


def alloc_padded(Cn, T, H, W, device="cuda"):
    g = padded_geometry(Cn, T, H, W)
    return torch.zeros(g["total"], dtype=torch.float32, device=device), g


def alloc_padded_batch(n, Cn, T, H, W, device="cuda"):
    """``n`` zero-haloed buffers of one geometry in ONE allocation, a fixed stride apart (a multiple of 64 floats): the inputs of a
    clip-batched decoder call (StemsegDecoderDesc.feat_clip_stride).  -> (list of n buffer views, geometry, stride in floats)"""
    g = padded_geometry(Cn, T, H, W)
    stride = (g["total"] + 63) // 64 * 64
    whole = torch.zeros(n * stride, dtype=torch.float32, device=device)
    return [whole[c * stride:c * stride + g["total"]] for c in range(n)], g, stride


def padded_halo_view(buf, g, Cn, T, H, W):
    return Volume(buf.data_ptr(), g["cs"], g["ts"], g["pitch"], Cn, T + 2, H + 2, W + 2, g["total"])


def padded_interior_view(buf, g, Cn, T, H, W):
    return Volume(buf.data_ptr() + 4 * g["interior"], g["cs"], g["ts"], g["pitch"], Cn, T, H, W, g["total"] - g["interior"])


def padded_to_dense(buf, g, Cn, T, H, W):
    """Test helper: extract the interior of a zero-haloed buffer as a dense [C,T,H,W] tensor (torch indexing)."""
    v = buf[:Cn * g["cs"]].view(Cn, T + 2, H + 2, g["pitch"])
    return v[:, 1:T + 1, 1:H + 1, 1:W + 1].contiguous()


# ------------------------------------------------------------------------------------------------ decoder ops
def pack_conv_weight(w):
    """w: [Cout, Cin, kt, kh, kw] -> packed [Cin/4][taps][4][Cout] (flat tensor)."""
    w = w.contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    out = torch.empty(Cout * Cin * taps, dtype=torch.float32, device=w.device)
    check(lib().stemseg_hip_pack_conv_weight(ptr(w, torch.float32), ptr(out), Cout, Cin, taps, stream()))
    return out


PRECISIONS = {"f32": 0, "bf16x6": 2, "f16x3": 3}
# what the matrix products of a mode compute with (bench.py's ``dtype`` names these, never a bare "f32"):
#   operand_significand_bits -- significand bits each fp32 operand keeps (fp32 has 24); products are exact, accumulation is fp32
PRECISION_INFO = {
    "f32": dict(operand_significand_bits=24, products_per_fp32_product=1, mfma="v_mfma_f32_32x32x2_f32", exponent_range="fp32"),
    "bf16x6": dict(operand_significand_bits=24, products_per_fp32_product=6, mfma="v_mfma_f32_32x32x16_bf16", exponent_range="fp32"),
    "f16x3": dict(operand_significand_bits=22, products_per_fp32_product=3, mfma="v_mfma_f32_32x32x16_f16",
                  exponent_range="2.5e-4 <= |activation| < 2.6e5 at full width; beyond: non-finite output, flagged, re-run in bf16x6"),
}
# MFMA mode of every convolution unless a module's ``precision`` is set (InferenceModel.set_precision).  "f16x3" (default) =
# operands scaled by powers of two and split into two fp16 terms, three products, fp32 accumulation; "bf16x6" = every fp32 operand
# split EXACTLY into three bf16 terms, six products (fp32's full exponent range, twice the matrix work).  Both give fp32-level
# results: error vs an fp64 convolution = that of the fp32-input MFMA kernel on every kernel class (tests/test_gpu_bf16x6.py),
# labels identical on every reference flow.  "f32" = v_mfma_f32_32x32x2_f32 on fp32 operands.  STEMSEG_PRECISION overrides the default.
DEFAULT_PRECISION = os.environ.get("STEMSEG_PRECISION", "f16x3")
assert DEFAULT_PRECISION in PRECISIONS, DEFAULT_PRECISION


def pack_conv_weight_any(w, precision="f32"):
    """Pack for the given MFMA mode: 'f32' (exact fp32 MFMA), 'bf16x6' (exact three-term split, 6 products: fp32-level results) or
    'f16x3' (two fp16 terms of the power-of-two-scaled operands, 3 products: fp32-level results for |activation| < 2.6e5); fp32
    accumulate throughout."""
    if precision == "f32":
        return pack_conv_weight(w)
    assert precision in ("bf16x6", "f16x3"), precision
    code = PRECISIONS[precision]
    w = w.contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    nbytes = lib().stemseg_hip_packed_weight_bytes_prec(Cout, Cin, taps, code)
    if nbytes <= 0:
        raise ValueError("pack_conv_weight_any: unsupported shape / precision (%s, Cout %d, Cin %d, taps %d)" % (precision, Cout, Cin, taps))
    out = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)      # opaque 16-B-aligned blob
    check(lib().stemseg_hip_pack_conv_weight_prec(ptr(w, torch.float32), ptr(out), Cout, Cin, taps, code, stream()))
    return out


def stem_conv(frames, w, bias):
    """frames float32 [T,3,H,W], w [64,3,7,7] (FrozenBN folded), bias [64] -> [64,T,H/2,W/2] = relu(conv 7x7 s2 p3 + bias)."""
    require_gpu()
    T, _, H, W = frames.shape
    out = torch.empty(64, T, H // 2, W // 2, dtype=torch.float32, device=frames.device)
    wt = w.reshape(64, 147).t().contiguous()
    check(lib().stemseg_hip_stem_conv(ptr(frames.contiguous(), torch.float32), ptr(wt, torch.float32), ptr(bias.contiguous(), torch.float32), ptr(out), T, H, W, stream()))
    return out


def conv3d(vin, packed_w, bias, vout, k, tile_cfg=0, splitk_scratch=None, epilogue=None):
    """k: int (k x k x k) or a (kt, kh, kw) tuple; epilogue: dict(relu=, residual=tensor, res_strides=(c,t,y), decode=(H,W),
    precision=, plan=(frames, plan_frames, plan_scratch_floats): decide tile / split-K as if the launch held plan_frames frames)."""
    n = 0 if splitk_scratch is None else splitk_scratch.numel()
    kt, kh, kw = (k, k, k) if isinstance(k, int) else k
    e = None
    if epilogue is not None:
        e = ConvEpilogue()
        e.relu = int(epilogue.get("relu", 0))
        r = epilogue.get("residual")
        if r is not None:
            e.residual = r.data_ptr()
            e.res_c_stride, e.res_t_stride, e.res_y_stride = epilogue["res_strides"]
        e.decode_H, e.decode_W = epilogue.get("decode", (0, 0))
        e.precision = PRECISIONS[epilogue.get("precision", "f32")]
        e.frames, e.plan_frames, e.plan_scratch_floats = epilogue.get("plan", (0, 0, 0))
    check(lib().stemseg_hip_conv3d(C.byref(vin), ptr(packed_w), ptr(bias), C.byref(vout), kt, kh, kw, tile_cfg,
                                   ptr(splitk_scratch), n, C.byref(e) if e is not None else None, stream()))


def conv3d_gn(vin, packed_w, bias, vout, k, groups, eps=1e-5, tile_cfg=0, splitk_scratch=None, precision="f32"):
    """Convolution + GroupNorm statistics of its output in one pass -> stats float32 [2 * groups] (mean, rstd per group)."""
    n = 0 if splitk_scratch is None else splitk_scratch.numel()
    kt, kh, kw = (k, k, k) if isinstance(k, int) else k
    dev = packed_w.device
    stats = torch.empty(2 * groups, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib().stemseg_hip_conv3d_gn_scratch_doubles(vout.C, groups), dtype=torch.float64, device=dev)
    check(lib().stemseg_hip_conv3d_gn(C.byref(vin), ptr(packed_w), ptr(bias), C.byref(vout), kt, kh, kw, tile_cfg, ptr(splitk_scratch), n,
                                      PRECISIONS[precision], groups, eps, ptr(stats), ptr(scratch), stream()))
    return stats


def groupnorm_stats(x, groups, eps=1e-5):
    Cn = x.shape[0]
    S = x[0].numel()
    stats = torch.empty(2 * groups, dtype=torch.float32, device=x.device)
    scratch = torch.empty(groups * 128, dtype=torch.float64, device=x.device)
    check(lib().stemseg_hip_groupnorm_stats(ptr(x, torch.float32), Cn, S, groups, eps, ptr(stats), ptr(scratch), stream()))
    return stats


def gn_relu_pool(x, groups, stats, gamma, beta, pool, vout):
    Cn, T, H, W = x.shape
    check(lib().stemseg_hip_gn_relu_pool(ptr(x, torch.float32), Cn, T, H, W, groups, ptr(stats), ptr(gamma), ptr(beta),
                                         int(pool), C.byref(vout), stream()))


def upsample_trilinear(x, st, sy, sx, vout=None):
    Cn, T, H, W = x.shape
    out = None
    if vout is None:
        out = torch.empty(Cn, T * st, H * sy, W * sx, dtype=torch.float32, device=x.device)
        vout = dense_volume(out)
    check(lib().stemseg_hip_upsample_trilinear(ptr(x, torch.float32), Cn, T, H, W, st, sy, sx, C.byref(vout), stream()))
    return out


def copy_to_volume(x, layout, vout):
    check(lib().stemseg_hip_copy_to_volume(ptr(x, torch.float32), layout, C.byref(vout), stream()))


def heads(x, w, bias, act, grid_axis, gt, gy, gx):
    Cin, T, H, W = x.shape
    n_out = w.shape[0]
    out = torch.empty(n_out, T, H, W, dtype=torch.float32, device=x.device)
    a = (C.c_int32 * n_out)(*act)
    g = (C.c_int32 * n_out)(*grid_axis)
    check(lib().stemseg_hip_heads(ptr(x, torch.float32), Cin, T, H, W, ptr(w.contiguous()), ptr(bias), n_out, a, g,
                                  ptr(gt), ptr(gy), ptr(gx), ptr(out), stream()))
    return out


NONFINITE_FLAGS = 64
MAX_HEAD_OUT = 10          # STEMSEG_MAX_HEAD_OUT: widest head the fused heads kernel serves (wider: the 1x1x1 MFMA conv)


class NonFiniteError(FloatingPointError):
    """A head output (embedding / bandwidth / seediness / class logits) of a clip holds inf or NaN: an operand left the range of
    the split convolution mode (f16x3: |activation| < 2.6e5).  Re-run the clip in 'bf16x6' (ClipPipeline.step_checked does)."""


def nonfinite_flags(x, flags=None):
    """x: contiguous float32 device tensor -> int32 [NONFINITE_FLAGS] on the device (1 where a chunk of x holds inf / NaN); no
    synchronisation.  ``flags``: an existing int32 view to (re)write."""
    if flags is None:
        flags = torch.empty(NONFINITE_FLAGS, dtype=torch.int32, device=x.device)
    assert x.is_contiguous() and x.dtype == torch.float32 and flags.numel() == NONFINITE_FLAGS
    check(lib().stemseg_hip_nonfinite_flags(ptr(x), x.numel(), ptr(flags, torch.int32), NONFINITE_FLAGS, stream()))
    return flags


def overflow_status(tensors):
    """Overflow flags of a clip's head outputs: int32 [k, NONFINITE_FLAGS] on the device, one launch per run of tensors that are
    adjacent in memory (embeddings and bandwidths are channel slices of one decoder output).  No synchronisation; hand the result to
    ``read_cluster_meta(meta, status)``."""
    runs = []
    for t in tensors:
        assert t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda
        if runs and runs[-1][0] + 4 * runs[-1][1] == t.data_ptr():
            runs[-1][1] += t.numel()
        else:
            runs.append([t.data_ptr(), t.numel(), t])
    status = torch.empty(len(runs), NONFINITE_FLAGS, dtype=torch.int32, device=tensors[0].device)
    for i, (p_, n, _) in enumerate(runs):
        check(lib().stemseg_hip_nonfinite_flags(C.c_void_p(p_), n, ptr(status[i], torch.int32), NONFINITE_FLAGS, stream()))
    return status


# ------------------------------------------------------------------------------------------------ clustering ops
def seediness_accumulate(acc, plane, first):
    check(lib().stemseg_hip_seediness_accumulate(ptr(acc, torch.float32), ptr(plane, torch.float32), plane.numel(), int(first), stream()))


def fg_mask(acc, count, thr):
    mask = torch.empty(acc.shape, dtype=torch.uint8, device=acc.device)
    check(lib().stemseg_hip_fg_mask(ptr(acc, torch.float32), float(count), float(thr), ptr(mask), acc.numel(), stream()))
    return mask


def fg_mask_frames(acc, counts, thr):
    """acc float32 [F,h,w] (per-frame sums), counts float32 [F] on the device -> uint8 [F,h,w]: acc / counts > thr."""
    Fn = acc.shape[0]
    mask = torch.empty(acc.shape, dtype=torch.uint8, device=acc.device)
    check(lib().stemseg_hip_fg_mask_frames(ptr(acc, torch.float32), ptr(counts, torch.float32), float(thr), ptr(mask), Fn,
                                           acc[0].numel() if Fn else 0, stream()))
    return mask


def fg_gather(emb, bw, seed, fg):
    """emb [E,T,H,W], bw [Ev,T,H,W], seed [1,T,H,W] (or [T,H,W]), fg uint8 [T,H,W].
    Returns max-size outputs (first N rows valid) + frame_offsets [T+1] int64 (device)."""
    E, T, H, W = emb.shape
    Ev = bw.shape[0]
    V = T * H * W
    dev = emb.device
    emb_o = torch.empty(V, E, dtype=torch.float32, device=dev)
    bw_o = torch.empty(V, max(Ev, 1), dtype=torch.float32, device=dev)[:, :Ev].contiguous() if Ev == 0 else \
        torch.empty(V, Ev, dtype=torch.float32, device=dev)
    seed_o = torch.empty(V, dtype=torch.float32, device=dev)
    vox = torch.empty(V, dtype=torch.int32, device=dev)
    offs = torch.empty(T + 1, dtype=torch.int64, device=dev)
    scratch = torch.empty(16 * (V // 1024 + 4), dtype=torch.uint8, device=dev)
    check(lib().stemseg_hip_fg_gather(ptr(emb, torch.float32), ptr(bw, torch.float32), ptr(seed, torch.float32), ptr(fg, torch.uint8),
                                      E, Ev, T, H * W, ptr(emb_o), ptr(bw_o), ptr(seed_o), ptr(vox), ptr(offs), ptr(scratch), stream()))
    return emb_o, bw_o, seed_o, vox, offs


def make_cluster_params(primary, secondary, min_seed, max_instances, free_dim_stds):
    p = ClusterParams()
    p.primary_prob_thresh, p.secondary_prob_thresh, p.min_seediness_prob = primary, secondary, min_seed
    p.max_instances = int(max_instances)
    p.n_free_dims = len(free_dim_stds)
    if len(free_dim_stds):
        # 1 / std^2 exactly as clusterers.py:101-103 (fp32 tensor ops on the host: a handful of scalars)
        fb = (1. / (torch.tensor(list(free_dim_stds), dtype=torch.float32) ** 2)).tolist()
        for i, v in enumerate(fb):
            p.free_dim_bandwidths[i] = v
    return p


def cluster(emb, bw, seed, params, label_start, n_points_dev=None, want_masks=False, want_probs=False):
    """emb [Nmax,E], bw [Nmax,Ev], seed [Nmax].  Returns labels int64 [Nmax], meta (device uint8 blob), masks, probs."""
    n_max, E = emb.shape
    Ev = bw.shape[1]
    dev = emb.device
    labels = torch.empty(n_max, dtype=torch.int64, device=dev)
    meta = torch.empty(C.sizeof(ClusterMeta), dtype=torch.uint8, device=dev)
    masks = torch.empty(params.max_instances, n_max, dtype=torch.uint8, device=dev) if want_masks else None
    probs = torch.empty(params.max_instances, n_max, dtype=torch.float32, device=dev) if want_probs else None
    ws_bytes = lib().stemseg_hip_cluster_workspace_bytes(n_max)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib().stemseg_hip_cluster(ptr(emb, torch.float32), ptr(bw, torch.float32), ptr(seed, torch.float32), n_max,
                                    ptr(n_points_dev), E, Ev, C.byref(params), int(label_start), ptr(labels), ptr(meta),
                                    ptr(masks), ptr(probs), ptr(ws), ws_bytes, stream()))
    return labels, meta, masks, probs


def cluster_batch(point_sets, params, label_start=1):
    """Several independent point sets (the clips of one step) through ONE sequence of launches (grid.y = set): point_sets = list of
    (emb [Nmax,E], bw [Nmax,Ev], seed [Nmax], n_points_dev | None).  Returns a list of (labels int64 [Nmax], meta device blob), bit
    identical to separate ``cluster`` calls; no synchronisation."""
    n = len(point_sets)
    items = (ClusterItem * n)()
    keep, outs = [], []
    E, Ev = point_sets[0][0].shape[1], point_sets[0][1].shape[1]
    for i, (emb, bw, seed, n_dev) in enumerate(point_sets):
        assert emb.shape[1] == E and bw.shape[1] == Ev
        n_max, dev = emb.shape[0], emb.device
        labels = torch.empty(n_max, dtype=torch.int64, device=dev)
        meta = torch.empty(C.sizeof(ClusterMeta), dtype=torch.uint8, device=dev)
        ws_bytes = lib().stemseg_hip_cluster_workspace_bytes(n_max)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        seed = seed.reshape(-1)
        it = items[i]
        it.emb, it.bw, it.seed = ptr(emb, torch.float32), ptr(bw, torch.float32) if Ev else None, ptr(seed, torch.float32)
        it.n_max, it.n_points_dev, it.label_start = n_max, ptr(n_dev), int(label_start)
        it.labels, it.meta_dev, it.opt_masks, it.opt_probs = ptr(labels), ptr(meta), None, None
        it.workspace, it.ws_bytes = ptr(ws), ws_bytes
        keep.append(ws)
        outs.append((labels, meta))
    check(lib().stemseg_hip_cluster_batch(items, n, E, Ev, C.byref(params), stream()))
    return outs


_meta_pinned = {}


def read_cluster_meta(meta_dev, status=None):
    """Device->host copy of the StemsegClusterMeta record (synchronises the current stream).  The copy lands in a
    persistent PINNED host buffer: a direct DMA, no pageable staging (``.cpu()`` on a tensor that lives in a hipGraph's
    private pool was seen to fault the GPU after a few replays on ROCm 7.2).  ``status``: the clip's overflow flags
    (nonfinite_flags, int32 on the device), read with the same synchronisation; raises NonFiniteError when one is set."""
    n = meta_dev.numel()
    key = (meta_dev.device.index, n)
    buf = _meta_pinned.get(key)
    if buf is None:
        buf = _meta_pinned[key] = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    buf.copy_(meta_dev, non_blocking=True)
    sbuf = None
    if status is not None:
        skey = (meta_dev.device.index, "status", status.numel())
        sbuf = _meta_pinned.get(skey)
        if sbuf is None:
            sbuf = _meta_pinned[skey] = torch.empty(status.numel(), dtype=torch.int32, pin_memory=True)
        sbuf.copy_(status.reshape(-1), non_blocking=True)
    torch.cuda.current_stream(meta_dev.device).synchronize()
    if sbuf is not None and bool(sbuf.any()):
        raise NonFiniteError("a head output of this clip holds inf / NaN (an operand left the convolution mode's range): not clustered "
                             "results -- re-run the clip with precision 'bf16x6'")
    return ClusterMeta.from_buffer_copy(buf.numpy().tobytes())


def overlap_counts(labels_a, labels_b, lut_a, lut_b, Ka, Kb):
    dev = labels_a.device
    inter = torch.empty(max(Ka * Kb, 1), dtype=torch.int64, device=dev)
    ca = torch.empty(max(Ka, 1), dtype=torch.int64, device=dev)
    cb = torch.empty(max(Kb, 1), dtype=torch.int64, device=dev)
    check(lib().stemseg_hip_overlap_counts(ptr(labels_a, torch.int64), ptr(labels_b, torch.int64), labels_a.numel(),
                                           ptr(lut_a, torch.int32), lut_a.numel(), ptr(lut_b, torch.int32), lut_b.numel(),
                                           Ka, Kb, ptr(inter), ptr(ca), ptr(cb), stream()))
    return inter[:Ka * Kb].view(Ka, Kb), ca[:Ka], cb[:Kb]


def label_presence(labels_list, cap):
    """-> (present uint8 [cap + 1] on the device: byte l = label l occurs, byte ``cap`` = a negative (outlier) label occurs;
    max_plus_1 int64 [1] on the device) over all tensors of ``labels_list`` (int64, same device); no synchronisation."""
    dev = labels_list[0].device
    present = torch.empty(int(cap) + 1, dtype=torch.uint8, device=dev)
    mx = torch.empty(1, dtype=torch.int64, device=dev)
    for k, l in enumerate(labels_list):
        check(lib().stemseg_hip_label_presence(ptr(l, torch.int64) if l.numel() else None, l.numel(), ptr(present), int(cap),
                                               ptr(mx), int(k > 0), stream()))
    return present, mx


def relabel(labels, mapping):
    check(lib().stemseg_hip_relabel(ptr(labels, torch.int64), labels.numel(), ptr(mapping, torch.int64), mapping.numel(), stream()))


# ---- clip-parallel stitching (pipeline.run_sequence_sharded) ------------------------------------------------
def fg_compact(fg):
    """fg uint8 [T,H,W] -> (voxel_index int32 [V] (first N valid), frame_offsets int64 [T+1]) on the device, no sync."""
    T = fg.shape[0]
    HW = fg[0].numel()
    V = T * HW
    vox = torch.empty(V, dtype=torch.int32, device=fg.device)
    offs = torch.empty(T + 1, dtype=torch.int64, device=fg.device)
    scratch = torch.empty(16 * (V // 1024 + 4), dtype=torch.uint8, device=fg.device)
    check(lib().stemseg_hip_fg_compact(ptr(fg, torch.uint8), T, HW, ptr(vox), ptr(offs), ptr(scratch), stream()))
    return vox, offs


def labels_to_codes(labels, vox, n_points_dev, label_start, codes_out):
    """codes_out uint8 [V] (a contiguous view, e.g. one clip's [T,h,w] block of the exchange buffer) := 0, then
    codes_out[vox[i]] = labels[i] - label_start + 1 (255: outlier) for the first N points."""
    check(lib().stemseg_hip_labels_to_codes(ptr(labels, torch.int64) if labels.numel() else None, ptr(vox, torch.int32) if vox.numel() else None,
                                            ptr(n_points_dev), min(labels.numel(), codes_out.numel()), int(label_start),
                                            ptr(codes_out, torch.uint8), codes_out.numel(), stream()))
    return codes_out


def pair_tables(codes, plane_a, plane_b, B):
    """codes uint8 [P, HW]; plane_a / plane_b int32 [n] on the device -> int32 [n, B, B] on the device (no sync)."""
    n = plane_a.numel()
    HW = codes.shape[1]
    tables = torch.empty(n, B, B, dtype=torch.int32, device=codes.device)
    check(lib().stemseg_hip_pair_tables(ptr(codes, torch.uint8), ptr(plane_a, torch.int32), ptr(plane_b, torch.int32), n, HW, B,
                                        ptr(tables), stream()))
    return tables


def codes_to_labels(codes, vox, items, lut, max_count, n_out):
    """items int64 [n,5] = (src_begin, count, vbase, plane, out_begin), lut int64 [n,B] (both on the device) -> int64 [n_out]."""
    n, B = lut.shape
    out = torch.empty(n_out, dtype=torch.int64, device=codes.device)
    check(lib().stemseg_hip_codes_to_labels(ptr(codes, torch.uint8), ptr(vox, torch.int32), ptr(items, torch.int64), n, int(max_count),
                                            ptr(lut, torch.int64), codes.shape[1], B, ptr(out), stream()))
    return out


def semseg_accumulate(acc, clip_logits, frame_index):
    """acc [F,C,H,W] (zero-initialised) += clip_logits [C,T,H,W] at the clip's frames (host list of T distinct ints)."""
    require_gpu()
    Cn, T, H, W = clip_logits.shape
    assert acc.is_contiguous() and clip_logits.is_contiguous() and tuple(acc.shape[1:]) == (Cn, H, W)
    idx = (C.c_int32 * T)(*[int(t) for t in frame_index])
    check(lib().stemseg_hip_semseg_accumulate(ptr(acc, torch.float32), ptr(clip_logits, torch.float32), Cn, T, H * W, idx, acc.shape[0], stream()))


def semseg_masks(acc, counts, output_type="probs"):
    """acc [F,C,H,W], counts float32 [F] (device) -> (fg [F,H,W] float32, multiclass | None); inference_model.py:197-231."""
    require_gpu()
    Fn, Cn, H, W = acc.shape
    code = SEMSEG_OUTPUT_TYPES[output_type]
    fg = torch.empty(Fn, H, W, dtype=torch.float32, device=acc.device)
    mc = None
    if Cn > 2 and code == 3:
        mc = torch.empty(Fn, H, W, dtype=torch.int64, device=acc.device)
    elif Cn > 2 and code in (1, 2):
        mc = torch.empty(Fn, Cn - 1, H, W, dtype=torch.float32, device=acc.device)
    check(lib().stemseg_hip_semseg_masks(ptr(acc, torch.float32), ptr(counts, torch.float32), Fn, Cn, H * W, code, ptr(fg), ptr(mc) if mc is not None else None, stream()))
    return fg, mc


def semseg_fg_clip(clip_logits, thr=0.5, want_prob=False):
    """One independent clip: logits [C,T,H,W] -> (fg mask uint8 [T,H,W], fg probability float [T,H,W] | None)."""
    require_gpu()
    Cn, T, H, W = clip_logits.shape
    mask = torch.empty(T, H, W, dtype=torch.uint8, device=clip_logits.device)
    prob = torch.empty(T, H, W, dtype=torch.float32, device=clip_logits.device) if want_prob else None
    check(lib().stemseg_hip_semseg_fg_clip(ptr(clip_logits, torch.float32), Cn, T, H * W, float(thr), ptr(prob), ptr(mask), stream()))
    return mask, prob


def scatter_instance_index(ys, xs, labels, lut, H, W):
    """dense uint8 [H,W]: lut[label + 1] at the (ys, xs) of the frame's foreground points, 0 elsewhere (davis.py:76-77)."""
    require_gpu()
    dense = torch.empty(H, W, dtype=torch.uint8, device=lut.device)
    n = labels.numel()
    check(lib().stemseg_hip_scatter_instance_index(ptr(ys, torch.int64) if n else None, ptr(xs, torch.int64) if n else None,
                                                   ptr(labels, torch.int64) if n else None, n, ptr(lut, torch.int32), lut.numel(),
                                                   ptr(dense), H, W, stream()))
    return dense


def resample_instance_masks(dense, mask_scale, crop_hw, out_hw):
    """dense uint8 [h,w] -> condensed uint8 [out_h,out_w] through x mask_scale, crop, resize, > 0.5 (davis.py:79-110)."""
    require_gpu()
    h, w = dense.shape
    out = torch.empty(out_hw[0], out_hw[1], dtype=torch.uint8, device=dense.device)
    check(lib().stemseg_hip_resample_instance_masks(ptr(dense, torch.uint8), h, w, float(mask_scale), int(crop_hw[0]), int(crop_hw[1]),
                                                    int(out_hw[0]), int(out_hw[1]), ptr(out), stream()))
    return out


def preprocess_frames(frames_u8, new_hw, pad_hw, mean, std, unit_scale=False, flip_channels=False):
    """uint8 [T,H0,W0,3] (device) -> float32 [T,3,pad_h,pad_w]: resize, normalise, pad in one launch."""
    require_gpu()
    T, H0, W0, ch = frames_u8.shape
    assert ch == 3
    out = torch.empty(T, 3, pad_hw[0], pad_hw[1], dtype=torch.float32, device=frames_u8.device)
    m, s = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    check(lib().stemseg_hip_preprocess_frames(ptr(frames_u8, torch.uint8), T, H0, W0, int(new_hw[0]), int(new_hw[1]), int(pad_hw[0]), int(pad_hw[1]),
                                              m, s, int(bool(unit_scale)), int(bool(flip_channels)), ptr(out), stream()))
    return out
