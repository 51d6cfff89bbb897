#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container: it imports the reference from
/root/reference through tools/ref_shim.py, feeds it the deterministic synthetic inputs/weights of
tests/synth.py and stores the *outputs* (plus tiny hand-made inputs) as .npz fixtures under
tests/golden/.  The reference publishes no tests or golden vectors of its own,
so these files are what pins the oracle and the host logic.

    python tools/make_goldens.py            # regenerate everything (spawns one process per cfg preset,
                                            # because the reference's `cfg` is a process-global singleton)
    python tools/make_goldens.py --group dec_T8

Fixtures are data only: inputs and expected outputs.  No reference source text is stored.
"""
import argparse
import os
import subprocess
import sys
from functools import partial

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden")

from tests import synth  # noqa: E402

GROUPS = ["dec_T8", "dec_T16", "dec_T4", "dec_T2", "dec_T24", "semseg", "masks", "config0", "model_ytvis", "model_kitti", "encoder", "model_davis", "cluster", "chainer", "chainer_long", "chainer_ties", "misc"]


def _save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-40s %8.1f kB" % (os.path.relpath(path, ROOT), os.path.getsize(path) / 1e3))


def _load_synth_weights(module, seed, prefix=""):
    import torch
    sd = module.state_dict()
    new = {k: torch.from_numpy(np.asarray(synth.synth_param(prefix + k, v.shape, seed))).reshape(v.shape)
           for k, v in sd.items()}
    module.load_state_dict(new)


# ------------------------------------------------------------------------------------------------
def gen_decoders(T):
    import ref_shim
    ref_shim.install(num_frames=T)
    import torch
    import torch.nn as nn
    from stemseg.modeling.embedding_decoder import EMBEDDING_HEAD_REGISTRY
    from stemseg.modeling.seediness_decoder import SEEDINESS_HEAD_REGISTRY
    Emb = EMBEDDING_HEAD_REGISTRY["squeeze_expand_decoder"]
    Seed = SEEDINESS_HEAD_REGISTRY["squeeze_expand_decoder"]
    norm = partial(nn.GroupNorm, 32)
    inter = [256, 256, 128, 128]

    cases = [
        # name,            E, mode,   tanh,  seed_out, H32, W32, wseed
        ("emb_xyff_noseed", 4, "xyff", True,  False,    3,   4,   1),
        ("emb_xyff_seed",   4, "xyff", True,  True,     3,   5,   2),
        ("emb_xyt_seed",    3, "xyt",  True,  True,     4,   3,   3),
        ("emb_xytf_notanh", 4, "xytf", False, True,     3,   3,   4),
        ("emb_xy",          2, "xy",   True,  False,    3,   3,   5),
    ]
    if T != 8:
        cases = cases[:2]
    out = {}
    with torch.no_grad():
        for name, E, mode, tanh, so, h32, w32, ws in cases:
            head = Emb(256, inter, E, tanh_activation=tanh, seediness_output=so, experimental_dims=mode,
                       PoolType=nn.AvgPool3d, NormType=norm).eval()
            _load_synth_weights(head, ws, prefix="embedding_head.")
            feats = [torch.from_numpy(f)[None] for f in synth.synth_features(T, h32, w32, seed=ws)]
            y = head(feats)[0].numpy()
            out[name] = y
            out[name + "__meta"] = np.array([E, h32, w32, ws, int(tanh), int(so)], np.int64)
            out[name + "__mode"] = np.array(mode)
        head = Seed(256, inter, PoolType=nn.AvgPool3d, NormType=norm).eval()
        _load_synth_weights(head, 6, prefix="seediness_head.")
        feats = [torch.from_numpy(f)[None] for f in synth.synth_features(T, 3, 4, seed=6)]
        out["seediness"] = head(feats)[0].numpy()
        out["seediness__meta"] = np.array([0, 3, 4, 6, 0, 0], np.int64)
    _save("decoder_T%d" % T, **out)


# ------------------------------------------------------------------------------------------------
def gen_semseg():
    """Semseg head (semseg_decoder.py:12-116) + InferenceModel.get_semseg_masks (inference_model.py:197-231)."""
    import ref_shim
    ref_shim.install(num_frames=8)
    import torch
    import torch.nn as nn
    from stemseg.modeling.semseg_decoder import SEMSEG_HEAD_REGISTRY
    from stemseg.modeling.inference_model import InferenceModel
    Sem = SEMSEG_HEAD_REGISTRY["squeeze_expand_decoder"]
    norm = partial(nn.GroupNorm, 32)
    out = {}
    with torch.no_grad():
        # name, num_classes, fg channel, H32, W32, seed     (2: davis-style binary; 3+1: kitti; 40+1: ytvis -> wide head)
        for name, ncls, fg, h32, w32, ws in (("sem_bin", 2, False, 3, 4, 31), ("sem_kitti", 3, True, 3, 3, 32),
                                             ("sem_ytvis", 40, True, 3, 4, 33)):
            head = Sem(256, ncls, [128, 128, 64, 64], (4, 8, 16, 32), foreground_channel=fg,
                       PoolType=nn.AvgPool3d, NormType=norm).eval()
            _load_synth_weights(head, ws, prefix="semseg_head.")
            feats = [torch.from_numpy(f)[None] for f in synth.synth_features(8, h32, w32, seed=ws)]
            y = head(feats[::-1])[0]                       # the head wants 4x..32x and reverses internally
            out[name] = y.numpy() if ncls < 8 else y.numpy().reshape(-1)[::3].copy()       # wide head: every 3rd value
            out[name + "__shape"] = np.array(y.shape, np.int64)
            out[name + "__meta"] = np.array([ncls, int(fg), h32, w32, ws], np.int64)
            # get_semseg_masks on "averaged" logits: frame t seen (1 + t % 3) times
            T = y.shape[1]
            acc = [[y[:, t][None] * float(1 + t % 3), 1 + t % 3] for t in range(T)]
            for kind in (("logits", "probs", "argmax") if ncls < 8 else ("argmax",)):
                stub = type("S", (), {})()
                stub._model = type("M", (), {"semseg_head": head})()
                stub.semseg_generation_on_gpu = False
                stub.semseg_output_type = kind
                try:
                    fgm, mc = InferenceModel.get_semseg_masks(stub, acc)
                except AttributeError:
                    # reference quirk: with a 2-channel head multiclass_masks stays a list and `.cpu()` raises
                    # (inference_model.py:231) -- recorded, not papered over
                    out["%s_masks_raise" % name] = np.array(1, np.int64)
                    continue
                out["%s_fg_%s" % (name, kind)] = fgm.numpy()
                if torch.is_tensor(mc):
                    out["%s_mc_%s" % (name, kind)] = mc.numpy()
    _save("semseg", **out)


# ------------------------------------------------------------------------------------------------
def _label_frames(n_frames, h, w, seed):
    """Per-frame label maps at mask resolution: moving rectangles / a disc with ids 1..5, an outlier id -1, 0 = background."""
    rng = np.random.RandomState(seed)
    maps = np.zeros((n_frames, h, w), np.int64)
    yy, xx = np.mgrid[0:h, 0:w]
    for t in range(n_frames):
        maps[t, 2:9, 1 + t:8 + t] = 1
        maps[t, 10 + (t % 3):18, 3:12] = 2
        if t >= 2:
            maps[t][(yy - 12) ** 2 + (xx - (w - 8 - t)) ** 2 <= 20] = 3
        if t < 4:
            maps[t, h - 5:h - 1, w - 7:w - 1] = 4            # reaches into the zero-padded border of the network input
        maps[t, 0:2, w - 3:w] = 5
        maps[t][(rng.uniform(size=(h, w)) < 0.02) & (maps[t] > 0)] = -1       # scattered outliers inside instances
    return maps


def gen_masks():
    """DavisOutputGenerator.process_sequence (output_utils/davis.py:38-116): label scatter -> one-hot -> bilinear x4 -> crop the
    zero padding -> bilinear resize to the image size -> > 0.5 -> condensed uint8 map; read back from the PNGs it writes."""
    import shutil
    import ref_shim
    cfg = ref_shim.install()
    import torch
    from PIL import Image
    from stemseg.inference.output_utils.davis import DavisOutputGenerator
    tmp = os.path.join(ROOT, ".tmp_goldens")
    out = {}
    names = []
    #        name       mask h, w   image h, w   MIN  MAX   frames max_tracks
    cases = [("up",      24, 32,     70, 100,     90,  128,  6,     10),     # resized 90x128 inside the padded 96x128
             ("ident",   24, 32,     96, 128,     96,  128,  4,     10),     # image already at network size
             ("down",    16, 24,     200, 311,    60,  96,   5,     3),      # shrink + max_tracks truncation
             ]
    try:
        for name, h, w, ih, iw, mn, mx, nf, max_tracks in cases:
            cfg.INPUT.update_param("MIN_DIM", mn)
            cfg.INPUT.update_param("MAX_DIM", mx)
            maps = _label_frames(nf, h, w, seed=h + nf)
            idxes, labels = [], []
            counts, life = {}, {}
            for t in range(nf):
                ys, xs = np.nonzero(maps[t])
                idxes.append((torch.from_numpy(ys), torch.from_numpy(xs)))
                labels.append(torch.from_numpy(maps[t][ys, xs]))
                for i in np.unique(maps[t][ys, xs]).tolist():
                    counts[i] = counts.get(i, 0) + int((maps[t] == i).sum())
                    lo, hi = life.get(i, (10000, -1))
                    life[i] = (min(lo, t), max(hi, t))
            lifetimes = {k: v[1] - v[0] for k, v in life.items()}
            seq = type("Seq", (), {"image_dims": (ih, iw), "id": name})()
            gen = DavisOutputGenerator(tmp, -1, False)
            keep, _ = gen.process_sequence(seq, idxes, labels, counts, lifetimes, None, (h, w), 4.0, max_tracks, device="cpu")
            pngs = [np.array(Image.open(os.path.join(tmp, "results", name, "%05d.png" % t))) for t in range(nf)]
            out[name + "__maps"] = maps
            out[name + "__dims"] = np.array([h, w, ih, iw, mn, mx, nf, max_tracks], np.int64)
            out[name + "__lifetime_keys"] = np.array(list(lifetimes.keys()), np.int64)
            out[name + "__lifetime_vals"] = np.array(list(lifetimes.values()), np.int64)
            out[name + "__keep"] = np.array(keep, np.int64)
            out[name + "__condensed"] = np.stack(pngs, 0).astype(np.uint8)
            names.append(name)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["__names"] = np.array(names)
    _save("masks", **out)


# ------------------------------------------------------------------------------------------------
def gen_encoder():
    import ref_shim
    cfg = ref_shim.install()
    import torch
    from stemseg.modeling.backbone import BACKBONE_REGISTRY
    out = {}
    with torch.no_grad():
        for btype, hw, seed, stride in (("R-50-FPN", (64, 96), 11, 3), ("R-101-FPN", (64, 64), 12, 5)):
            cfg.MODEL.BACKBONE.update_param("TYPE", btype)
            bb = BACKBONE_REGISTRY[btype](cfg).eval()
            _load_synth_weights(bb, seed, prefix="backbone.")
            x = synth.synth_frames(2, hw[0], hw[1], seed=seed).astype(np.float32)
            x = torch.from_numpy(x).permute(0, 3, 1, 2) - torch.tensor([102.9801, 115.9465, 122.7717])[None, :, None, None]
            feats = bb(x)
            tag = btype.replace("-", "")
            for s, f in zip((4, 8, 16, 32), feats):
                out["%s_s%d" % (tag, s)] = f.numpy().reshape(-1)[::stride].copy()
                out["%s_s%d__shape" % (tag, s)] = np.array(f.shape, np.int64)
            out[tag + "__meta"] = np.array([hw[0], hw[1], seed, stride], np.int64)
    _save("encoder", **out)


# ------------------------------------------------------------------------------------------------
def gen_model_davis():
    """InferenceModel.forward end-to-end on the DAVIS preset with an R-50 backbone (BASELINE config 0
    scaled down), windowing via get_subsequence_frames, fg mask via get_fg_masks_from_seediness."""
    import ref_shim
    cfg = ref_shim.install()
    import torch
    cfg.merge_from_file(os.path.join(ref_shim.REFERENCE_ROOT, "stemseg", "config", "davis_1.yaml"))
    cfg.INPUT.update_param("MIN_DIM", 96)
    cfg.INPUT.update_param("MAX_DIM", 128)
    cfg.MODEL.BACKBONE.update_param("TYPE", "R-50-FPN")
    from stemseg.modeling.inference_model import InferenceModel
    from stemseg.inference.main import get_subsequence_frames, TrackGenerator

    model = InferenceModel(None, cpu_workers=0, preload_images=False, semseg_output_type=None, resize_scale=1.0)
    _load_synth_weights(model._model, 21)
    out = {}
    for tag, nframes, overlap in (("seq12", 12, 4), ("seq5", 5, 4)):
        frames = synth.synth_frames(nframes, 96, 128, seed=21)
        subseqs, _ = get_subsequence_frames(nframes, 8, "davis", overlap)
        res = model([f for f in frames], subseqs)
        out[tag + "__subseqs"] = np.array(subseqs, np.int64)
        for i, e in enumerate(res["embeddings"]):
            out["%s_c%d_frames" % (tag, i)] = np.array(e.subseq_frames, np.int64)
            out["%s_c%d_emb" % (tag, i)] = e.embeddings.numpy()
            out["%s_c%d_bw" % (tag, i)] = e.bandwidths.numpy()
            out["%s_c%d_seed" % (tag, i)] = e.seediness.numpy()
        allseed = np.concatenate([e.seediness.numpy().ravel() for e in res["embeddings"]])
        thr = float(np.median(allseed))

        class _Fake:
            seediness_fg_threshold = thr
        out[tag + "__fg_thr"] = np.float64(thr)
        out[tag + "__fg"] = TrackGenerator.get_fg_masks_from_seediness(_Fake, res).numpy()
    _save("model_davis", **out)


# ------------------------------------------------------------------------------------------------
def gen_config0():
    """BASELINE configs[0]: one synthetic 8 x 256 x 448 clip, random-init ResNet-50, through the REFERENCE's CPU path --
    InferenceModel.forward (encoder, both decoders, bandwidth activation), fg mask = seediness > 0.25, OnlineChainer /
    SequentialClustering.  Float maps are stored as every 5th value, labels and the fg mask whole."""
    import ref_shim
    cfg = ref_shim.install()
    import torch
    cfg.merge_from_file(os.path.join(ref_shim.REFERENCE_ROOT, "stemseg", "config", "davis_1.yaml"))
    cfg.INPUT.update_param("MIN_DIM", 256)
    cfg.INPUT.update_param("MAX_DIM", 448)
    cfg.MODEL.BACKBONE.update_param("TYPE", "R-50-FPN")
    from stemseg.modeling.inference_model import InferenceModel
    from stemseg.inference.clusterers import SequentialClustering
    from stemseg.inference.online_chainer import OnlineChainer
    model = InferenceModel(None, cpu_workers=0, preload_images=False, semseg_output_type=None, resize_scale=1.0)
    _load_synth_weights(model._model, 71)
    with torch.no_grad():
        model._model.seediness_head.conv_out.weight.mul_(30.0)       # spread the random-init seediness over (0, 1)
    frames = synth.synth_frames(8, 256, 448, seed=71)
    res = model([f for f in frames], [list(range(8))])
    e = res["embeddings"][0]
    thr = float(np.float32(e.seediness.median()))                    # half of the pixels foreground (the threshold is a CLI knob)
    min_seed = float(np.float32(e.seediness.flatten().quantile(0.9)))
    fg = (e.seediness[0] > thr).byte()
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, min_seed, 2, [0.3, 0.3], "cpu"), 1.0)
    d = dict(frames=list(e.subseq_frames), embeddings=e.embeddings.clone(), bandwidths=e.bandwidths.clone(), seediness=e.seediness.clone())
    (track, counts, life), _, _, _, meta = ch.process(fg, [d])
    out = {"emb": e.embeddings.numpy().reshape(-1)[::5].copy(), "bw": e.bandwidths.numpy().reshape(-1)[::5].copy(),
           "seed": e.seediness.numpy().reshape(-1)[::5].copy(), "shape": np.array(e.embeddings.shape, np.int64),
           "fg_bits": np.packbits(fg.numpy().astype(bool).reshape(-1)), "fg_shape": np.array(fg.shape, np.int64),
           "labels": np.concatenate([l.numpy() for l in track]).astype(np.int16),
           "instance_labels": np.array(meta[0]["instance_labels"], np.int64),
           "pt_counts": np.array(sorted(counts.items()), np.int64).reshape(-1, 2),
           "thresholds": np.array([thr, min_seed], np.float64)}
    print("config0: %d fg points, %d instances" % (int(fg.sum()), len(meta[0]["instance_labels"])))
    _save("config0", **out)


# ------------------------------------------------------------------------------------------------
def gen_model_ytvis():
    """BASELINE configs[2] flow at a reduced size through the REFERENCE: youtube_vis.yaml (7-channel embedding head with its own
    seediness, 40+1-channel semseg head with inter [256]*4), --resize_embeddings (resize_scale 4): semseg logits resized x4 and
    averaged over the clips, fg mask = sigmoid(fg logit) > 0.5, class argmax; OnlineChainer(embedding_resize_factor=4) resizes
    embeddings / bandwidths / seediness x4 and clusters at full resolution, two overlapping clips stitched."""
    import ref_shim
    cfg = ref_shim.install()
    import torch
    cfg.merge_from_file(os.path.join(ref_shim.REFERENCE_ROOT, "stemseg", "config", "youtube_vis.yaml"))
    cfg.INPUT.update_param("MIN_DIM", 96)
    cfg.INPUT.update_param("MAX_DIM", 128)
    cfg.MODEL.BACKBONE.update_param("TYPE", "R-50-FPN")
    from stemseg.modeling.inference_model import InferenceModel
    from stemseg.inference.clusterers import SequentialClustering
    from stemseg.inference.online_chainer import OnlineChainer
    from stemseg.inference.main import get_subsequence_frames
    model = InferenceModel(None, cpu_workers=0, preload_images=False, semseg_output_type="argmax", resize_scale=4.0,
                           semseg_generation_on_gpu=False)
    _load_synth_weights(model._model, 81)
    with torch.no_grad():
        model._model.embedding_head.conv_seediness.weight.mul_(6.0)      # spread, but do not saturate, the sigmoid (ties!)
    frames = synth.synth_frames(12, 96, 128, seed=81)
    subseqs, _ = get_subsequence_frames(12, 8, "ytvis", 4)
    res = model([f for f in frames], subseqs)
    fg_probs, mc = res["fg_masks"], res["multiclass_masks"]
    fg = (fg_probs > 0.5).byte()
    s0 = res["embeddings"][0].seediness
    min_seed = float(np.float32(s0.flatten().quantile(0.9)))
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, min_seed, 2, [0.3, 0.3], "cpu"), 4.0)
    dicts = [dict(frames=list(e.subseq_frames), embeddings=e.embeddings.clone(), bandwidths=e.bandwidths.clone(), seediness=e.seediness.clone())
             for e in res["embeddings"]]
    (track, counts, life), _, _, _, meta = ch.process(fg, dicts)
    out = {"subseqs": np.array(subseqs, np.int64), "min_seed": np.float64(min_seed),
           "fg_probs": fg_probs.numpy().reshape(-1)[::3].copy(), "fg_shape": np.array(fg.shape, np.int64),
           "fg_bits": np.packbits(fg.numpy().astype(bool).reshape(-1)), "multiclass": mc.numpy().astype(np.int8),
           "labels": np.concatenate([l.numpy() for l in track]).astype(np.int16),
           "pt_counts": np.array(sorted(counts.items()), np.int64).reshape(-1, 2),
           "lifetimes": np.array(sorted(life.items()), np.int64).reshape(-1, 2)}
    for i, e in enumerate(res["embeddings"]):
        out["c%d_emb" % i] = e.embeddings.numpy().reshape(-1)[::3].copy()
        out["c%d_seed" % i] = e.seediness.numpy().reshape(-1)[::3].copy()
        out["c%d_instance_labels" % i] = np.array(meta[i]["instance_labels"], np.int64)
    print("model_ytvis: fg %d of %d, tracks %s" % (int(fg.sum()), fg.numel(), sorted(counts.items())[:8]))
    _save("model_ytvis", **out)


# ------------------------------------------------------------------------------------------------
def gen_model_kitti():
    """KITTI-MOTS preset (kitti_mots_2.yaml: 'xyt' embeddings -- the time coordinate is part of the embedding, no free dims --
    in-head seediness, 3+1-channel semseg head, MIN_SEEDINESS_PROB from the config family) through the REFERENCE at a reduced,
    wide-aspect size: 14 frames, clips of 8 with overlap 4, fg from the semseg head, tracks stitched by the reference chainer."""
    import ref_shim
    cfg = ref_shim.install()
    import torch
    cfg.merge_from_file(os.path.join(ref_shim.REFERENCE_ROOT, "stemseg", "config", "kitti_mots_2.yaml"))
    cfg.INPUT.update_param("MIN_DIM", 96)
    cfg.INPUT.update_param("MAX_DIM", 320)
    cfg.MODEL.BACKBONE.update_param("TYPE", "R-50-FPN")
    from stemseg.modeling.inference_model import InferenceModel
    from stemseg.inference.clusterers import SequentialClustering
    from stemseg.inference.online_chainer import OnlineChainer
    from stemseg.inference.main import get_subsequence_frames
    model = InferenceModel(None, cpu_workers=0, preload_images=False, semseg_output_type="probs", resize_scale=1.0,
                           semseg_generation_on_gpu=False)
    _load_synth_weights(model._model, 91)
    with torch.no_grad():
        model._model.embedding_head.conv_seediness.weight.mul_(6.0)
    frames = synth.synth_frames(14, 60, 190, seed=91)                 # KITTI-like 1 : 3.2 aspect -> resized 96 x 304 -> padded 96 x 320
    subseqs, _ = get_subsequence_frames(14, 8, "kittimots", 4)
    res = model([f for f in frames], subseqs)
    fg_probs, mc = res["fg_masks"], res["multiclass_masks"]
    fg = (fg_probs > 0.5).byte()
    min_seed = float(np.float32(res["embeddings"][0].seediness.flatten().quantile(0.3)))
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, min_seed, 0, [], "cpu"), 1.0)
    dicts = [dict(frames=list(e.subseq_frames), embeddings=e.embeddings.clone(), bandwidths=e.bandwidths.clone(), seediness=e.seediness.clone())
             for e in res["embeddings"]]
    (track, counts, life), _, _, _, meta = ch.process(fg, dicts)
    out = {"subseqs": np.array(subseqs, np.int64), "min_seed": np.float64(min_seed),
           "fg_probs": fg_probs.numpy().reshape(-1)[::3].copy(), "fg_shape": np.array(fg.shape, np.int64),
           "fg_bits": np.packbits(fg.numpy().astype(bool).reshape(-1)), "class_probs": mc.numpy().reshape(-1)[::5].copy(),
           "labels": np.concatenate([l.numpy() for l in track]).astype(np.int16),
           "pt_counts": np.array(sorted(counts.items()), np.int64).reshape(-1, 2),
           "lifetimes": np.array(sorted(life.items()), np.int64).reshape(-1, 2)}
    for i, e in enumerate(res["embeddings"]):
        out["c%d_emb" % i] = e.embeddings.numpy().reshape(-1)[::3].copy()
        out["c%d_instance_labels" % i] = np.array(meta[i]["instance_labels"], np.int64)
    print("model_kitti: clips %s, fg %d of %d, tracks %s" % (subseqs, int(fg.sum()), fg.numel(), sorted(counts.items())[:8]))
    _save("model_kitti", **out)


# ------------------------------------------------------------------------------------------------
def _cluster_cases():
    """(name, emb[N,E], bw[N,Ev], seed[N,1], kwargs)"""
    cases = []
    # structured margin cases from the shared synthetic driver
    for K, seed in ((0, 0), (1, 1), (5, 2), (10, 3), (20, 4), (25, 5)):
        emb, bw, sd, fg = synth.synth_cluster_case(4, 20, 28, K, seed=seed)
        m = fg.astype(bool)
        e = np.stack([emb[c][m] for c in range(emb.shape[0])], 1)
        b = np.stack([bw[c][m] for c in range(bw.shape[0])], 1)
        s = sd[0][m][:, None]
        cases.append(("blobs_K%d" % K, e, b, s, dict(n_free_dims=2, free_dim_stds=[0.3, 0.3], label_start=1 + K)))
    # kitti-like: no free dims, E == Ev == 3, min seediness 0.95
    emb, bw, sd, fg = synth.synth_cluster_case(4, 16, 24, 6, E=3, Ev=3, seed=9)
    m = fg.astype(bool)
    cases.append(("kitti_like", np.stack([emb[c][m] for c in range(3)], 1), np.stack([bw[c][m] for c in range(3)], 1),
                  sd[0][m][:, None], dict(n_free_dims=0, free_dim_stds=[], label_start=7, min_seediness_prob=0.95)))
    # SURVEY.md A.2 probes (E=2, bw=1)
    x = np.array([0.0, 2.0, 1.7, 5.5, 3.4], np.float32)
    e = np.stack([x, np.zeros_like(x)], 1)
    s = np.array([0.99, 0.1, 0.1, 0.1, 0.98], np.float32)[:, None]
    cases.append(("quirk_max_distance", e, np.ones_like(e), s, dict(n_free_dims=0, free_dim_stds=[], label_start=1)))
    x = np.array([0, 1, 1.5, 2, 2.5, 9], np.float32)
    e = np.stack([x, np.zeros_like(x)], 1)
    s = np.array([0.99, 0.2, 0.98, 0.2, 0.2, 0.97], np.float32)[:, None]
    cases.append(("quirk_stale_mask", e, np.ones_like(e), s,
                  dict(n_free_dims=0, free_dim_stds=[], label_start=5, max_instances=2)))
    # all seeds below threshold, ties in seediness (first index must win), single point
    rng = np.random.RandomState(77)
    e = rng.standard_normal((50, 4)).astype(np.float32)
    cases.append(("all_low_seed", e, np.full((50, 2), 20, np.float32), np.full((50, 1), 0.5, np.float32),
                  dict(n_free_dims=2, free_dim_stds=[0.3, 0.3], label_start=1)))
    e = np.concatenate([rng.standard_normal((30, 2)) * 0.05, 4 + rng.standard_normal((30, 2)) * 0.05]).astype(np.float32)
    s = np.full((60, 1), 0.9, np.float32)      # every seediness equal -> argmax tie-break = lowest index
    cases.append(("seed_ties", e, np.full((60, 2), 30, np.float32), s,
                  dict(n_free_dims=0, free_dim_stds=[], label_start=3)))
    cases.append(("single_point", np.zeros((1, 4), np.float32), np.ones((1, 2), np.float32),
                  np.full((1, 1), 0.99, np.float32), dict(n_free_dims=2, free_dim_stds=[0.3, 0.3], label_start=1)))
    cases.append(("empty", np.zeros((0, 4), np.float32), np.zeros((0, 2), np.float32),
                  np.zeros((0, 1), np.float32), dict(n_free_dims=2, free_dim_stds=[0.3, 0.3], label_start=1)))
    # secondary assignment actually firing with K == 1 (ring of points between the two thresholds)
    ang = np.linspace(0, 2 * np.pi, 40, endpoint=False)
    ring = np.stack([1.9 * np.cos(ang), 1.9 * np.sin(ang)], 1)
    core = rng.standard_normal((40, 2)) * 0.1
    e = np.concatenate([core, ring]).astype(np.float32)
    s = np.concatenate([np.full(40, 0.95), np.full(40, 0.1)]).astype(np.float32)[:, None]
    s[0] = 0.99
    cases.append(("secondary_K1", e, np.ones_like(e), s, dict(n_free_dims=0, free_dim_stds=[], label_start=1)))
    # dense random cloud near the thresholds ("adversarial": compared with a tolerance band)
    e = (rng.standard_normal((4000, 4)) * 0.6).astype(np.float32)
    b = (2.0 + rng.uniform(0, 2, (4000, 2))).astype(np.float32)
    s = rng.uniform(0.5, 1.0, (4000, 1)).astype(np.float32)
    cases.append(("adversarial_cloud", e, b, s, dict(n_free_dims=2, free_dim_stds=[0.5, 0.5], label_start=1)))
    return cases


def gen_cluster():
    import ref_shim
    ref_shim.install()
    import torch
    from stemseg.inference.clusterers import SequentialClustering
    out = {}
    names = []
    for name, e, b, s, kw in _cluster_cases():
        kw = dict(kw)
        cl = SequentialClustering(0.5, 0.3, kw.pop("min_seediness_prob", 0.8), kw["n_free_dims"], kw["free_dim_stds"],
                                  "cpu", max_instances=kw.pop("max_instances", 20))
        labels, meta = cl(torch.from_numpy(e), bandwidths=torch.from_numpy(b), seediness=torch.from_numpy(s),
                          cluster_label_start=kw["label_start"], return_label_masks=True)
        names.append(name)
        out[name + "__emb"], out[name + "__bw"], out[name + "__seed"] = e, b, s
        out[name + "__labels"] = labels.numpy()
        out[name + "__instance_labels"] = np.array(meta["instance_labels"], np.int64)
        E = e.shape[1]
        out[name + "__centers"] = np.array(meta["instance_centers"], np.float32).reshape(-1, E)
        out[name + "__stds"] = np.array(meta["instance_stds"], np.float32).reshape(-1, E)
        out[name + "__masks"] = (np.stack([m.numpy() for m in meta["instance_masks"]]) if meta["instance_masks"]
                                 else np.zeros((0, e.shape[0]), bool))
        out[name + "__params"] = np.array([cl.min_seediness_prob, cl.max_instances, kw["label_start"],
                                           kw["n_free_dims"]] + list(kw["free_dim_stds"]), np.float64)
    out["__names"] = np.array(names)
    _save("cluster", **out)


# ------------------------------------------------------------------------------------------------
def _chainer_sequence(n_frames, H, W, seed):
    """Sequence-level structured head outputs with instance births/deaths.  Returns per-frame maps
    emb[4,F,H,W], bw[2,F,H,W], seed[1,F,H,W], fg[F,H,W]."""
    rng = np.random.RandomState(seed)
    emb = (10 + 3 * rng.standard_normal((4, n_frames, H, W))).astype(np.float32)
    bw = (25 + rng.uniform(0, 1, (2, n_frames, H, W))).astype(np.float32)
    sd = rng.uniform(0, 0.2, (1, n_frames, H, W)).astype(np.float32)
    fg = np.zeros((n_frames, H, W), np.uint8)
    inst = [  # y0, x0, h, w, first, last, free-dim code
        (1, 1, 5, 6, 0, n_frames - 1, (-0.6, 0.6)),
        (9, 2, 5, 5, 0, 9, (0.6, 0.6)),
        (2, 14, 6, 6, 3, n_frames - 1, (0.6, -0.6)),
        (10, 15, 4, 7, 10, n_frames - 1, (-0.6, -0.6)),
        (7, 9, 3, 3, 6, 13, (0.0, 1.2)),
    ]
    for (y0, x0, h, w, a, b, free) in inst:
        for t in range(a, min(b, n_frames - 1) + 1):
            cy, cx = -1 + 2 * (y0 + h / 2) / H, -1.3 + 2.6 * (x0 + w / 2) / W
            c = np.array([cy, cx, free[0], free[1]], np.float32)
            nz = (0.04 * rng.standard_normal((4, h, w))).astype(np.float32)
            emb[:, t, y0:y0 + h, x0:x0 + w] = c[:, None, None] + nz
            sd[0, t, y0:y0 + h, x0:x0 + w] = np.clip(1 - np.sqrt((nz ** 2).sum(0)), 0, 1)
            fg[t, y0:y0 + h, x0:x0 + w] = 1
    fg = np.where(rng.uniform(size=fg.shape) < 0.02, 1, fg).astype(np.uint8)
    return emb, bw, sd, fg


def gen_chainer():
    import ref_shim
    ref_shim.install()
    import torch
    from stemseg.inference.clusterers import SequentialClustering
    from stemseg.inference.online_chainer import OnlineChainer
    from stemseg.inference.main import get_subsequence_frames
    out = {}
    for tag, n_frames, overlap, seed in (("seq20_ov4", 20, 4, 31), ("seq14_ov6", 14, 6, 32), ("seq8_single", 8, 4, 33)):
        H, W = 16, 24
        emb, bw, sd, fg = _chainer_sequence(n_frames, H, W, seed)
        subseqs, _ = get_subsequence_frames(n_frames, 8, "davis", overlap)
        dicts = [dict(frames=list(fr), embeddings=torch.from_numpy(emb[:, fr].copy()),
                      bandwidths=torch.from_numpy(bw[:, fr].copy()), seediness=torch.from_numpy(sd[:, fr].copy()))
                 for fr in subseqs]
        ch = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), 1.0)
        (track_labels, pt_counts, lifetimes), mask_idxes, subseq_labels, _, meta = ch.process(torch.from_numpy(fg), dicts)
        out[tag + "__seed"] = np.int64(seed)
        out[tag + "__subseqs"] = np.array(subseqs, np.int64)
        out[tag + "__emb"], out[tag + "__bw"], out[tag + "__sd"], out[tag + "__fg"] = emb, bw, sd, fg
        for t, l in enumerate(track_labels):
            out["%s_track_%02d" % (tag, t)] = l.numpy()
        out[tag + "__pt_counts"] = np.array(sorted(pt_counts.items()), np.int64).reshape(-1, 2)
        out[tag + "__lifetimes"] = np.array(sorted(lifetimes.items()), np.int64).reshape(-1, 2)
        for i, ls in enumerate(subseq_labels):
            out["%s_clip%d_labels" % (tag, i)] = np.concatenate([l.numpy() for l in ls]) if ls else np.zeros(0, np.int64)
            out["%s_clip%d_instance_labels" % (tag, i)] = np.array(meta[i]["instance_labels"], np.int64)
    # resize path (online_chainer.py:127-140): x4 trilinear of emb / seed / (activated) bw
    rng = np.random.RandomState(5)
    e = rng.standard_normal((4, 2, 5, 7)).astype(np.float32)
    sub = {"embeddings": torch.from_numpy(e), "seediness": torch.from_numpy(e[:1].copy()), "bandwidths": torch.from_numpy(e[:2].copy())}
    OnlineChainer(None, 4.0).resize_tensors(sub)
    out["resize__in"] = e
    out["resize__emb"] = sub["embeddings"].numpy()
    _save("chainer", **out)


def gen_chainer_long():
    """48 clips whose track ids climb past 190 (tests/synth.synth_long_sequence) through the reference's chainer: the
    regime where a max-id-sized association table breaks.  Only the OUTPUTS are stored (the inputs are regenerated from
    the seed; their checksums are kept to detect drift)."""
    import ref_shim
    ref_shim.install()
    import zlib
    import torch
    from stemseg.inference.clusterers import SequentialClustering
    from stemseg.inference.online_chainer import OnlineChainer
    from stemseg.inference.main import get_subsequence_frames
    n_clips, seed = 48, 3
    emb, bw, sd, fg = synth.synth_long_sequence(n_clips, seed=seed)
    F = fg.shape[0]
    subseqs, _ = get_subsequence_frames(F, 8, "kittimots", 4)
    assert len(subseqs) == n_clips
    dicts = [dict(frames=list(fr), embeddings=torch.from_numpy(emb[:, fr].copy()), bandwidths=torch.from_numpy(bw[:, fr].copy()),
                  seediness=torch.from_numpy(sd[:, fr].copy())) for fr in subseqs]
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), 1.0)
    (track_labels, pt_counts, lifetimes), _, subseq_labels, _, meta = ch.process(torch.from_numpy(fg), dicts)
    out = {"n_clips": np.int64(n_clips), "seed": np.int64(seed),
           "input_crc": np.array([zlib.crc32(a.tobytes()) for a in (emb, bw, sd, fg)], np.int64),
           "track_sizes": np.array([l.numel() for l in track_labels], np.int64),
           "track_labels": np.concatenate([l.numpy() for l in track_labels]).astype(np.int32),
           "pt_counts": np.array(sorted(pt_counts.items()), np.int64).reshape(-1, 2),
           "lifetimes": np.array(sorted(lifetimes.items()), np.int64).reshape(-1, 2),
           "instance_label_sizes": np.array([len(m["instance_labels"]) for m in meta], np.int64),
           "instance_labels": np.concatenate([np.array(m["instance_labels"], np.int64) for m in meta])}
    print("chainer_long: %d frames, %d clips, highest track id %d" % (F, n_clips, max(pt_counts)))
    assert max(pt_counts) > 150
    _save("chainer_long", **out)


def gen_chainer_ties():
    """Exact Hungarian cost ties (tests/synth.synth_tie_sequence) through the reference's chainer: which old track a new
    instance joins depends on the reference's id enumeration order, list(set(unique) - {-1}) (online_chainer.py:308-309).  Also
    stores direct associate_clusters cases on label arrays with large, colliding ids.  Checks at generation time that the
    sequence IS order-sensitive: this repo's chain with ascending ids gives another result."""
    import ref_shim
    ref_shim.install()
    import zlib
    import torch
    from stemseg.inference.clusterers import SequentialClustering
    from stemseg.inference.online_chainer import OnlineChainer
    n_clips, seed = 12, 1
    per_clip, fg, clips = synth.synth_tie_sequence(n_clips, seed=seed)
    dicts = [dict(frames=list(fr), embeddings=torch.from_numpy(e.copy()), bandwidths=torch.from_numpy(b.copy()), seediness=torch.from_numpy(s.copy()))
             for fr, (e, b, s) in zip(clips, per_clip)]
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), 1.0)
    (track_labels, pt_counts, lifetimes), _, subseq_labels, _, meta = ch.process(torch.from_numpy(fg), dicts)
    out = {"n_clips": np.int64(n_clips), "seed": np.int64(seed),
           "input_crc": np.array([zlib.crc32(np.concatenate([a.reshape(-1) for a in pc]).tobytes()) for pc in per_clip] + [zlib.crc32(fg.tobytes())], np.int64),
           "track_sizes": np.array([l.numel() for l in track_labels], np.int64),
           "track_labels": np.concatenate([l.numpy() for l in track_labels]).astype(np.int32),
           "pt_counts": np.array(sorted(pt_counts.items()), np.int64).reshape(-1, 2),
           "lifetimes": np.array(sorted(lifetimes.items()), np.int64).reshape(-1, 2),
           "instance_label_sizes": np.array([len(m["instance_labels"]) for m in meta], np.int64),
           "instance_labels": np.concatenate([np.array(m["instance_labels"], np.int64) for m in meta])}
    # direct association cases: ids that collide in CPython's set table, with and without the outlier id
    rs = np.random.RandomState(4)
    n_cases = 0
    for case in range(40):
        k1, k2 = rs.randint(2, 7), rs.randint(1, 6)
        ids = rs.choice(np.arange(1, 400), k1 + k2, replace=False)
        ids1, ids2 = ids[:k1], ids[k1:]
        n = 600
        l1 = rs.choice(ids1, n).astype(np.int64)
        l2 = np.full(n, -1, np.int64)
        # a few real overlaps, the rest of the new ids on points where labels_1 is the outlier: zero-IoU rows AND columns
        matched = rs.permutation(min(k1, k2))[:rs.randint(0, min(k1, k2))]
        for m in matched:
            l2[l1 == ids1[m]] = ids2[m]
        free2 = [i for i in range(k2) if i not in set(matched.tolist())]
        out_pts = rs.uniform(size=n) < 0.25
        l1[out_pts] = -1
        l2[out_pts] = rs.choice(ids2[free2], int(out_pts.sum())) if free2 else -1
        if rs.uniform() < 0.3:
            l1[l1 == -1] = ids1[0]                              # no outlier id on side 1
        assoc = ch.associate_clusters(torch.from_numpy(l1), torch.from_numpy(l2))[0]
        out["assoc%02d_l1" % case], out["assoc%02d_l2" % case] = l1.astype(np.int32), l2.astype(np.int32)
        out["assoc%02d_pairs" % case] = np.array(assoc, np.int64).reshape(-1, 2)
        n_cases += 1
    out["n_assoc"] = np.int64(n_cases)
    # order sensitivity: the product's chain (oracle-backed ops) reproduces the reference; with ascending ids it does not
    sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))
    from stemseg_amd.inference import online_chainer as oc
    from stemseg_amd.inference.clusterers import SequentialClustering as SC
    from tests.oracle_ops import OracleChainerOps

    def ours():
        d2 = [dict(frames=list(fr), embeddings=torch.from_numpy(e.copy()), bandwidths=torch.from_numpy(b.copy()), seediness=torch.from_numpy(s.copy()))
              for fr, (e, b, s) in zip(clips, per_clip)]
        return oc.OnlineChainer(SC(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cpu"), 1.0, ops=OracleChainerOps()).process(torch.from_numpy(fg), d2)[0][0]
    same = all(np.array_equal(a.numpy(), b.numpy()) for a, b in zip(ours(), track_labels))
    keep = oc.reference_id_order
    oc.reference_id_order = lambda ids, has_outlier: sorted(ids)
    same_sorted = all(np.array_equal(a.numpy(), b.numpy()) for a, b in zip(ours(), track_labels))
    oc.reference_id_order = keep
    print("chainer_ties: %d clips, highest id %d; product chain == reference: %s; with ascending ids: %s" % (n_clips, max(pt_counts), same, same_sorted))
    assert same and not same_sorted, "the tie sequence must be order-sensitive and reproduced"
    _save("chainer_ties", **out)


# ------------------------------------------------------------------------------------------------
def gen_misc():
    import ref_shim
    cfg = ref_shim.install()
    import torch
    from stemseg.inference.main import get_subsequence_frames
    from stemseg.modeling.embedding_utils import (creat_spatiotemporal_grid, add_spatiotemporal_offset,
                                                  get_nb_embedding_dims, get_nb_free_dims)
    from stemseg.data.common import compute_resize_params_2
    from stemseg.data.inference_image_loader import InferenceImageLoader, collate_fn
    out = {}
    # windowing table (inference/main.py:23-49)
    rows = []
    for ds in ("davis", "ytvis", "kittimots"):
        for seq_len in (3, 5, 8, 9, 12, 20, 36, 64):
            for T in (8, 16):
                for ov in (-1, 1, 4, 6):
                    subseqs, padded = get_subsequence_frames(seq_len, T, ds, ov)
                    key = "win_%s_%d_%d_%d" % (ds, seq_len, T, ov)
                    out[key] = np.array(subseqs, np.int64)
                    out[key + "__padded"] = np.array(padded if padded is not None else [], np.int64)
                    rows.append(key)
    out["win__keys"] = np.array(rows)
    # coordinate grids (embedding_utils.py:28-41)
    for (H, W, T) in ((16, 24, 8), (120, 216, 8), (152, 488, 8), (7, 5, 4), (24, 16, 16), (1, 9, 2)):
        t, y, x = creat_spatiotemporal_grid(H, W, T, 1.0)
        out["grid_%d_%d_%d_t" % (H, W, T)] = t[:, 0, 0].numpy()
        out["grid_%d_%d_%d_y" % (H, W, T)] = y[0, :, 0].numpy()
        out["grid_%d_%d_%d_x" % (H, W, T)] = x[0, 0, :].numpy()
    modes = ["xy", "ff", "xyt", "xyf", "xytf", "xyff", "xytff", "xyfff"]
    out["modes"] = np.array(modes)
    out["modes__nb_dims"] = np.array([get_nb_embedding_dims(m) for m in modes], np.int64)
    out["modes__nb_free"] = np.array([get_nb_free_dims(m) for m in modes], np.int64)
    for m in modes:
        z = torch.zeros(1, get_nb_embedding_dims(m), 3, 4, 6)
        out["offset_" + m] = add_spatiotemporal_offset(z, torch.tensor(1.0), m)[0].numpy()
    # resize parameters (data/common.py:142-159) and pad-to-32 rule (structures/image_list.py:93-95)
    dims = [(854, 480), (640, 360), (1242, 375), (96, 64), (70, 50), (1280, 720), (500, 333)]
    cfgs = [(480, 854), (736, 1248), (640, 1196), (800, 1948), (64, 96), (800, 1333)]
    tab = []
    for (w, h) in dims:
        for (mn, mx) in cfgs:
            nw, nh, _ = compute_resize_params_2((w, h), mn, mx)
            tab.append([w, h, mn, mx, nw, nh])
    out["resize_params"] = np.array(tab, np.int64)
    # preprocessing of one odd-sized frame (inference_image_loader.py:23-43) incl. padding by collate_fn
    cfg.INPUT.update_param("MIN_DIM", 64)
    cfg.INPUT.update_param("MAX_DIM", 96)
    img = synth.synth_frames(1, 50, 70, seed=3)[0]
    loader = InferenceImageLoader([img])
    il, _ = collate_fn([loader[0]])
    out["preproc__in"] = img
    out["preproc__out"] = il.tensors[0, 0].numpy()
    _save("misc", **out)


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group", default=None, choices=GROUPS)
    args = ap.parse_args()
    if args.group is None:
        for g in GROUPS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--group", g])
        return
    g = args.group
    if g.startswith("dec_T"):
        gen_decoders(int(g[5:]))
    elif g == "semseg":
        gen_semseg()
    elif g == "masks":
        gen_masks()
    elif g == "config0":
        gen_config0()
    elif g == "model_ytvis":
        gen_model_ytvis()
    elif g == "model_kitti":
        gen_model_kitti()
    elif g == "encoder":
        gen_encoder()
    elif g == "model_davis":
        gen_model_davis()
    elif g == "cluster":
        gen_cluster()
    elif g == "chainer":
        gen_chainer()
    elif g == "chainer_long":
        gen_chainer_long()
    elif g == "chainer_ties":
        gen_chainer_ties()
    elif g == "misc":
        gen_misc()


if __name__ == "__main__":
    main()
