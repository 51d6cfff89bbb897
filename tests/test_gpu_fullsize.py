"""GPU parity at BASELINE.json's OWN sizes (the reduced-size twins live in test_gpu_parity.py):

  configs[1]  DAVIS-shape clip, T=8, 480x854 -> 480x864, ResNet-101-FPN, embedding + seediness decoders, clustering
  configs[2]  YouTube-VIS secondary shape 640x1152 with --resize_embeddings: x4 trilinear of the head outputs and
              clustering at FULL resolution (5.9 M voxels)
  configs[3]  64-frame sequence, overlap 4 -> 15 clips at 480x864 through run_sequence_sharded (one process) and through
              InferenceModel.forward's feature cache, stitched, vs the oracle chain on the same head outputs
  configs[4]  KITTI-MOTS preset (xyt embeddings, in-head seediness, 3+1-channel semseg head), --max_dim 1948 ->
              608x1952, h4 x w4 = 152x488

Float outputs <= 1e-3 absolute (BASELINE.json north_star; bandwidths relative), integer outputs identical wherever the
oracle's own decision has margin (SURVEY.md A.2 'bit-exactness test design').  Every call goes through the C-ABI.
"""
import time

import numpy as np
import pytest
import torch

from oracle import decoder as odec
from oracle import pipeline as opipe
from oracle.clusterer import sequential_clustering
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from stemseg_amd import hip as h
    h.require_gpu()
    return h


def _maxerr(name, got, ref):
    err = float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max())
    print("[fullsize] %-34s max|err| %.3e  (max|ref| %.3g)" % (name, err, float(np.abs(np.asarray(ref)).max())))
    return err


def _model(preset, backbone, seed, gains, min_dim, max_dim, **kw):
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    config.load_preset(preset)
    config.cfg.MODEL.BACKBONE.TYPE = backbone
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = min_dim, max_dim
    model = InferenceModel(**kw)
    names = [(k, v.shape) for k, v in model._model.state_dict().items()]
    sd = synth.synth_state_dict(names, seed)
    for k, g in gains.items():
        sd[k] = sd[k] * g
    model._model.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(model._model.state_dict()[k].shape) for k, v in sd.items()})
    return model.cuda(), sd


def _frames(T, H, W, valid_w, seed):
    """Mean-subtracted float frames [T,3,H,W]; columns >= valid_w are the zero padding of image_list.py:93-104."""
    from stemseg_amd import config
    x = torch.from_numpy(synth.synth_frames(T, H, W, seed=seed).astype(np.float32)).permute(0, 3, 1, 2) - \
        torch.tensor(config.cfg.INPUT.IMAGE_MEAN)[None, :, None, None]
    x[..., valid_w:] = 0
    return x.contiguous()


def _assert_exact_or_in_band(what, got, ref_labels, ref_meta, band):
    """Integer outputs: identical, or every mismatch is a point whose ORACLE probability sits within ``band`` of a threshold
    (0.5 primary, 0.3 secondary) in some round -- SURVEY.md A.2 'bit-exactness test design' (libm vs GPU exp / sqrt last-ulp
    differences: band 2e-6 on identical inputs; 1e-4 when the float maps themselves differ by the 1e-6 of two fp32 pipelines)."""
    bad = np.flatnonzero(np.asarray(got) != np.asarray(ref_labels))
    print("[fullsize] %s: %d / %d labels differ" % (what, bad.size, np.asarray(ref_labels).size))
    if bad.size:
        P = np.stack(ref_meta["instance_probs"])
        near = (np.abs(P - 0.5) < band).any(0) | (np.abs(P - 0.3) < band).any(0)
        assert np.all(near[bad]), "%s: %d label mismatches away from the threshold band" % (what, int((~near[bad]).sum()))
    return bad.size


def _labels_on_grid(fg, labels):
    out = np.full(fg.size, -2, np.int64)
    out[np.flatnonzero(np.asarray(fg).reshape(-1))] = np.asarray(labels)
    return out


# ------------------------------------------------------------------------------------------------ configs[1]
def test_config1_davis_r101_clip_480x864_vs_oracle(hip):
    from stemseg_amd import config
    from stemseg_amd.pipeline import ClipPipeline
    try:
        model, sd = _model("davis", "R-101-FPN", 1234, {"seediness_head.conv_out.weight": 30.0}, 480, 854)
        pipe = ClipPipeline(model)
        frames = _frames(8, 480, 864, 854, seed=11)
        out = pipe.step(frames.cuda())
        torch.cuda.synchronize()
        t0 = time.time()
        ref = opipe.embed_and_cluster_clip(frames, sd, "R-101-FPN", "xyff", 4, True, free_dim_stds=[0.3, 0.3], return_probs=True)
        print("[fullsize] oracle clip took %.1f s" % (time.time() - t0))
        assert _maxerr("config1 emb", out["emb"].cpu().numpy(), ref["emb"].numpy()) <= 1e-3
        assert _maxerr("config1 seediness", out["seed"].cpu().numpy(), ref["seed"].numpy()) <= 1e-3
        assert _maxerr("config1 bandwidth (rel)", (out["bw"].cpu() / ref["bw"]).numpy(), np.ones(ref["bw"].shape)) <= 1e-3
        g_fg, c_fg = out["fg"].cpu().numpy().astype(bool), ref["fg"].numpy().astype(bool)
        n = int(out["frame_offsets"].cpu()[-1])
        meta = hip.read_cluster_meta(out["meta"])
        g_lab = _labels_on_grid(g_fg, out["labels"][:n].cpu().numpy())
        c_lab = _labels_on_grid(c_fg, ref["labels"])
        both = (g_fg & c_fg).reshape(-1)
        agree = float((g_lab[both] == c_lab[both]).mean())
        print("[fullsize] config1: fg %d vs %d (%d differ), K %d vs %d, labels identical on %.5f of the common fg"
              % (g_fg.sum(), c_fg.sum(), (g_fg != c_fg).sum(), meta.K, len(ref["meta"]["instance_labels"]), agree))
        assert n == g_fg.sum() > 100000 and (g_fg != c_fg).mean() < 1e-4
        assert meta.K == len(ref["meta"]["instance_labels"]) >= 10
        if np.array_equal(g_fg, c_fg):           # same points on both sides: every label identical, or inside the threshold band
            _assert_exact_or_in_band("config1 HIP maps vs oracle maps", out["labels"][:n].cpu().numpy(), ref["labels"], ref["meta"], 1e-4)
        else:
            assert agree >= 0.9999
        # integer bookkeeping proper: the oracle's float maps through the HIP gather + clusterer -> the oracle's labels
        o2 = pipe.cluster(ref["emb"].cuda().contiguous(), ref["bw"].cuda().contiguous(), ref["seed"].cuda().contiguous())
        n2 = int(o2["frame_offsets"].cpu()[-1])
        assert n2 == ref["labels"].shape[0]
        _assert_exact_or_in_band("config1 clusterer on the oracle's maps", o2["labels"][:n2].cpu().numpy(), ref["labels"], ref["meta"], 2e-6)
    finally:
        config.load_preset("defaults")


def test_config1_davis_reference_resize_704x1248_vs_oracle(hip):
    """The reference-faithful DAVIS resize (davis_1.yaml: MIN_DIM 736 / MAX_DIM 1248 -> 480x854 frames become 701x1248, padded
    704x1248; SURVEY 8(d) 'secondary' shape): h4 x w4 = 176 x 312, 439 k voxels per map.  R-50 weights (the shape is what is
    under test); encoder + both decoders + clustering vs the oracle."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import compute_resize_params_2, pad_to_multiple_of_32
    from stemseg_amd.pipeline import ClipPipeline
    try:
        nw, nh, _ = compute_resize_params_2((854, 480), 736, 1248)
        H, W = pad_to_multiple_of_32(nh, nw)
        assert (nh, nw, H, W) == (701, 1248, 704, 1248)
        model, sd = _model("davis", "R-50-FPN", 21, {"seediness_head.conv_out.weight": 30.0}, 736, 1248)
        pipe = ClipPipeline(model)
        frames = _frames(8, H, W, nw, seed=12)
        frames[:, :, nh:] = 0
        out = pipe.step(frames.cuda())
        torch.cuda.synchronize()
        t0 = time.time()
        ref = opipe.embed_and_cluster_clip(frames, sd, "R-50-FPN", "xyff", 4, True, free_dim_stds=[0.3, 0.3], return_probs=True)
        print("[fullsize] oracle 704x1248 clip took %.1f s" % (time.time() - t0))
        assert tuple(out["emb"].shape) == (4, 8, 176, 312)
        assert _maxerr("davis 704x1248 emb", out["emb"].cpu().numpy(), ref["emb"].numpy()) <= 1e-3
        assert _maxerr("davis 704x1248 seediness", out["seed"].cpu().numpy(), ref["seed"].numpy()) <= 1e-3
        assert _maxerr("davis 704x1248 bandwidth (rel)", (out["bw"].cpu() / ref["bw"]).numpy(), np.ones(ref["bw"].shape)) <= 1e-3
        o2 = pipe.cluster(ref["emb"].cuda().contiguous(), ref["bw"].cuda().contiguous(), ref["seed"].cuda().contiguous())
        n2 = int(o2["frame_offsets"].cpu()[-1])
        assert n2 == ref["labels"].shape[0]
        print("[fullsize] davis 704x1248: %d fg points, K %d" % (n2, len(ref["meta"]["instance_labels"])))
        _assert_exact_or_in_band("davis 704x1248 clusterer on the oracle's maps", o2["labels"][:n2].cpu().numpy(), ref["labels"], ref["meta"], 2e-6)
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ configs[4]
def test_config4_kitti_clip_608x1952_vs_oracle(hip):
    """One KITTI-MOTS-preset clip at the --max_dim 1948 size: 375x1242 frames -> 588x1948 -> padded 608x1952 (h4 x w4 =
    152 x 488, 593 k voxels per map, workspaces ~4x the DAVIS ones): encoder, embedding decoder (xyt, E = Ev = 3, in-head
    seediness), semseg decoder (3+1 channels), fg from the semseg head, clustering."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import compute_resize_params_2, pad_to_multiple_of_32
    try:
        # inference/main.py:217-221: --max_dim 1948 -> MIN_DIM = int(round(1948 / (1792 / 736))) = 800
        nw, nh, _ = compute_resize_params_2((1242, 375), int(round(1948 / (1792 / 736))), 1948)
        H, W = pad_to_multiple_of_32(nh, nw)
        assert (nh, nw, H, W) == (588, 1948, 608, 1952)
        model, sd = _model("kittimots", "R-50-FPN", 91, {"embedding_head.conv_seediness.weight": 8.0}, 800, 1948, semseg_output_type="probs")
        frames = _frames(8, H, W, nw, seed=41)
        frames[:, :, nh:] = 0
        emb, bw, seed = model.embed_frames(frames.cuda())
        logits = model.semseg_logits_clip(8, H, W, emb.device)
        torch.cuda.synchronize()
        assert tuple(emb.shape) == (3, 8, 152, 488) and tuple(bw.shape) == (3, 8, 152, 488) and tuple(logits.shape) == (4, 8, 152, 488)
        t0 = time.time()
        from oracle import encoder as oenc
        feats = oenc.resnet_fpn(frames, sd, "R-50-FPN")
        stacks = [feats[s].permute(1, 0, 2, 3).contiguous() for s in (32, 16, 8, 4)]
        o = odec.embedding_decoder(stacks, sd, "xyt", True)
        r_emb, r_bw, r_seed = o[:3], opipe.bandwidth_activation(o[3:6]), o[6:7]
        r_logits = odec.semseg_decoder(stacks, sd)
        print("[fullsize] oracle KITTI clip took %.1f s" % (time.time() - t0))
        assert _maxerr("config4 emb", emb.cpu().numpy(), r_emb.numpy()) <= 1e-3
        assert _maxerr("config4 seediness", seed.cpu().numpy(), r_seed.numpy()) <= 1e-3
        assert _maxerr("config4 bandwidth (rel)", (bw.cpu() / r_bw).numpy(), np.ones(r_bw.shape)) <= 1e-3
        assert _maxerr("config4 semseg logits", logits.cpu().numpy(), r_logits.numpy()) <= 1e-3
        # fg mask from the semseg head (inference_model.py:197-231, main.py:142-144), on both sides from their own logits
        acc = logits.permute(1, 0, 2, 3).contiguous()
        fg_p, _ = hip.semseg_masks(acc, torch.ones(8, device=acc.device), "probs")
        r_fg_p, _ = odec.semseg_masks(r_logits.permute(1, 0, 2, 3), "probs")
        assert _maxerr("config4 fg probability", fg_p.cpu().numpy(), r_fg_p.numpy()) <= 1e-3
        r_fg = (r_fg_p > 0.5).numpy().astype(np.uint8)
        fg = torch.stack([hip.fg_mask(p.contiguous(), 1.0, 0.5) for p in fg_p], 0)
        assert (fg.cpu().numpy() != r_fg).mean() < 1e-4
        # clustering at this N: the oracle's maps + mask through the HIP gather / clusterer vs the oracle's labels
        min_seed = float(np.quantile(r_seed.numpy()[0][r_fg.astype(bool)], 0.9)) if r_fg.any() else 0.5
        ref_lab, ref_meta, _ = opipe.cluster_clip(r_emb, r_bw, r_seed, r_fg, min_seediness=min_seed, free_dim_stds=[], return_probs=True)
        from stemseg_amd.inference.clusterers import SequentialClustering
        cl = SequentialClustering(0.5, 0.3, min_seed, 0, [], "cuda:0")
        e, b, s, vox, offs = hip.fg_gather(r_emb.cuda().contiguous(), r_bw.cuda().contiguous(), r_seed.cuda().contiguous(), torch.from_numpy(r_fg).cuda())
        labels, meta_dev, _, _ = cl.enqueue(e, b, s, 1, offs[8:])
        n = int(offs.cpu()[-1])
        meta = hip.read_cluster_meta(meta_dev)
        print("[fullsize] config4: %d fg points, K %d vs %d" % (n, meta.K, len(ref_meta["instance_labels"])))
        assert n == ref_lab.shape[0] and meta.K == len(ref_meta["instance_labels"])
        _assert_exact_or_in_band("config4 clusterer on the oracle's maps", labels[:n].cpu().numpy(), ref_lab, ref_meta, 2e-6)
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ configs[2]
def test_config2_ytvis_640x1152_resize_and_cluster_full_resolution(hip):
    """YouTube-VIS secondary shape (640x1138 -> 640x1152) under --resize_embeddings: the chainer resizes emb / seediness /
    (activated) bandwidths x4 to 640x1152 (online_chainer.py:127-140) and clusters every foreground voxel of the 8 frames at
    full resolution -- 5.9 M voxels, ~1.2 M foreground points here, labels int64.  Structured head outputs with 12
    instances; vs torch-CPU trilinear + the oracle's clusterer."""
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    emb, bw, sd, fg4 = synth.synth_cluster_case(8, 160, 288, 12, seed=57, bg_fraction=0.1)
    fg = np.repeat(np.repeat(fg4, 4, axis=1), 4, axis=2)
    ch = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cuda:0"), 4.0)
    sub = dict(frames=list(range(8)), embeddings=torch.from_numpy(emb).cuda(), bandwidths=torch.from_numpy(bw).cuda(),
               seediness=torch.from_numpy(sd).cuda())
    (track, counts, life), mask_idxes, _, _, meta = ch.process(torch.from_numpy(fg), [sub])
    torch.cuda.synchronize()
    up = lambda a: torch.nn.functional.interpolate(torch.from_numpy(a)[None], scale_factor=(1.0, 4.0, 4.0), mode="trilinear",  # noqa: E731
                                                   align_corners=False)[0].numpy()
    t0 = time.time()
    e, b, s, cnt = opipe.gather_fg(up(emb), up(bw), up(sd), fg)
    ref, ref_meta = sequential_clustering(e, b, s, label_start=1, free_dim_stds=[0.3, 0.3])
    print("[fullsize] oracle resize + clustering of %d points took %.1f s" % (e.shape[0], time.time() - t0))
    got = torch.cat([t.cpu() for t in track]).numpy()
    bad = int((got != ref).sum())
    print("[fullsize] config2 full-resolution clustering: %d points, K %d, %d labels differ" % (ref.size, len(ref_meta["instance_labels"]), bad))
    assert got.shape == ref.shape and ref.size > 1_000_000
    assert meta[0]["instance_labels"] == ref_meta["instance_labels"] and len(ref_meta["instance_labels"]) >= 12
    assert bad == 0                                   # the trilinear kernel is bit-identical to torch-CPU, the clusterer exact
    assert [int(c) for c in cnt] == [t.numel() for t in track]


# ------------------------------------------------------------------------------------------------ configs[3]
def test_config3_64_frames_15_clips_480x864_sharded_and_cached(hip):
    """64 DAVIS-shape frames, overlap 4 -> 15 clips (get_subsequence_frames, main.py:23-49), R-101 at 480x864:
      (i)  run_sequence_sharded in one process (the code path the 8-GPU run takes, world 1: no collective),
      (ii) TrackGenerator / InferenceModel.forward with the cross-clip feature cache (inference_model.py:83-108),
    both vs the oracle chain (numpy clusterer, Hungarian stitching) on the SAME head outputs: tracks, counts, lifetimes and
    per-clip instance lists identical.  Head outputs of (i) and (ii) agree to float tolerance (different launch shapes)."""
    from stemseg_amd import config
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.main import TrackGenerator, get_subsequence_frames
    from stemseg_amd.inference.online_chainer import OnlineChainer
    from stemseg_amd.pipeline import run_sequence_sharded
    from tests.oracle_ops import OracleChainerOps
    try:
        model, _ = _model("davis", "R-101-FPN", 1234, {"seediness_head.conv_out.weight": 30.0}, 480, 854)
        F = 64
        base = _frames(4, 480, 864, 854, seed=5)
        # 64 distinct frames from 4 random ones (shifted / mixed), so that overlapping clips see consistent content
        frames = torch.stack([torch.roll(base[t % 4], shifts=(3 * (t // 4)), dims=-1) * (0.8 + 0.005 * t) for t in range(F)], 0)
        frames[..., 854:] = 0
        frames = frames.cuda().contiguous()
        clips, _ = get_subsequence_frames(F, 8, "davis", 4)
        assert len(clips) == 15 and clips[-1] == list(range(56, 64))
        thr = 0.25                                             # the presets' own thresholds (seediness head gain 30, as in bench.py)
        tg = TrackGenerator(model, "davis", seediness_thresh=thr, frame_overlap=4)
        heads = {}

        def embed(fr):
            out = model.embed_frames(frames[torch.as_tensor(fr, device=frames.device)].contiguous())
            heads[tuple(fr)] = tuple(o.clone() for o in out)
            return out
        t0 = time.time()
        (track, counts, life), _, _, _, meta = run_sequence_sharded(F, embed, tg.chainer, "davis", frame_overlap=4, seediness_thresh=thr)
        torch.cuda.synchronize()
        t_sharded = time.time() - t0
        # oracle chain on the same head outputs
        from stemseg_amd.inference.main import fg_masks_from_seediness
        entries = [(list(fr),) + heads[tuple(fr)] for fr in clips]
        fg = fg_masks_from_seediness(entries, thr)
        dicts = [dict(frames=list(fr), embeddings=heads[tuple(fr)][0].cpu(), bandwidths=heads[tuple(fr)][1].cpu(),
                      seediness=heads[tuple(fr)][2].cpu()) for fr in clips]
        c = config.cfg.CLUSTERING
        ref_chain = OnlineChainer(SequentialClustering(c.PRIMARY_PROB_THRESHOLD, c.SECONDARY_PROB_THRESHOLD, c.MIN_SEEDINESS_PROB, 2,
                                                       [0.3, 0.3], "cpu"), 1.0, ops=OracleChainerOps())
        t0 = time.time()
        (rtrack, rcounts, rlife), _, _, _, rmeta = ref_chain.process(fg.cpu(), dicts)
        print("[fullsize] config3: sharded driver %.2f s for 15 clips (eager, incl. stitching); oracle chain %.1f s; %d fg points; "
              "track ids up to %d; K per clip %s" % (t_sharded, time.time() - t0, sum(counts.values()), max(list(counts) + [0]), [len(m["instance_labels"]) for m in meta]))
        assert len(track) == F
        for t in range(F):
            assert torch.equal(track[t].cpu(), rtrack[t]), "frame %d: track labels differ from the oracle chain" % t
        assert dict(counts) == dict(rcounts) and dict(life) == dict(rlife)
        assert [m["instance_labels"] for m in meta] == [m["instance_labels"] for m in rmeta]
        assert sum(counts.values()) > 500_000 and max(len(m["instance_labels"]) for m in meta) >= 5
        # (ii) the reference-shaped driver with the feature cache: same clips, encoder run per NEW frame batch
        embeddings, fg2, _ = tg.do_inference(frames)
        assert [list(e.subseq_frames) for e in embeddings] == clips
        worst = max(float((e.embeddings - heads[tuple(fr)][0]).abs().max()) for e, fr in zip(embeddings, clips))
        worst_s = max(float((e.seediness - heads[tuple(fr)][2]).abs().max()) for e, fr in zip(embeddings, clips))
        print("[fullsize] config3: cached-encoder driver vs per-clip encoder: emb %.2e, seediness %.2e, fg masks differ in %d voxels"
              % (worst, worst_s, int((fg2 != fg).sum())))
        assert worst <= 1e-3 and worst_s <= 1e-3
        (track2, counts2, _), _, _, _, meta2 = tg.do_clustering(embeddings, fg2)
        dicts2 = [dict(frames=list(e.subseq_frames), embeddings=e.embeddings.cpu(), bandwidths=e.bandwidths.cpu(), seediness=e.seediness.cpu())
                  for e in model(frames, clips)["embeddings"]]
        (rtrack2, rcounts2, _), _, _, _, rmeta2 = ref_chain.process(fg2.cpu(), dicts2)
        assert all(torch.equal(a.cpu(), b) for a, b in zip(track2, rtrack2)) and dict(counts2) == dict(rcounts2)
        assert [m["instance_labels"] for m in meta2] == [m["instance_labels"] for m in rmeta2]
    finally:
        config.load_preset("defaults")
