"""Results do not depend on the batch / launch shape.

The K-partition of a split-K convolution is a summation order.  Through round 4 the launcher chose it from the size of the launch, so
the last bits of a frame's encoder maps depended on how many frames shared the encoder pass -- one clip, four clips of a bench step,
the union of eight overlapping windows, or whatever block of clips a rank of a world-N job owns -- and the clusterer behind them is a
chain of hard thresholds (/root/reference/stemseg/inference/clusterers.py:106-146): a world-3 job differed from world-1 in 2 of
1 656 561 labels.  Now every encoder launch decides tile shape and split-K factor on its PLANNING shape (per-frame layer shape x
``plan_frames``; conv_igemm.hip, PlanCtx), and these tests hold the product to the reference's contract
(/root/reference/stemseg/inference/online_chainer.py:193-236: one answer per sequence): ``torch.equal`` everywhere, one label checksum
at world 1 / 2 / 3 / 8.  Also here: the T = 16 flow the reference's CLI loads by default (config/davis_2.yaml:5,
inference/main.py:188-195), and the overflow re-run inside a graph lane."""
import zlib

import numpy as np
import pytest
import torch

from oracle import pipeline as opipe
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from stemseg_amd import hip as h
    h.require_gpu()
    return h


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _rand(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def _davis_model(seed, backbone="R-50-FPN", preset="davis", seed_gain=30.0):
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    config.load_preset(preset)
    config.cfg.MODEL.BACKBONE.TYPE = backbone
    model = InferenceModel()
    names = [(k, v.shape) for k, v in model._model.state_dict().items()]
    sd = synth.synth_state_dict(names, seed)
    sd["seediness_head.conv_out.weight"] = sd["seediness_head.conv_out.weight"] * seed_gain
    model._model.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(model._model.state_dict()[k].shape) for k, v in sd.items()})
    return model, sd


# ------------------------------------------------------------------------------------------------ the convolution itself
@pytest.mark.parametrize("precision", ["f16x3", "bf16x6", "f32"])
@pytest.mark.parametrize("kind", ["k2", "k1_expand", "k1_reduce_decode"])
def test_conv_output_of_a_frame_does_not_depend_on_the_frames_in_the_launch(hip, kind, precision):
    """One layer, the same 40 frames, launched as 40 / 32 / 12 / 8 / 5 / 1 frames at a time with the SAME planning frame count: every
    frame's output is bit-identical whatever shared its launch (shapes where the un-planned launcher changes its split-K factor
    between these sizes: layer-3 / layer-4 maps).  And the plan is what does it: the same launches WITHOUT a plan differ somewhere."""
    Fr = 40
    if kind == "k2":
        Cin, Cout, h, w = 256, 256, 15, 27
    elif kind == "k1_expand":
        Cin, Cout, h, w = 256, 1024, 8, 14
    else:
        Cin, Cout, h, w = 1024, 256, 8, 14
    rs = 7 + len(kind)
    x = _rand((Cin, Fr, h, w), rs)
    taps = 9 if kind == "k2" else 1
    wt = _rand((Cout, Cin, 1, 3, 3) if kind == "k2" else (Cout, Cin, 1, 1, 1), rs + 1, 1.0 / np.sqrt(Cin * taps))
    b = dev(_rand((Cout,), rs + 2))
    pw = hip.pack_conv_weight_any(dev(wt), precision)
    scratch = torch.empty(64 << 20, device="cuda")
    plan_scratch = 32 << 20
    r_full = _rand((Cout, Fr, h, w), rs + 3) if kind == "k1_expand" else None

    def run(frames_per_launch, plan):
        """-> [Cout, Fr, h, w] assembled from launches of ``frames_per_launch`` frames"""
        out = torch.full((Cout, Fr, h, w), float("nan"), device="cuda")
        for f0 in range(0, Fr, frames_per_launch):
            n = min(frames_per_launch, Fr - f0)
            xs = x[:, f0:f0 + n]
            epi = dict(precision=precision, relu=1)
            if plan:
                epi["plan"] = (n, 32, plan_scratch)
            if kind == "k2":
                pitch = (w + 2 + 3) // 4 * 4
                buf = torch.zeros(Cin, n, h + 2, pitch, device="cuda")
                buf[:, :, 1:h + 1, 1:w + 1] = dev(xs)
                vin = hip.Volume(buf.data_ptr(), n * (h + 2) * pitch, (h + 2) * pitch, pitch, Cin, n, h + 2, w + 2, buf.numel())
                o = torch.full((Cout, n, h, w), float("nan"), device="cuda")
                hip.conv3d(vin, pw, b, hip.dense_volume(o), (1, 3, 3), 0, scratch, epi)
            else:
                xd = dev(xs.reshape(Cin, -1))
                o = torch.full((Cout, n, h, w), float("nan"), device="cuda")
                if kind == "k1_reduce_decode":                 # the encoder's conv1: flat [C][V] input, decoded into a (t, y, x) layout
                    epi["decode"] = (h, w)
                    hip.conv3d(hip.flat_volume(xd), pw, b, hip.dense_volume(o), 1, 0, scratch, epi)
                else:
                    r = dev(r_full[:, f0:f0 + n].reshape(Cout, -1))
                    epi.update(residual=r, res_strides=(n * h * w, 0, 0))
                    hip.conv3d(hip.flat_volume(xd), pw, b, hip.flat_volume(o.view(Cout, -1)), 1, 0, scratch, epi)
            out[:, f0:f0 + n] = o
        torch.cuda.synchronize()
        return out

    ref = run(32, True)
    assert bool(torch.isfinite(ref).all())
    for fpl in (40, 12, 8, 5, 1):
        got = run(fpl, True)
        assert torch.equal(got, ref), "%s %s: %d frames per launch differs from 32 per launch under the same plan" % (kind, precision, fpl)
    # sanity of the test itself: un-planned launches of these sizes do NOT all agree (else the shapes above prove nothing)
    free = [run(fpl, False) for fpl in (40, 8, 1)]
    assert any(not torch.equal(a, free[0]) for a in free[1:]), "un-planned launches agree on this shape: pick one where split-K varies"
    # and a planned launch whose scratch cannot hold frames / plan_frames times the plan's slabs fails loudly instead of re-partitioning
    small = torch.empty(1024, device="cuda")
    xd = dev(x[:, :8].reshape(Cin, -1)) if kind != "k2" else None
    if kind == "k1_expand":
        with pytest.raises(RuntimeError, match="scratch too small"):
            hip.conv3d(hip.flat_volume(xd), pw, b, hip.flat_volume(torch.empty(Cout, 8 * h * w, device="cuda")), 1, 0, small,
                       dict(precision=precision, plan=(8, 1, plan_scratch)))


# ------------------------------------------------------------------------------------------------ encoder + decoders
def test_embeddings_are_bit_identical_for_every_batch_shape_and_entry_point(hip):
    """A 36-frame sequence cut into 8 windows (overlap 4), R-50, 96 x 160: the [E+Ev+1, T, h4, w4] block of every clip through
    ``embed_many`` with 1 / 2 / 3 / 5 / 8 windows per encoder pass (graph replays and eager), through ``embed_frames`` clip by
    clip, through ``step_batch`` of stacked clips, and through ``InferenceModel.forward``'s feature cache (encoder passes of 8
    then 4 new frames) -- all ``torch.equal``.  ``plan_frames`` is a throughput knob, not a result knob WITHIN one value; across
    values the maps agree to fp32 rounding."""
    from stemseg_amd import config
    from stemseg_amd.pipeline import ClipPipeline, get_subsequence_frames
    model, _ = _davis_model(3)
    try:
        pipe = ClipPipeline(model)
        frames = (torch.from_numpy(synth.synth_frames(36, 96, 160, seed=3).astype(np.float32)).permute(0, 3, 1, 2) - 110.0).cuda().contiguous()
        clips, _ = get_subsequence_frames(36, 8, "davis", 4)
        assert len(clips) == 8
        ref = [torch.cat(pipe.embed(frames[c].contiguous()), 0).clone() for c in clips]
        for batch, graph in ((1, False), (2, True), (3, False), (5, True), (8, True), (8, False)):
            got = pipe.embed_many(frames, clips, batch=batch, lanes=2, use_graph=graph)
            torch.cuda.synchronize()
            for i in range(len(clips)):
                assert torch.equal(got[i], ref[i]), "embed_many(batch=%d, graph=%s): clip %d differs from the clip embedded on its own" % (batch, graph, i)
        # a rank of a world-3 job owns 3 + 3 + 2 windows, of a world-8 job one each: same blocks
        for block in ([0, 1, 2], [3, 4, 5], [6, 7], [4]):
            got = pipe.embed_many(frames, [clips[i] for i in block], batch=8, lanes=2)
            torch.cuda.synchronize()
            assert all(torch.equal(g, ref[i]) for g, i in zip(got, block))
        # stacked independent clips (the bench's step) and the reference-API sequence driver
        outs = pipe.step_batch(torch.cat([frames[clips[0]], frames[clips[5]], frames[clips[7]]], 0).contiguous(), 3)
        for o, i in zip(outs, (0, 5, 7)):
            assert torch.equal(torch.cat([o["emb"], o["bw"], o["seed"]], 0), ref[i])
        res = model(frames, clips)
        for i, e in enumerate(res["embeddings"]):
            assert torch.equal(torch.cat([e.embeddings, e.bandwidths, e.seediness], 0), ref[i]), "InferenceModel.forward, clip %d" % i
        # another planning value: a different (fixed) K-partition -> fp32-rounding-level differences, nothing more
        bb = model._model.backbone
        bb.plan_frames = 8
        other = torch.cat(pipe.embed(frames[clips[2]].contiguous()), 0)
        many = pipe.embed_many(frames, clips, batch=8, lanes=1, use_graph=False)
        torch.cuda.synchronize()
        assert torch.equal(many[2], other)
        sc = np.maximum(1.0, np.abs(ref[2].cpu().numpy()))
        assert float(np.abs((other - ref[2]).cpu().numpy() / sc).max()) <= 1e-4        # (the x30 gain on the seediness logits of this fixture amplifies the trunk's last bits)
        bb.plan_frames = 32
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ the decoders' clip batch
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("which", ["embedding", "seediness", "semseg_wide", "embedding_no_norm"])
def test_decoder_clip_batch_is_bit_identical_to_single_clip_calls(hip, which, precision):
    """The decoders run all clips of a step in ONE launch per stage (StemsegDecoderDesc.n_clips: the clip is a grid dimension of every
    kernel -- conv + GroupNorm partial sums + split-K reduce, finalize, apply + pool, trilinear up-sampling, 1x1x1 fuse, heads).
    Every launch decision is taken on one clip's shape, so clip c of a batch of 3 must equal the single-clip call on clip c's
    inputs BIT FOR BIT -- for the fused 6-channel heads, the seediness decoder, the 41-class head that goes through the MFMA conv
    and a decoder without normalisation layers (max pooling)."""
    from stemseg_amd.modeling.embedding_decoder import SqueezingExpandDecoder as Emb
    from stemseg_amd.modeling.seediness_decoder import SqueezingExpandDecoder as Seed
    from stemseg_amd.modeling.semseg_decoder import SqueezeExpandDecoder as Sem
    T, h32, w32, N = 8, 3, 5, 3
    gn = lambda c: torch.nn.GroupNorm(32, c)     # noqa: E731
    if which == "embedding":
        m = Emb(256, [256, 256, 128, 128], 4, True, False, "xyff", NormType=gn, num_frames=T)
    elif which == "embedding_no_norm":
        m = Emb(256, [256, 256, 128, 128], 5, True, True, "xytff", PoolType=torch.nn.MaxPool3d, NormType=lambda c: torch.nn.Identity(), num_frames=T)
    elif which == "seediness":
        m = Seed(256, [256, 256, 128, 128], NormType=gn, num_frames=T)
    else:
        m = Sem(256, 40, [128, 128, 64, 64], (4, 8, 16, 32), foreground_channel=True, NormType=gn, num_frames=T)
    sd = synth.synth_state_dict([(k, v.shape) for k, v in m.state_dict().items()], 19, prefix="x.")
    if which == "embedding_no_norm":
        sd = {k: (np.asarray(v) * (0.35 if np.asarray(v).ndim == 5 and np.asarray(v).shape[-1] == 3 else 1.0)).astype(np.float32) for k, v in sd.items()}
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(m.state_dict()[k].shape) for k, v in sd.items()})
    m = m.cuda().eval()
    m.precision = precision
    Cn = 256
    levels = [hip.alloc_padded_batch(N, Cn, T, h32 * s, w32 * s, "cuda") for s in (1, 2, 4, 8)]      # 32x, 16x, 8x, 4x
    for c in range(N):
        feats = synth.synth_features(T, h32, w32, seed=60 + c)
        for (bufs, g, _), f, s in zip(levels, feats, (1, 2, 4, 8)):
            hip.copy_to_volume(dev(f), 0, hip.padded_interior_view(bufs[c], g, Cn, T, h32 * s, w32 * s))
    shape = (T, h32 * 8, w32 * 8)
    singles = [m.forward_single(([lv[0][c] for lv in levels], shape), 2).clone() for c in range(N)]
    batch = m.forward_single(([lv[0][0] for lv in levels], shape), 2, clip_batch=(N, [lv[2] for lv in levels]))
    torch.cuda.synchronize()
    assert batch.shape[0] == N and batch.shape[1:] == singles[0].shape
    assert not torch.equal(singles[0], singles[1])
    for c in range(N):
        assert torch.equal(batch[c], singles[c]), "%s %s: clip %d of the batch differs from its single-clip call" % (which, precision, c)
    bad, where = m.check_workspaces()
    assert bad == 0, "guard words clobbered: %s" % (where,)


# ------------------------------------------------------------------------------------------------ the sequence, world 1 / 2 / 3 / 8
def test_sequence_label_checksum_is_the_same_at_world_1_2_3_8(hip):
    """bench.py --sequence at reduced size (52 frames -> 12 clips, R-50, 96 x 160) with the REAL embedding path of every rank: rank r
    of a world-N job embeds ITS block of clips with ``embed_many`` (its own encoder passes, of its own shapes) and the chain runs on N
    virtual ranks.  One label checksum for N = 1, 2, 3, 8 -- and it is the checksum of the reference-API flow
    (InferenceModel.forward + OnlineChainer.process) on the same frames."""
    from stemseg_amd import config, pipeline
    from stemseg_amd.inference.main import TrackGenerator
    from stemseg_amd.pipeline import ClipPipeline, get_subsequence_frames, shard_clips
    from tests.virtual_ranks import run_virtual_ranks
    model, _ = _davis_model(11, seed_gain=25.0)
    try:
        pipe = ClipPipeline(model, seediness_thresh=0.4)
        model.overlap_decoders = False
        n = 52
        frames = (torch.from_numpy(synth.synth_frames(n, 96, 160, seed=11).astype(np.float32)).permute(0, 3, 1, 2) - 110.0).cuda().contiguous()
        clips, _ = get_subsequence_frames(n, 8, "davis", 4)
        eh = model._model.embedding_head
        split = (eh.embedding_size, eh.variance_channels)
        chainer = pipe.tg.chainer

        def crc_of(track):
            return zlib.crc32(torch.cat([t.cpu() for t in track]).numpy().tobytes())

        # the reference-API flow
        tg = TrackGenerator(model, "davis", seediness_thresh=0.4, frame_overlap=4)
        out = model(frames, clips)
        fg = pipeline.fg_masks_from_seediness(out["embeddings"], 0.4)
        dicts = [{"frames": e.subseq_frames, "embeddings": e.embeddings, "bandwidths": e.bandwidths, "seediness": e.seediness} for e in out["embeddings"]]
        (t_ref, c_ref, _), _, _, _, meta_ref = tg.chainer.process(fg, dicts)
        want = crc_of(t_ref)
        assert int(fg.sum()) > 2000 and len(c_ref) >= 2, "the fixture should hold several instances (fg %d, ids %s)" % (int(fg.sum()), sorted(c_ref))
        crcs = {}
        for world in (1, 2, 3, 8):
            blocks = {}
            for r in range(world):                         # every rank's own encoder passes (sequentially: one device, one model)
                mine = shard_clips(len(clips), r, world)
                blocks[r] = [b.clone() for b in pipe.embed_many(frames, [clips[i] for i in mine], batch=8, lanes=2)] if mine else []
            torch.cuda.synchronize()

            def rank_fn(comm, world=world, blocks=blocks):
                (track, counts, life), _, _, _, meta = pipeline.run_sequence_sharded(
                    n, None, chainer, "davis", frame_overlap=4, seediness_thresh=0.4, embed_many_fn=lambda my: blocks[comm.rank if comm is not None else 0],
                    channel_split=split, **({"comm": comm} if comm is not None else {}))
                return crc_of(track), sorted(counts.items()), [m["instance_labels"] for m in meta]
            res = [rank_fn(None)] if world == 1 else run_virtual_ranks(world, rank_fn)
            assert len({r[0] for r in res}) == 1 and all(r[1:] == res[0][1:] for r in res), "ranks of a world-%d job disagree" % world
            crcs[world] = res[0][0]
            assert res[0][1] == sorted(c_ref.items()) and res[0][2] == [m["instance_labels"] for m in meta_ref]
        print("[invariance] label checksum by world: %s (reference-API flow: %d)" % (crcs, want))
        assert set(crcs.values()) == {want}
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ NUM_FRAMES = 16
def test_davis_2_num_frames_16_flow_vs_oracle_and_under_graph_capture(hip):
    """config/davis_2.yaml (NUM_FRAMES 16: what the reference's CLI loads for --dataset davis, inference/main.py:188-195): a 24-frame
    sequence -> windows of 16 at overlap 6 through InferenceModel.forward + TrackGenerator, the first clip against the CPU oracle
    (decoders with three temporal pooling stages and x2 temporal up-sampling on every level: common.py:8-35), and a captured
    ClipPipeline step on a 16-frame clip == the eager step."""
    from stemseg_amd import config
    from stemseg_amd.inference.main import TrackGenerator
    from stemseg_amd.pipeline import ClipPipeline
    model, sd = _davis_model(17, preset="davis_2")
    try:
        assert config.cfg.INPUT.NUM_FRAMES == 16 and model._model.embedding_head.num_frames == 16
        model = model.cuda()
        frames_u8 = synth.synth_frames(24, 96, 128, seed=17)
        frames = torch.from_numpy(frames_u8.astype(np.float32)).permute(0, 3, 1, 2) - torch.tensor(config.cfg.INPUT.IMAGE_MEAN)[None, :, None, None]
        tg = TrackGenerator(model, "davis", seediness_thresh=0.25, frame_overlap=6)
        embeddings, fg, _ = tg.do_inference(frames.cuda())
        assert [len(e.subseq_frames) for e in embeddings] == [16, 16] and embeddings[1].subseq_frames[0] == 8
        ref = opipe.embed_clip(frames[:16], sd, "R-50-FPN", "xyff", 4, True)
        e0 = embeddings[0]
        assert float((e0.embeddings.cpu() - ref[0]).abs().max()) <= 1e-3
        assert float((e0.bandwidths.cpu() / ref[1] - 1).abs().max()) <= 1e-3
        assert float((e0.seediness.cpu() - ref[2]).abs().max()) <= 1e-3
        (track, counts, life), _, _, _, meta = tg.do_clustering(embeddings, fg)
        assert len(track) == 24 and sum(int(t.numel()) for t in track) == int(fg.sum())
        # the bench's unit of work at T = 16: eager step == captured step, labels through the oracle's clusterer on the oracle's maps
        pipe = ClipPipeline(model)
        clip = frames[:16].cuda().contiguous()
        eager = pipe.step(clip)
        keep = {k: eager[k].clone() for k in ("emb", "bw", "seed", "labels", "fg")}
        g = pipe.capture(clip)
        for _ in range(2):
            o = g.run(clip)
            torch.cuda.synchronize()
            assert all(torch.equal(o[k], keep[k]) for k in keep)
        assert torch.equal(torch.cat([keep["emb"], keep["bw"], keep["seed"]], 0), torch.cat([e0.embeddings, e0.bandwidths, e0.seediness], 0))
        full = opipe.embed_and_cluster_clip(frames[:16], sd, "R-50-FPN", "xyff", 4, True, free_dim_stds=[0.3, 0.3], return_probs=True)
        o2 = pipe.cluster(full["emb"].cuda().contiguous(), full["bw"].cuda().contiguous(), full["seed"].cuda().contiguous())
        nn = int(o2["frame_offsets"].cpu()[-1])
        assert nn == full["labels"].shape[0] and hip.read_cluster_meta(o2["meta"]).K == len(full["meta"]["instance_labels"])
        bad = np.flatnonzero(o2["labels"][:nn].cpu().numpy() != full["labels"])
        if bad.size:
            P = np.stack(full["meta"]["instance_probs"])
            near = (np.abs(P - 0.5) < 2e-6).any(0) | (np.abs(P - 0.3) < 2e-6).any(0)
            assert near[bad].all()
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ overflow inside a graph lane
def test_graph_lane_reruns_an_overflowing_batch_in_bf16x6_instead_of_raising(hip):
    """A captured f16x3 lane fed a batch whose activations leave the fp16 range: ``GraphedStep.collect`` re-runs the lane's batch
    eagerly in bf16x6 on the lane's own stream and workspaces, returns finite maps + labels identical to a plain bf16x6 step, leaves
    the model in f16x3, and the NEXT replay of the same lane (a normal batch) is bit-identical to before the incident."""
    from stemseg_amd import config
    from stemseg_amd.pipeline import ClipPipeline
    model, _ = _davis_model(29, seed_gain=40.0)
    try:
        model.set_precision("f16x3")
        pipe = ClipPipeline(model, seediness_thresh=0.5)
        a, b = (dev(synth.synth_frames(8, 96, 160, seed=s).astype(np.float32).transpose(0, 3, 1, 2) - 110.0) for s in (4, 5))
        good = torch.cat([a, b], 0).contiguous()
        bad = torch.cat([a, b * 3.0e4], 0).contiguous()                # clip 1 of the batch overflows
        lane = pipe.capture(good, n_clips=2, lane=1)
        lane.run_async(good)
        outs, metas = lane.collect()
        before = [{k: o[k].clone() for k in ("emb", "labels")} for o in outs]
        assert all(m.K >= 0 for m in metas)
        lane.run_async(bad)
        with pytest.raises(hip.NonFiniteError):
            lane.collect(fallback_precision=None)
        lane.run_async(bad)
        outs2, metas2 = lane.collect()
        assert all(bool(torch.isfinite(o["emb"]).all()) and int(o["status"].sum()) == 0 for o in outs2)
        assert model.precisions() == {k: "f16x3" for k in model.precisions()} and model.lane == 0
        got = [{k: o[k].clone() for k in ("emb", "bw", "seed", "labels")} for o in outs2]
        model.set_precision("bf16x6")
        ref = pipe.step_batch(bad, 2)
        for g_, r_, m2 in zip(got, ref, metas2):
            assert all(torch.equal(g_[k], r_[k]) for k in g_) and hip.read_cluster_meta(r_["meta"], r_["status"]).K == m2.K
        model.set_precision("f16x3")
        lane.run_async(good)
        outs3, _ = lane.collect()
        assert all(torch.equal(o["emb"], p_["emb"]) and torch.equal(o["labels"], p_["labels"]) for o, p_ in zip(outs3, before))
    finally:
        config.load_preset("defaults")


# ------------------------------------------------------------------------------------------------ workspace canaries
@pytest.mark.parametrize("preset", ["davis", "ytvis"])
def test_no_kernel_writes_outside_its_workspace_slice(hip, preset):
    """SURVEY.md section 5 (race / sanitizer row): every slice of the encoder / decoder workspaces is followed by a guard block holding
    a canary.  After single steps, batched steps, graph replays on two lanes and windowed encoder passes -- ragged sizes included --
    not one guard word has changed; and the check does notice a stray write."""
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    config.load_preset(preset)
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        yt = preset == "ytvis"
        model = InferenceModel(resize_scale=4.0) if yt else InferenceModel()
        msd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 61))).reshape(v.shape) for k, v in msd.items()}
        model._model.load_state_dict(new)
        pipe = ClipPipeline(model)
        for (Hh, Ww) in ((96, 160), (64, 96), (128, 224)):
            frames = (torch.from_numpy(synth.synth_frames(20, Hh, Ww, seed=61).astype(np.float32)).permute(0, 3, 1, 2) - 110.0).cuda().contiguous()
            for prec in ("f16x3", "bf16x6", "f32"):
                model.set_precision(prec)
                pipe.step(frames[:8].contiguous())
                pipe.step_batch(frames[:16].contiguous(), 2)
            model.set_precision("f16x3")
            clips = [list(range(s, s + 8)) for s in (0, 4, 8, 12)]
            pipe.embed_many(frames, clips, batch=2, lanes=2, with_fg_logits=yt)
            pipe.embed_many(frames, clips, batch=3, lanes=1, use_graph=False, with_fg_logits=yt)
            g = pipe.capture(frames[:16].contiguous(), n_clips=2, lane=1)
            g.run_async(frames[4:20].contiguous())
            g.collect()
            torch.cuda.synchronize()
        bad, where = model.check_workspaces()
        assert bad == 0, "guard words clobbered: %s" % (where,)
        n_ws = len(model._model.backbone._ws) + sum(len(getattr(model._model, n)._workspaces) for n in ("embedding_head", "seediness_head", "semseg_head")
                                                    if getattr(model._model, n) is not None)
        assert n_ws >= 6
        # the check sees a stray write: poke the guard behind the FIRST slice of one encoder workspace (the S0 buffer; X1 follows it)
        bb = model._model.backbone
        key = next(iter(bb._ws))
        offs = (hip.C.c_int64 * 25)()
        hip.check(hip.lib().stemseg_hip_encoder_plan_offsets(hip.C.byref(bb._ws_desc[key]), offs))
        ws_f = bb._ws[key].view(torch.float32)
        guard = int(offs[1]) - 64
        saved = ws_f[guard + 5].clone()
        ws_f[guard + 5] = 1.0
        bad, where = model.check_workspaces()
        assert bad == 1 and where[0][0] == "backbone" and where[0][2] == guard + 5
        ws_f[guard + 5] = saved
        assert model.check_workspaces()[0] == 0
    finally:
        config.load_preset("defaults")
