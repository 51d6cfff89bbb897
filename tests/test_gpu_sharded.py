"""GPU tests of the clip-parallel sequence path (pipeline.run_sequence_sharded): the four stitching entry
points against their numpy twins, the reference-generated chainer goldens through the HIP path at world 1, and N "virtual
ranks" on the one device (threads + an emulated all-gather: every rank's own-clip clustering with label_start = 1, the code
planes, the pair tables and the LUT gather all run the real kernels with the owner / plane arithmetic of a world-N job)."""
import zlib

import numpy as np
import pytest
import torch

from tests import synth
from tests.test_distributed_cpu import _case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from stemseg_amd import hip as h
    h.require_gpu()
    return h


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_stitching_kernels_vs_numpy_twins(hip):
    """fg_compact / labels_to_codes / pair_tables / codes_to_labels == tests/oracle_ops.py on ragged random planes."""
    from stemseg_amd.inference.online_chainer import HipChainerOps
    from tests.oracle_ops import OracleChainerOps
    ops, ref = HipChainerOps(), OracleChainerOps()
    rs = np.random.RandomState(3)
    for (F, h, w, K) in ((5, 7, 9, 3), (8, 120, 216, 20), (3, 33, 65, 64), (2, 1, 1, 1)):
        B = K + 2
        fg = (rs.uniform(size=(F, h, w)) < 0.4).astype(np.uint8)
        vox, offs = ops.compact(dev(fg))
        rvox, roffs = ref.compact(torch.from_numpy(fg))
        n = int(roffs[-1])
        assert np.array_equal(offs.cpu().numpy(), roffs.numpy()) and np.array_equal(vox.cpu().numpy()[:n], rvox.numpy()[:n])
        # a "clip" = all F frames: labels for the n points, label_start 7
        labels = np.where(rs.uniform(size=n) < 0.2, -1, 7 + rs.randint(0, K, n)).astype(np.int64)
        pts = dict(vox=vox, offs=offs, T=F)
        codes = torch.zeros(2, F * h * w, dtype=torch.uint8, device="cuda")
        ops.codes_from_labels(pts, dev(np.concatenate([labels, np.full(F * h * w - n, 5, np.int64)])), 7, codes[0])
        rcodes = torch.zeros(2, F * h * w, dtype=torch.uint8)
        ref.codes_from_labels(dict(vox=rvox, offs=roffs, T=F), torch.from_numpy(labels), 7, rcodes[0])
        assert np.array_equal(codes[0].cpu().numpy(), rcodes[0].numpy())
        # a second labelling of the same foreground for the pair tables
        labels2 = np.where(rs.uniform(size=n) < 0.1, -1, 1 + rs.randint(0, K, n)).astype(np.int64)
        ops.codes_from_labels(pts, dev(labels2), 1, codes[1])
        ref.codes_from_labels(dict(vox=rvox, offs=roffs, T=F), torch.from_numpy(labels2), 1, rcodes[1])
        planes, rplanes = codes.view(2 * F, h * w), rcodes.view(2 * F, h * w)
        pa = [-1 if t % 3 == 0 else t for t in range(F)]
        pb = [F + t for t in range(F)]
        tabs = ops.pair_tables(planes, pa, pb, B)
        (tabs_h,) = ops.read_back(tabs)
        assert np.array_equal(tabs_h, ref.pair_tables(rplanes, pa, pb, B).numpy())
        assert int(tabs_h.sum()) == n
        items, luts, cur = [], [], 0
        o = roffs.numpy()
        for t in range(F):
            items.append((int(o[t]), int(o[t + 1] - o[t]), t * h * w, F + t, cur))
            cur += int(o[t + 1] - o[t])
            luts.append(rs.randint(-1, 4000, B).astype(np.int64))
        mx = max(i[1] for i in items)
        out = ops.labels_from_codes(planes, vox, np.asarray(items, np.int64), np.stack(luts), mx, cur)
        rout = ref.labels_from_codes(rplanes, rvox, np.asarray(items, np.int64), np.stack(luts), mx, cur)
        assert np.array_equal(out.cpu().numpy(), rout.numpy())


def _run_gpu(tag, comm=None, fn_name="run_sequence_sharded"):
    from stemseg_amd import config, pipeline
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    emb, bw, sd, fg, clips, overlap, exp = _case(tag)
    config.load_preset("davis")
    emb_d, bw_d, sd_d, fg_d = dev(emb), dev(bw), dev(sd), dev(fg)

    def embed(frames):
        idx = torch.as_tensor(frames, device="cuda")
        return emb_d[:, idx].contiguous(), bw_d[:, idx].contiguous(), sd_d[:, idx].contiguous()
    chainer = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cuda:0"), 1.0)
    kw = {"comm": comm} if comm is not None else {}
    (track, counts, life), mask_idxes, clip_labels, _, meta = getattr(pipeline, fn_name)(
        fg.shape[0], embed, chainer, "davis", frame_overlap=overlap, fg_mask_fn=lambda entries, thr: fg_d, **kw)
    ok = len(track) == len(exp["track"]) and all(np.array_equal(l.cpu().numpy(), e) for l, e in zip(track, exp["track"]))
    ok = ok and sorted(counts.items()) == exp["counts"] and sorted(life.items()) == exp["life"]
    ok = ok and [m["instance_labels"] for m in meta] == exp["instance_labels"]
    for t in range(fg.shape[0]):
        ys, xs = np.nonzero(fg[t])
        ok = ok and np.array_equal(mask_idxes[t][0].cpu().numpy(), ys) and np.array_equal(mask_idxes[t][1].cpu().numpy(), xs)
    if "clip_labels" in exp:
        ok = ok and all(np.array_equal(torch.cat(clip_labels[i]).cpu().numpy(), exp["clip_labels"][i]) for i in range(len(clips)))
    crc = zlib.crc32(torch.cat([t.cpu() for t in track]).numpy().tobytes())
    centers = [m["instance_centers"] for m in meta]
    return bool(ok), crc, centers


@pytest.mark.parametrize("tag", ["seq20_ov4", "seq14_ov6", "seq8_single", "long"])
def test_clip_parallel_chain_on_gpu_vs_golden(hip, tag):
    ok, crc, centers = _run_gpu(tag)
    assert ok


@pytest.mark.parametrize("world,tag", [(2, "seq20_ov4"), (3, "seq14_ov6"), (8, "long"), (8, "seq20_ov4")])
def test_virtual_ranks_on_one_gpu_identical_tracks(hip, world, tag):
    """Every virtual rank ends with the reference's tracks and the same checksum as the one-rank run."""
    from tests.virtual_ranks import run_virtual_ranks
    ok1, crc1, centers1 = _run_gpu(tag)
    res = run_virtual_ranks(world, lambda comm: _run_gpu(tag, comm=comm))
    assert ok1 and all(r[0] for r in res)
    assert {r[1] for r in res} == {crc1}
    assert all(r[2] == centers1 for r in res)


@pytest.mark.parametrize("flow", ["ytvis", "kitti"])
def test_sharded_semseg_presets_match_the_single_process_chain(hip, flow):
    """the clip-parallel sequence path for the presets whose foreground comes from the semseg head -- YouTube-VIS
    (41+1 class logits, --resize_embeddings: everything x4, clustering at full resolution) and KITTI-MOTS (3+1 classes) -- on the
    reduced-size flows of tests/golden/model_{ytvis,kitti}.npz.  Fed with the head outputs of ``InferenceModel.forward`` the
    sharded driver must reproduce ``TrackGenerator.do_inference + do_clustering`` BIT FOR BIT at world 1 and on 2 / 3 virtual
    ranks -- and likewise when fed by ``ClipPipeline.embed_many`` (its own encoder passes, of other frame counts: the encoder plans
    every launch for a fixed frame count, so the maps do not depend on them)."""
    from stemseg_amd import config, pipeline
    from stemseg_amd.inference.main import TrackGenerator
    from stemseg_amd.modeling.inference_model import InferenceModel, preprocess_frames
    from tests.virtual_ranks import run_virtual_ranks
    yt = flow == "ytvis"
    config.load_preset("ytvis" if yt else "kittimots")
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = (96, 128) if yt else (96, 320)
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    config.cfg.CLUSTERING.MIN_SEEDINESS_PROB = 0.8
    try:
        r = 4.0 if yt else 1.0
        model = InferenceModel(resize_scale=r)
        msd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 81 if yt else 91))).reshape(v.shape) for k, v in msd.items()}
        new["embedding_head.conv_seediness.weight"] = new["embedding_head.conv_seediness.weight"] * 6.0
        model._model.load_state_dict(new)
        model = model.cuda()
        frames = synth.synth_frames(12, 96, 128, seed=81) if yt else synth.synth_frames(14, 60, 190, seed=91)
        name = "ytvis" if yt else "kittimots"
        tg = TrackGenerator(model, name, resize_scale=r, frame_overlap=4)
        n = len(frames)
        out = model([f for f in frames], pipeline.get_subsequence_frames(n, 8, name, 4)[0])
        fg1 = torch.stack([hip.fg_mask(p.contiguous(), 1.0, 0.5) for p in out["fg_masks"]], 0)
        dicts = [{"frames": e.subseq_frames, "embeddings": e.embeddings, "bandwidths": e.bandwidths, "seediness": e.seediness} for e in out["embeddings"]]
        (t0, c0, l0), m0, _, _, meta0 = tg.chainer.process(fg1, dicts)
        assert int(fg1.sum()) > 1000 and len(c0) >= 2
        by = {tuple(e.subseq_frames): (e.embeddings, e.bandwidths, e.seediness, fl) for e, fl in zip(out["embeddings"], out["clip_fg_logits"])}
        Cfg = out["clip_fg_logits"][0].shape[0]

        def run(comm=None):
            (t1, c1, l1), m1, _, _, meta1 = pipeline.run_sequence_sharded(
                n, lambda fr: by[tuple(sorted(set(fr)))], tg.chainer, name, frame_overlap=4, fg_logit_channels=Cfg,
                **({"comm": comm} if comm is not None else {}))
            ok = all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(t0, t1)) and dict(c0) == dict(c1) and l0 == l1
            ok = ok and all(torch.equal(a[0].cpu(), b[0].cpu()) and torch.equal(a[1].cpu(), b[1].cpu()) for a, b in zip(m0, m1))
            return ok and [m["instance_labels"] for m in meta0] == [m["instance_labels"] for m in meta1]
        assert run()
        for world in (2, 3):
            assert all(run_virtual_ranks(world, lambda comm: run(comm)))
        # the full device path: the rank's own encoder passes + third decoder (embed_many with the foreground logits)
        pipe = pipeline.ClipPipeline(model)
        x, _ = preprocess_frames(np.stack([np.asarray(f) for f in frames], 0), "cuda")
        eh = model._model.embedding_head
        (t2, c2, _), _, _, _, _ = pipeline.run_sequence_sharded(
            n, None, tg.chainer, name, frame_overlap=4, fg_logit_channels=Cfg, channel_split=(eh.embedding_size, eh.variance_channels),
            embed_many_fn=lambda my: pipe.embed_many(x, my, batch=2, lanes=1, use_graph=False, with_fg_logits=True))
        assert len(t0) == len(t2) and all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(t0, t2)) and dict(c0) == dict(c2), \
            "%s: the embed_many-fed sequence differs from the reference-API flow" % flow
    finally:
        config.load_preset("defaults")
