"""Lane soak: three captured pipelines (own workspaces, own HIP streams) replay clip batches concurrently for >= 170 rounds per
convolution mode -- >= 500 clip results -- and EVERY result must be bit-identical to the first result of its input batch,
whichever lane produced it and whatever else was in flight (a 1 % flip rate must fail the GPU suite).

History: rounds 3 / 4 saw ~1 % (on some boxes 20-60 %) of the multi-lane steps differ from the lone
replay.  tools/soak_probe.py localised every one of them to the encoder's VALU stem kernel -- 5-13 wrong words in one 16-word run
of its output, one accumulator register, lanes 48..63 of a wave -- never to an MFMA convolution; with the stem on the matrix cores
the differences are gone (0 of 6 900 lane-rounds).  This test is the guard: it fails on such a box with the VALU stem of the
experiment build (-DSS_EXPERIMENTS + STEMSEG_STEM=valu; the product library does not contain that kernel)."""
import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu

LANES, ROUNDS, NC = 3, 60, 3          # 3 lanes x 60 rounds x 3 clips = 540 clip results per mode


@pytest.mark.parametrize("precision", ["f16x3", "bf16x6"])
def test_three_lanes_every_clip_result_is_bit_identical(precision):
    from stemseg_amd import config, hip
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    hip.require_gpu()
    config.load_preset("davis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel()
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 41))).reshape(v.shape) for k, v in sd.items()}
        new["seediness_head.conv_out.weight"] = new["seediness_head.conv_out.weight"] * 40.0
        model._model.load_state_dict(new)
        model.set_precision(precision)
        pipe = ClipPipeline(model, seediness_thresh=0.5)
        pipe.model.overlap_decoders = False
        H, W = 256, 448                                       # BASELINE configs[0]'s frame size: every conv class launches more than one round of workgroups
        batches = [torch.cat([torch.as_tensor(synth.synth_frames(8, H, W, seed=100 * b + c).astype(np.float32).transpose(0, 3, 1, 2) - 110.0).cuda()
                              for c in range(NC)], 0) for b in range(2)]
        pipe.step_batch(batches[0], NC)
        torch.cuda.synchronize()
        lanes = [pipe.capture(batches[0], n_clips=NC, lane=k) for k in range(LANES)]

        def signature(outs):
            """per clip: exact, order-independent int64 sums of the bit patterns of the embedding / bandwidth / seediness maps and of the labels"""
            rows = []
            for o in outs:
                n = o["frame_offsets"][-1]
                lab = torch.where(torch.arange(o["labels"].numel(), device=o["labels"].device) < n, o["labels"], torch.zeros_like(o["labels"]))
                rows.append(torch.stack([o["emb"].view(torch.int32).sum(dtype=torch.int64), o["bw"].view(torch.int32).sum(dtype=torch.int64),
                                         o["seed"].view(torch.int32).sum(dtype=torch.int64), lab.sum(dtype=torch.int64), n.to(torch.int64)]))
            return torch.stack(rows)
        ref = []
        for b in range(2):                                    # the lone replay of lane 0 defines the expected result of each batch
            sig = signature(lanes[0].run(batches[b])).cpu()
            torch.cuda.synchronize()
            ref.append(sig)
        assert not torch.equal(ref[0], ref[1])
        sigs = []
        for r in range(ROUNDS):
            which = [(r + k) % 2 for k in range(LANES)]
            for k, g in enumerate(lanes):
                g.run_async(batches[which[k]])
            for k, g in enumerate(lanes):
                with torch.cuda.stream(g.stream):
                    sigs.append((r, k, which[k], signature(g.out)))
        torch.cuda.synchronize()
        bad = [(r, k) for r, k, b, sgn in sigs if not torch.equal(sgn.cpu(), ref[b])]
        n = len(sigs) * NC
        print("[soak] %s: %d clip results over %d lanes x %d rounds, %d lane-rounds differ from the lone replay" % (precision, n, LANES, ROUNDS, len(bad)))
        assert n >= 500
        assert not bad, "%d of %d lane-rounds differ from the lone replay (first: round %d lane %d)" % (len(bad), len(sigs), bad[0][0], bad[0][1])
    finally:
        config.load_preset("defaults")


def test_three_lanes_ytvis_semseg_and_resize_path():
    """The same soak on the YouTube-VIS flow: in-head seediness, the 41+1-class semseg decoder, x4 trilinear
    resize of every head output and clustering at full resolution -- 3 lanes x 60 rounds x 3 clips = 540 clip results, each
    bit-identical to the lone replay of its batch."""
    from stemseg_amd import config, hip
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    hip.require_gpu()
    config.load_preset("ytvis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel(resize_scale=4.0)
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 43))).reshape(v.shape) for k, v in sd.items()}
        new["embedding_head.conv_seediness.weight"] = new["embedding_head.conv_seediness.weight"] * 6.0
        model._model.load_state_dict(new)
        pipe = ClipPipeline(model)
        pipe.model.overlap_decoders = False
        H, W = 128, 224
        batches = [torch.cat([torch.as_tensor(synth.synth_frames(8, H, W, seed=300 * b + c).astype(np.float32).transpose(0, 3, 1, 2) - 110.0).cuda()
                              for c in range(NC)], 0) for b in range(2)]
        pipe.step_batch(batches[0], NC)
        torch.cuda.synchronize()
        lanes = [pipe.capture(batches[0], n_clips=NC, lane=k) for k in range(LANES)]

        def signature(outs):
            rows = []
            for o in outs:
                n = o["frame_offsets"][-1]
                lab = torch.where(torch.arange(o["labels"].numel(), device=o["labels"].device) < n, o["labels"], torch.zeros_like(o["labels"]))
                rows.append(torch.stack([o["emb"].view(torch.int32).sum(dtype=torch.int64), o["bw"].view(torch.int32).sum(dtype=torch.int64),
                                         o["seed"].view(torch.int32).sum(dtype=torch.int64), o["semseg_logits"].view(torch.int32).sum(dtype=torch.int64),
                                         o["fg"].sum(dtype=torch.int64), lab.sum(dtype=torch.int64), n.to(torch.int64)]))
            return torch.stack(rows)
        ref = []
        for b in range(2):
            sig = signature(lanes[0].run(batches[b])).cpu()
            torch.cuda.synchronize()
            ref.append(sig)
        assert not torch.equal(ref[0], ref[1]) and int(ref[0][:, -1].min()) > 1000
        sigs = []
        for r in range(ROUNDS):
            which = [(r + k) % 2 for k in range(LANES)]
            for k, g in enumerate(lanes):
                g.run_async(batches[which[k]])
            for k, g in enumerate(lanes):
                with torch.cuda.stream(g.stream):
                    sigs.append((r, k, which[k], signature(g.out)))
        torch.cuda.synchronize()
        bad = [(r, k) for r, k, b, sgn in sigs if not torch.equal(sgn.cpu(), ref[b])]
        n = len(sigs) * NC
        print("[soak] ytvis: %d clip results over %d lanes x %d rounds, %d lane-rounds differ from the lone replay" % (n, LANES, ROUNDS, len(bad)))
        assert n >= 500 and not bad, "%d of %d lane-rounds differ from the lone replay" % (len(bad), len(sigs))
    finally:
        config.load_preset("defaults")


def test_sequence_path_soak():
    """... and on the sequence path: a 36-frame sequence (8 clips) through ``run_sequence_sharded`` 64 times = 512 clip results -- the
    rank's clips embedded 4 windows per encoder pass as hipGraph replays alternating over two lanes (so two passes are in flight),
    cross-clip foreground mask, own-clip clustering, label codes, pair tables, Hungarian chain, LUT gather.  Every iteration gives
    the same per-clip maps (bit sums), the same clustering records and the same stitched tracks."""
    import zlib
    from stemseg_amd import config, hip, pipeline
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    hip.require_gpu()
    config.load_preset("davis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    try:
        model = InferenceModel()
        sd = model._model.state_dict()
        new = {k: torch.from_numpy(np.asarray(synth.synth_param(k, v.shape, 47))).reshape(v.shape) for k, v in sd.items()}
        new["seediness_head.conv_out.weight"] = new["seediness_head.conv_out.weight"] * 40.0
        model._model.load_state_dict(new)
        pipe = ClipPipeline(model, seediness_thresh=0.5)
        pipe.model.overlap_decoders = False
        n = 36
        frames = torch.as_tensor(synth.synth_frames(n, 192, 320, seed=47).astype(np.float32).transpose(0, 3, 1, 2) - 110.0).cuda().contiguous()
        eh = model._model.embedding_head
        sums = []

        def embed_many(my):
            blocks = pipe.embed_many(frames, my, batch=4, lanes=2)
            sums.append(torch.stack([b.view(torch.int32).sum(dtype=torch.int64) for b in blocks]))
            return blocks
        seen = []
        for it in range(64):
            (track, counts, _), _, _, _, meta = pipeline.run_sequence_sharded(
                n, None, pipe.tg.chainer, "davis", frame_overlap=4, seediness_thresh=0.5, embed_many_fn=embed_many,
                channel_split=(eh.embedding_size, eh.variance_channels), outputs_on_cpu=False)
            seen.append((zlib.crc32(torch.cat([t.cpu() for t in track]).numpy().tobytes()), sorted(counts.items()), [m["instance_labels"] for m in meta]))
        torch.cuda.synchronize()
        sums = torch.stack(sums).cpu()
        bad_maps = int((sums != sums[0:1]).any(1).sum())
        bad_tracks = sum(1 for s_ in seen if s_ != seen[0])
        print("[soak] sequence: %d clip results over %d sequences; %d sequences with differing maps, %d with differing tracks (%d ids, %d fg points)"
              % (sums.numel(), len(seen), bad_maps, bad_tracks, len(seen[0][1]), sum(c for _, c in seen[0][1])))
        assert sums.numel() >= 500 and len(seen[0][1]) >= 2
        assert bad_maps == 0 and bad_tracks == 0
    finally:
        config.load_preset("defaults")
