#!/usr/bin/env python3
"""bench.py -- clips/sec of the embed+cluster hot path on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = a batch of --clips-per-step (default 4) synthetic clips (each T=8 frames, padded 480x864, ResNet-101-FPN,
DAVIS heads: embedding decoder + separate seediness decoder): ONE encoder pass over all their frames (the encoder is
per-frame), then per clip 3-D decoders -> fused heads -> fg mask -> fg gather -> sequential clustering -> read-back of the
clustering record (K, instance list), with the input frames already resident in HBM.  The step is captured as a hipGraph
per lane; --lanes (default 3) steps are in flight on their own streams and workspaces.  value = clips / s.  Every clip result of the
run is checked bitwise against the first result of its input batch (a mismatch: exit code 5, no line).  At N = 1 the same step is then
timed in the two reference-width MFMA modes as well (``alt_precision``: bf16x6 and f32, shorter regions).  Clips are independent, so
ranks share nothing in ``value`` (weak scaling, no data-path collective; barrier / max-over-ranks around the timed region only).
The same process then runs BASELINE configs[3] -- one 64-frame sequence sharded over the ranks, BOTH all-gathers (RCCL at N > 1)
inside its timed region -- and attaches it as ``sequence`` (strong scaling, per-collective times, the ranks the process group saw,
and a label checksum that is the same at every N); ``--no-sequence-leg`` skips it.  Prints ONE JSON line on rank 0.

    python bench.py --gpus N                          # N > 1 without a launcher: re-executes itself under torch.distributed.run
    python bench.py --workload davis|ytvis|kitti      # BASELINE configs[1] (default) | configs[2] | configs[4]
    python bench.py --sequence [--frames 64|36]      # BASELINE configs[3]: ONE long sequence sharded over the ranks

--sequence: a step = the whole sequence through ``pipeline.run_sequence_sharded`` -- clips dealt to the ranks in contiguous blocks
(64 frames / overlap 4 -> 15 clips; 36 frames -> exactly 8, one per GPU at N = 8); every rank embeds its block of clips (up to 8
windows per encoder pass; frames shared by neighbouring clips pass the trunk once; full batches as hipGraph replays), all-gather #1
(RCCL over xGMI) of the SEEDINESS planes -> cross-clip foreground mask; every rank gathers + clusters ITS OWN clips with
label_start = 1; all-gather #2 of the one-byte label codes; the Hungarian chain runs on label-pair tables (host, microseconds per
clip).  Both collectives are INSIDE the timed region.  value = clips / s of the whole job (strong scaling: the sequence is
fixed); the line also carries the exchanges' bytes / time and a checksum of the stitched track labels that IS the same at
every N (the encoder plans its launches for a fixed frame count, so a clip's embeddings do not depend on how many clips share its
pass).  The default mode and this one share the model, weights and kernels.

--workload: ``davis`` = BASELINE configs[1] (the metric's configuration, the default: T=8, 480x854 -> 480x864, R-101-FPN,
embedding + seediness decoders); ``ytvis`` = configs[2] (T=8, 360x640 -> 384x640, youtube_vis.yaml heads: in-head seediness, 41+1
class semseg decoder, --resize_embeddings: x4 trilinear of the head outputs and clustering at FULL resolution); ``kitti`` =
configs[4] (T=8, 375x1242 --max_dim 1948 -> 608x1952, kitti_mots_2.yaml: xyt embeddings, 3+1 class semseg decoder, foreground
from the semseg head).  Same JSON schema for all three.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stem-seg_amd"))

import torch  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA, dense
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_conv3d_block4x_latest.json")   # written by tools/pmc_conv.py's summary step
PEAK_MFMA_BF16_TFLOPS = 2500.0    # same guide: bf16 MFMA, dense (the 2:1-sparse 5 PF figure is not used)
PEAK_HBM_GBPS = 8000.0            # same guide: HBM3E, ~8 TB/s
BACKBONE = "R-101-FPN"
# BASELINE.json configs -> concrete shapes: padded network input H x W, the un-padded (valid) size, preset
WORKLOADS = {
    "davis": dict(T=8, H=480, W=864, valid=(480, 854), preset="davis", min_dim=480, max_dim=854, resize=1.0, model_kw={}, dim_mode="xyff",
                  name="BASELINE configs[1]: DAVIS-shape clips T=8 480x854 (padded 480x864), %s, embedding + seediness decoders, "
                       "fg = seediness > 0.25, SequentialClustering (<= 20 instances)" % BACKBONE),
    "ytvis": dict(T=8, H=384, W=640, valid=(360, 640), preset="ytvis", min_dim=360, max_dim=640, resize=4.0, model_kw={"resize_scale": 4.0}, dim_mode="xyff",
                  name="BASELINE configs[2]: YouTube-VIS-shape clips T=8 360x640 (padded 384x640), %s, youtube_vis.yaml heads (embedding "
                       "decoder with in-head seediness, 41+1-class semseg decoder), --resize_embeddings: x4 trilinear of emb / bandwidth / "
                       "seediness / class logits, fg = semseg fg probability > 0.5, clustering at FULL resolution (1.97 M voxels)" % BACKBONE),
    "kitti": dict(T=8, H=608, W=1952, valid=(588, 1948), preset="kittimots", min_dim=800, max_dim=1948, resize=1.0, model_kw={}, dim_mode="xyt",
                  name="BASELINE configs[4]: KITTI-MOTS-shape clips T=8 375x1242 --max_dim 1948 -> 588x1948 (padded 608x1952), %s, "
                       "kitti_mots_2.yaml heads (xyt embeddings E = Ev = 3 with in-head seediness, 3+1-class semseg decoder), fg = semseg fg "
                       "probability > 0.5, SequentialClustering (MIN_SEEDINESS_PROB 0.95)" % BACKBONE),
}
WL = WORKLOADS["davis"]
T, H, W = WL["T"], WL["H"], WL["W"]


def select_workload(name):
    global WL, T, H, W
    WL = WORKLOADS[name]
    T, H, W = WL["T"], WL["H"], WL["W"]


def synth_weights_(module, seed, seediness_gain=30.0):
    """Random-init weights of the named architecture (no checkpoint is available offline).  He-normal convs,
    near-identity norms; the seediness head's last conv is scaled so that seediness spans (0, 1) -- with the default
    init every pixel sits near 0.5 < MIN_SEEDINESS_PROB and the clusterer would legitimately stop after one round
    (SURVEY.md section 8(d)); the gain makes it run its full <= 20 rounds, i.e. the expensive case."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if v.dim() == 0:
            new[k] = v.clone()
            continue
        n = torch.randn(v.shape, generator=g, dtype=torch.float32)
        if v.dim() >= 2:
            fan_in = v[0].numel()
            n = n * (2.0 / fan_in) ** 0.5
            if k.endswith("seediness_head.conv_out.weight") or k.endswith("conv_seediness.weight"):
                n = n * seediness_gain
        elif k.endswith("running_var"):
            n = 0.5 + 0.5 * n.abs()
        elif k.endswith("running_mean"):
            n = 0.1 * n
        elif k.endswith("bn3.weight"):
            n = 0.3 + 0.05 * n
        elif k.endswith(".weight"):
            n = 1.0 + 0.2 * n
        else:
            n = 0.1 * n
        new[k] = n
    module.load_state_dict(new)
    return new


def build_pipeline(device):
    from stemseg_amd import config
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    config.load_preset(WL["preset"])
    config.cfg.MODEL.BACKBONE.TYPE = BACKBONE
    config.cfg.INPUT.MIN_DIM, config.cfg.INPUT.MAX_DIM = WL["min_dim"], WL["max_dim"]
    model = InferenceModel(**WL["model_kw"])
    sd = synth_weights_(model._model, seed=1234)
    return ClipPipeline(model, device=device), sd


def make_clip(seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    frames = torch.randint(0, 256, (T, 3, H, W), generator=g, dtype=torch.int32).float()
    mean = torch.tensor([102.9801, 115.9465, 122.7717])[None, :, None, None]
    x = frames - mean
    vh, vw = WL["valid"]
    x[:, :, :, vw:] = 0.0                     # right / bottom padding is zero after mean subtraction (image_list.py:93-104)
    x[:, :, vh:, :] = 0.0
    return x.to(device)


def cpu_baseline(sd, frames_cpu, gpu_out=None, runs=3):
    """The CPU oracle (a restatement of the reference's PyTorch path, pinned against reference-generated goldens) on the
    host cores: ONE clip of the same workload, 1 warm-up + ``runs`` timed runs, median.  Threads: the
    box's cores up to 32 -- measured, torch's CPU conv / GroupNorm path gets SLOWER beyond that on the 2-socket host (141 s
    per clip with all 256 hardware threads) -- both counts are reported.  With ``gpu_out`` (the HIP path's result for the
    same clip) the two are also compared: the full-size parity check of this very run."""
    from oracle import pipeline as opipe
    n = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(n)
    times, out = [], None
    for i in range(1 + runs):
        t0 = time.time()
        out = opipe.embed_and_cluster_clip(frames_cpu, sd, BACKBONE, "xyff", 4, True, free_dim_stds=[0.3, 0.3])
        times.append(time.time() - t0)
    timed = sorted(times[1:])
    med = timed[len(timed) // 2]
    res = {"value": round(1.0 / med, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "host_hw_threads": os.cpu_count(), "kind": "port",
           "sample": "1 clip (T=8, 480x864, %s, DAVIS heads): 1 warm-up (%.1f s) + %d timed runs, median %.2f s (min %.2f, max %.2f); "
                     "%d fg points, %d instances; torch %s CPU backend"
                     % (BACKBONE, times[0], runs, med, timed[0], timed[-1], out["labels"].shape[0], len(out["meta"]["instance_labels"]), torch.__version__)}
    if gpu_out is not None:
        import numpy as np
        g_emb, g_seed = gpu_out["emb"].cpu().numpy(), gpu_out["seed"].cpu().numpy()
        g_fg = gpu_out["fg"].cpu().numpy().astype(bool)
        n = int(gpu_out["frame_offsets"][-1].item())
        g_lab = np.full(g_fg.size, -2, np.int64)
        g_lab[gpu_out["voxel_index"][:n].cpu().numpy()] = gpu_out["labels"][:n].cpu().numpy()
        c_fg = np.asarray(out["fg"]).astype(bool)
        c_lab = np.full(c_fg.size, -2, np.int64)
        c_lab[np.flatnonzero(c_fg.reshape(-1))] = np.asarray(out["labels"])
        both = (g_fg & c_fg).reshape(-1)
        res["parity_vs_hip_path"] = {
            "emb_max_abs_err": float(np.abs(g_emb - np.asarray(out["emb"])).max()),
            "seediness_max_abs_err": float(np.abs(g_seed - np.asarray(out["seed"])).max()),
            "fg_points_hip": int(g_fg.sum()), "fg_points_cpu": int(c_fg.sum()), "fg_mask_disagreements": int((g_fg != c_fg).sum()),
            "labels_identical_fraction_on_common_fg": float((g_lab[both] == c_lab[both]).mean()) if both.any() else None,
            "note": "fg / label differences come from points whose seediness or probability sits within the float tolerance of a threshold"}
    return res


def run_sequence_leg(args, pipe, device, rank, world, use_dist, frames_n, steps, warmup):
    """BASELINE configs[3]: one step = the whole sequence through ``pipeline.run_sequence_sharded`` (see the module docstring), both
    all-gathers inside the timed region.  Every rank ends with the same stitched tracks (asserted).  -> dict (identical on all ranks)."""
    import zlib
    import torch.distributed as dist
    from stemseg_amd.inference.main import get_subsequence_frames
    from stemseg_amd.pipeline import run_sequence_sharded
    F, overlap = frames_n, 4
    n_src = (F + T - 1) // T
    frames = torch.cat([make_clip(5000 + i, device) for i in range(n_src)], 0)[:F].contiguous()      # same frames on every rank
    clips, _ = get_subsequence_frames(F, T, "davis", overlap)
    prev_overlap = pipe.model.overlap_decoders
    pipe.model.overlap_decoders = False
    pipe.model.set_lane(0)
    eh = pipe.model._model.embedding_head
    split = (eh.embedding_size, eh.variance_channels)
    per_pass = 8                                      # up to 8 overlapping windows per encoder pass (36 frames), full batches as graph replays on two lanes

    def embed_many(my_clips):
        return pipe.embed_many(frames, my_clips, batch=per_pass, lanes=2, use_graph=not args.no_graph)
    chainer = pipe.tg.chainer
    stats, ag1_ms, ag2_ms, host_ms, res = {}, [], [], [], None

    def one():
        return run_sequence_sharded(F, None, chainer, "davis", frame_overlap=overlap, seediness_thresh=0.25, stats=stats,
                                    embed_many_fn=embed_many, channel_split=split, outputs_on_cpu=False)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
    for _ in range(max(warmup, 1)):
        res = one()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = one()
        ag1_ms.append(stats["allgather_seediness_ms"])
        ag2_ms.append(stats["allgather_codes_ms"])
        host_ms.append(stats.get("host_chain_ms", 0.0))
    sync()
    dt = time.perf_counter() - t0
    pipe.model.overlap_decoders = prev_overlap
    (track, counts, life) = res[0]
    crc = zlib.crc32(torch.cat([t.cpu() for t in track]).numpy().tobytes()) if track else 0
    if os.environ.get("STEMSEG_BENCH_DUMP") and rank == 0:      # (debugging aid: the stitched labels of the last step, for comparisons across N)
        import numpy as np
        np.save(os.environ["STEMSEG_BENCH_DUMP"], torch.cat([t.cpu() for t in track]).numpy())
    ranks = rank_devices(device, rank, world, use_dist)
    if use_dist:
        t = torch.tensor([dt, float(crc)], dtype=torch.float64, device=device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmin = t.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dt = float(tmax[0].item())
        assert float(tmax[1].item()) == float(tmin[1].item()), "ranks disagree on the stitched tracks"
    n_clips = len(clips)
    med = lambda v: round(sorted(v)[len(v) // 2], 3) if v else 0.0      # noqa: E731
    backend = (os.environ.get("STEMSEG_BENCH_BACKEND", "nccl") if use_dist else "none")
    return {
        "metric": "clips/sec (T=8, 480p) embed+cluster, one %d-frame sequence sharded over the GPUs" % F,
        "value": round(n_clips * steps / dt, 4), "unit": "clips/s", "n_gpus": len(ranks), "ranks": ranks, "steps": steps, "warmup": warmup,
        "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True, "scaling": "strong",
        "config": {"workload": "BASELINE configs[3]: %d DAVIS-shape frames (480x854 -> 480x864), T=8, overlap %d -> %d clips dealt in contiguous blocks "
                               "to %d rank(s), %s, both decoders; all-gather #1 of the seediness planes -> fg mask; every rank clusters ITS clips with "
                               "label_start = 1; all-gather #2 of one-byte label codes + clustering records; Hungarian chain on label-pair tables (host)"
                               % (F, overlap, n_clips, world, BACKBONE),
                   "clips": n_clips, "clips_per_rank_max": (n_clips + world - 1) // world,
                   "embed": "up to %d windows per encoder pass, %s" % (per_pass, "eager" if args.no_graph else "full batches as hipGraph replays on 2 lanes"),
                   "encoder_plan_frames": int(pipe.model._model.backbone.plan_frames)},
        "exchange": {"collective": ("all_gather x2 per sequence (%s)" % ("RCCL over xGMI" if backend == "nccl" else backend + ": functional check, ranks share a GPU"))
                     if world > 1 else "none (one rank: the process group, if any, has a single member and the exchange buffers are used in place)",
                     "backend": backend, "ranks_in_group": world,
                     "bytes_received_per_rank": stats["allgather_bytes"], "inside_timed_region": True,
                     "allgather_1_seediness_us_median": round(1e3 * med(ag1_ms), 1), "allgather_2_codes_us_median": round(1e3 * med(ag2_ms), 1),
                     "host_chain_ms_median": med(host_ms)},
        "result": {"frames": len(track), "fg_points": int(sum(counts.values())), "highest_track_id": int(max(list(counts) + [0])),
                   "label_checksum_crc32": int(crc),
                   "note": "identical on every rank of a run (asserted) AND across N: every convolution of the encoder decides its tile and split-K "
                           "factor for a fixed planning frame count, so a clip's embeddings are bit-identical whatever shares its encoder pass "
                           "(tests/test_gpu_sharded.py asserts the checksum at virtual world 1 / 2 / 3 / 8)"}}


def sequence_mode(args, pipe, device, rank, world, use_dist):
    """``--sequence``: the sequence leg as the line of its own."""
    from stemseg_amd import hip
    leg = run_sequence_leg(args, pipe, device, rank, world, use_dist, args.frames, args.steps, args.warmup)
    if rank == 0:
        leg.update({"vs_baseline": None, "dtype": PRECISION_DTYPE[args.precision], "data": "synthetic",
                    "operand_significand_bits": hip.PRECISION_INFO[args.precision]["operand_significand_bits"]})
        print(json.dumps(leg))


def rank_devices(device, rank, world, use_dist):
    """[{rank, device, name}] as the process group actually sees them (all-gathered) -- n_gpus in the line is len() of this."""
    import torch.distributed as dist
    if device.type != "cuda":
        mine = [rank, -1]
    else:
        mine = [rank, device.index]
    if not use_dist:
        return [{"rank": mine[0], "device": mine[1]}]
    t = torch.tensor(mine, dtype=torch.int64, device=device)
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [{"rank": int(o[0]), "device": int(o[1])} for o in outs]


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher environment: run this very command line under
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``, one rank per GPU over RCCL.
    Refuses (exit code 3, nothing printed on stdout) when the box has fewer than N devices."""
    import subprocess
    backend = os.environ.get("STEMSEG_BENCH_BACKEND", "nccl")
    stub = os.environ.get("STEMSEG_BENCH_STUB") == "1"
    if backend == "nccl" and not stub:
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print("[bench] --gpus %d requested but only %d device(s) are visible: not printing a line" % (args.gpus, n_dev), file=sys.stderr)
            sys.exit(3)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["STEMSEG_BENCH_CHILD"] = "1"
    print("[bench] launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


def stub_main(args, rank, world, use_dist):
    """STEMSEG_BENCH_STUB=1 (tests only): the launcher, the process group, the barrier / max-over-ranks timing and the JSON
    line of the real bench with a no-op step -- runs on a machine without a GPU (gloo)."""
    import torch.distributed as dist
    device = torch.device("cpu")

    def sync():
        if use_dist:
            dist.barrier()
    for _ in range(args.warmup):
        time.sleep(0.001)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002)
    sync()
    dt = time.perf_counter() - t0
    ranks = rank_devices(device, rank, world, use_dist)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        # the same top-level keys (and the contract's roofline / cpu_baseline sub-keys) as the real line at ANY N, so that whatever parses the
        # N = 1 line parses the N = 8 one (tests/test_bench_launcher.py)
        res = {"metric": "clips/sec (T=8, 480p) embed+cluster", "value": round(args.steps * world / dt, 4), "unit": "clips/s",
               "n_gpus": len(ranks), "ranks": ranks, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "none", "operand_significand_bits": None, "data": "synthetic", "alt_precision": None,
               "config": {"workload": "STUB (no-op step): launcher / process-group test only"},
               "roofline": {"bound": "mfma", "achieved": None, "peak": None, "unit": "TFLOP/s", "frac": None, "traffic": None},
               "sequence": None,
               "cpu_baseline": {"value": None, "unit": "clips/s", "cores": 0, "kind": "port", "sample": "stub"},
               "stub": True}
        assert set(LINE_KEYS) <= set(res), sorted(set(LINE_KEYS) - set(res))
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


# ``dtype`` names the arithmetic the matrix products compute in -- it never opens with "f32" unless the operands ARE fp32
PRECISION_DTYPE = {
    "f32": "f32 (fp32 operands on v_mfma_f32_32x32x2_f32, fp32 accumulate)",
    "bf16x6": "bf16x6: fp32 emulated from 3 bf16 terms per operand (exact split, 24-bit operand significands, fp32 exponent range), 6 bf16 MFMA "
              "products per fp32 product, fp32 accumulate",
    "f16x3": "f16x3: fp32 emulated from 2 fp16 terms per operand (22-bit operand significands; fp32 has 24), power-of-two operand scaling, 3 fp16 MFMA "
             "products per fp32 product, fp32 accumulate; |activation| >= 2.6e5 -> non-finite, flagged on the device, clip re-run in bf16x6",
}


PRODUCTS = {"bf16x6": 6.0, "f16x3": 3.0}
# top-level keys of the line, at every N (the real line is asserted against this list before it is printed; so is the stub's)
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
             "operand_significand_bits", "data", "alt_precision", "config", "roofline", "sequence", "cpu_baseline")


PRECISION_NOTE = {
    "f32": "v_mfma_f32_32x32x2_f32 on fp32 operands",
    "bf16x6": "every fp32 operand split EXACTLY into three bf16 terms (24 significand bits), the six products of weight >= 2^-16 on "
              "v_mfma_f32_32x32x16_bf16, fp32 accumulation; the dropped products are <= 2^-23 |a*b|.  Evidence that this is fp32-level arithmetic, "
              "not a reduced precision: max error vs an fp64 convolution = 0.78-1.56x that of the fp32-input MFMA kernel on every kernel class / "
              "tile / epilogue (tests/test_gpu_bf16x6.py); labels identical to the reference's own CPU results on its flows (tests/test_gpu_parity.py) "
              "and to the CPU oracle at full size in THIS run (cpu_baseline.parity_vs_hip_path); --precision f32 runs the fp32-input MFMA kernels",
    "f16x3": "every fp32 operand is scaled by a power of two (activations 2^-2; every OUTPUT CHANNEL's weights so that its largest lands in [2^13, 2^14)) and "
             "split into two fp16 terms (22 significand bits; the low activation term is stored as lo * 2^11 against a hi * 2^-11 weight operand, so "
             "the pair keeps 22 bits, or 2^-36 absolute, for 2.5e-4 <= |a| < 2.6e5); lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16, fp32 "
             "accumulation, accumulators scaled back exactly.  The dropped lo*lo product and the split remainder are <= 2^-22 |a*b|: measured "
             "against an fp64 convolution the error stays at the fp32-input MFMA kernel's own level on every kernel class / tile / epilogue and "
             "over activation magnitudes of 1e-4 ... 2e4 (tests/test_gpu_bf16x6.py, both modes); labels are checked against the reference's CPU results (tests/test_gpu_parity.py) and the "
             "CPU oracle at full size in THIS run (cpu_baseline.parity_vs_hip_path).  Half the matrix work of bf16x6; |activation| >= 2.6e5 "
             "overflows to non-finite outputs: the lane that reads such a clip back re-runs its batch in bf16x6 (GraphedStep.collect); "
             "--precision bf16x6 has fp32's full range.  NOT the reference's arithmetic width: the alt_precision legs of this line are",
}


def mark(msg):
    if os.environ.get("STEMSEG_BENCH_WATCHDOG"):
        print("[bench] " + msg, file=sys.stderr, flush=True)


class ClockSampler:
    """Shader clock (and board power) of one GPU, polled from the amdgpu driver's sysfs files by a thread while a region runs -- the
    UN-PROFILED clock of the very launches the line reports (the round-5 line divided an un-profiled time by a clock
    taken from a profiled PMC pass).  Sources, first that exists: hwmon ``freq1_input`` (Hz; label sclk), else the ``*`` line of
    ``pp_dpm_sclk``; power: hwmon ``power1_average`` / ``power1_input`` (microwatts).  No GPU work, no rocprof, no root.  The card is
    the one whose PCI address matches the torch device (else the only amdgpu card).  STEMSEG_SYSFS_DRM overrides the directory
    (tests)."""

    def __init__(self, device_index=0, period_s=0.01):
        import glob
        self.period = period_s
        self.freq_file = self.dpm_file = self.power_file = None
        self.source = self._stop = self._thread = None
        self.samples = []
        root = os.environ.get("STEMSEG_SYSFS_DRM", "/sys/class/drm")
        cards = []
        for d in sorted(glob.glob(os.path.join(root, "card[0-9]*"))):
            dev = os.path.join(d, "device")
            if os.path.exists(os.path.join(dev, "pp_dpm_sclk")) or glob.glob(os.path.join(dev, "hwmon", "hwmon*", "freq1_input")):
                cards.append(dev)
        want = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:  # noqa: BLE001
            pass
        pick = [c for c in cards if want and os.path.basename(os.path.realpath(c)).startswith(want)]
        dev = pick[0] if pick else (cards[0] if len(cards) == 1 else (cards[device_index] if device_index < len(cards) else None))
        if dev is None:
            return
        self.card = os.path.basename(os.path.dirname(dev))
        for hw in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))):
            f = os.path.join(hw, "freq1_input")
            if os.path.exists(f) and self.freq_file is None:
                self.freq_file, self.source = f, "sysfs hwmon freq1_input (sclk)"
            for pf in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(hw, pf)) and self.power_file is None:
                    self.power_file = os.path.join(hw, pf)
        if self.freq_file is None and os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            self.dpm_file, self.source = os.path.join(dev, "pp_dpm_sclk"), "sysfs pp_dpm_sclk (current level)"

    def available(self):
        return self.source is not None

    def _read_mhz(self):
        try:
            if self.freq_file:
                return float(open(self.freq_file).read().strip()) / 1e6
            for line in open(self.dpm_file).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:  # noqa: BLE001
            pass
        return None

    def _read_watts(self):
        try:
            return float(open(self.power_file).read().strip()) / 1e6 if self.power_file else None
        except Exception:  # noqa: BLE001
            return None

    def start(self):
        import threading
        if not self.available():
            return self
        self.samples = []
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                mhz = self._read_mhz()
                if mhz is not None and mhz > 0:
                    self.samples.append((mhz, self._read_watts()))
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        """-> {"sclk_ghz_mean", "sclk_ghz_min", "sclk_ghz_max", "power_w_mean", "samples", "source"} or None"""
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join(timeout=2.0)
        self._thread = None
        if not self.samples:
            return None
        f = [m for m, _ in self.samples]
        w = [p_ for _, p_ in self.samples if p_ is not None]
        return {"sclk_ghz_mean": round(sum(f) / len(f) / 1e3, 4), "sclk_ghz_min": round(min(f) / 1e3, 4), "sclk_ghz_max": round(max(f) / 1e3, 4),
                "power_w_mean": round(sum(w) / len(w), 1) if w else None, "samples": len(f), "source": self.source}


def main():
    if os.environ.get("STEMSEG_BENCH_WATCHDOG"):       # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["STEMSEG_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60, help="timed steps (4 clips each by default): ~2.6 s of GPU time at the default")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the bf16x6 / f32 legs (N = 1, davis workload only)")
    ap.add_argument("--alt-steps", type=int, default=16, help="timed steps of each reference-width leg")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "bf16x6", "f16x3"],
                    help="MFMA mode of every convolution: f16x3 (default, the library's default) = two fp16 terms of the power-of-two-scaled "
                         "operands, three products, fp32 accumulate (fp32-level results); bf16x6 = exact three-term bf16 split, six products "
                         "(fp32-level, fp32's exponent range); f32 = fp32-input MFMA")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly instead of replaying the captured hipGraph of the step (use under rocprofv3)")
    ap.add_argument("--graph", action="store_true", help=argparse.SUPPRESS)      # (the default; kept for old command lines)
    ap.add_argument("--clips-per-step", type=int, default=None,
                    help="clips that share one encoder pass per step (frames are independent in the encoder; stacking clips fills "
                         "its small-map launches); decoders, fg gather and clustering run per clip.  1 = one clip per step")
    ap.add_argument("--lanes", type=int, default=3,
                    help="captured steps in flight on one GPU, each with its own workspaces and stream (graph mode): the kernels of "
                         "one step fill the tail rounds and memory-bound phases of the other (config.determinism reports whether every clip result of the run "
                         "was bit-identical to the first one of its input batch)")
    ap.add_argument("--sequence", action="store_true",
                    help="BASELINE configs[3]: one long sequence, clips sharded over the ranks in contiguous blocks; all-gather #1 of the seediness "
                         "planes, own-clip clustering, all-gather #2 of one-byte label codes (both inside the timed region), Hungarian chain on "
                         "label-pair tables (see the module docstring)")
    ap.add_argument("--frames", type=int, default=64, help="--sequence: frames of the sequence (64 -> 15 clips at overlap 4; 36 -> 8)")
    ap.add_argument("--workload", default="davis", choices=sorted(WORKLOADS), help="BASELINE config to run (default: configs[1], the metric's)")
    ap.add_argument("--fp32-stem", action="store_true", help="A/B: the exact fp32-MFMA stem in f16x3 mode too (default there: space-to-depth + 4x4 conv in f16x3)")
    ap.add_argument("--no-decoder-batch", action="store_true", help="A/B: the decoders clip by clip instead of all clips of a step per launch (same bits)")
    ap.add_argument("--no-sequence-leg", action="store_true", help="clip bench: skip the attached BASELINE configs[3] leg (``sequence`` in the line)")
    ap.add_argument("--sequence-steps", type=int, default=4, help="timed sequences of the attached leg (1 warm-up)")
    ap.add_argument("--plan-frames", type=int, default=None,
                    help="frames the encoder plans its launches for (default: the model's own constant, ResNetFPN.plan_frames = 32, whatever "
                         "--clips-per-step / --sequence say: two batchings of one job give the same bits).  A throughput knob for A/B runs only")
    ap.add_argument("--fuse-tail", type=int, default=None, help="A/B: stages whose bottleneck tails run fused (bit mask 1 | 2 | 4; default: the model's, 7; 0: none; + 8: stage 3 on the 16-column form)")
    ap.add_argument("--no-fold-conv4", action="store_true", help="A/B: the decoders' tail step by step (fuse convs, 128 / 256-channel up-samplings, heads on conv_4's output)")
    ap.add_argument("--no-linear-tail", action="store_true", help="A/B: the decoders' tail with conv_16 / conv_8 and their up-samplings as launches, only conv_4 folded into the head weights")
    ap.add_argument("--no-overlap", action="store_true", help="run both decoders and all their branches on one stream")
    ap.add_argument("--graph-overlap", action="store_true", help="capture the graph WITH the fork/join branch streams (experimental)")
    args = ap.parse_args()
    args.graph = not args.no_graph
    if args.clips_per_step is None:          # default: 4 independent clips per encoder pass; --sequence: 8 overlapping windows (36 frames)
        args.clips_per_step = 8 if args.sequence else (2 if args.workload == "kitti" else 4)      # (a KITTI clip is 2.9 DAVIS clips of pixels)

    select_workload(args.workload)
    if args.sequence and args.workload != "davis":
        ap.error("--sequence is BASELINE configs[3] (DAVIS-shape frames); --workload applies to the clip bench")
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ                  # under torch.distributed.run (ours or the driver's)
    if args.gpus > 1 and not launched:
        self_launch(args)                                                            # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        print("[bench] --gpus %d but the launcher started %d rank(s): refusing to print a line" % (args.gpus, world), file=sys.stderr)
        sys.exit(4)
    import torch.distributed as dist
    use_dist = world > 1 or launched
    stub = os.environ.get("STEMSEG_BENCH_STUB") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # STEMSEG_BENCH_BACKEND=gloo: functional check of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices,
        # the exchange goes through the host) -- never a performance number; the line's config says so
        backend = "gloo" if stub else os.environ.get("STEMSEG_BENCH_BACKEND", "nccl")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if stub:
        return stub_main(args, rank, world, use_dist)
    if use_dist:
        if backend == "nccl" and torch.cuda.device_count() < world:
            print("[bench] %d ranks but only %d device(s) visible: refusing to print a line" % (world, torch.cuda.device_count()), file=sys.stderr)
            sys.exit(3)
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from stemseg_amd import hip
    hip.require_gpu()
    pipe, sd = build_pipeline(device)
    pipe.model.set_precision(args.precision)
    # The planning frame count is a MODEL constant (ResNetFPN.plan_frames, 32): the bench never derives it from its batching knobs, so
    # ``--clips-per-step 2`` and ``--clips-per-step 4`` produce the same bits for the same clip; --plan-frames
    # overrides it for A/B runs and is then named in the line.
    if args.plan_frames is not None:
        pipe.model._model.backbone.plan_frames = int(args.plan_frames)
    args.plan_frames = int(pipe.model._model.backbone.plan_frames)
    if args.fuse_tail is not None:
        pipe.model._model.backbone.fuse_tail = int(args.fuse_tail)
    if args.no_fold_conv4 or args.no_linear_tail:
        for name in ("embedding_head", "seediness_head", "semseg_head"):
            head = getattr(pipe.model._model, name, None)
            if head is not None:
                head.fold_linear_tail = False
                if args.no_fold_conv4:
                    head.fold_conv4 = False
    if args.sequence:
        sequence_mode(args, pipe, device, rank, world, use_dist)
        if use_dist:
            dist.destroy_process_group()
        return
    NC = max(1, args.clips_per_step)
    clips = [torch.cat([make_clip(1000 + rank * 97 + i * NC + c, device) for c in range(NC)], 0) for i in range(2)]
    overlap = not (args.graph or args.no_overlap)

    def run_leg(precision, steps, warmup, lane0):
        """One measured leg in MFMA mode ``precision``: warm-up, hipGraph capture per lane (lane ids from ``lane0``: own workspaces),
        the timed region (barrier + synchronize on both sides), the determinism check, and the eager single-stream roofline pass.
        -> dict(dt, meta, determinism, prof, n_roof, lanes, graph)"""
        pipe.model.set_precision(precision)
        # Determinism monitor: the steps alternate between two clip batches, so every result of a batch must be bit-identical to the
        # batch's first one, whichever lane produced it and whatever else was in flight.  Per clip one tiny device-side reduction (int64
        # sum of the fp32 BIT PATTERNS of the embedding map: exact, order-independent), compared once after the timed region.
        bitsums = {0: [], 1: []}

        fallbacks = [0]

        def read_back(outs, batch=None, lane=None):        # the consumer's read-back (K, centres): one small D2H per clip
            if lane is not None:
                # graph lane: GraphedStep.collect reads the records; a batch with a non-finite head output is re-run in bf16x6 on the lane
                got, metas = lane.collect()
                fallbacks[0] += int(got[0] is not outs[0])
                outs = got
            else:
                metas = [hip.read_cluster_meta(o["meta"], o.get("status")) for o in outs]      # (eager steps: raises on a non-finite head output)
            if batch is not None:
                bitsums[batch].append(torch.stack([o["emb"].view(torch.int32).sum(dtype=torch.int64) for o in outs]))
            return metas[-1]

        def step(i):
            return read_back(pipe.step_batch(clips[i % len(clips)], NC), i % len(clips))

        def sync():
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()

        meta = None
        pipe.model.set_lane(lane0)
        pipe.model.batch_decoders = not args.no_decoder_batch
        pipe.model._model.backbone.stem_s2d = not args.fp32_stem
        if not overlap:
            pipe.model.overlap_decoders = False      # (the captured graph is single-stream; warm up in the same mode)
        for i in range(max(warmup, 1)):
            meta = step(i)
        sync()
        mark("warmup done (%s)" % precision)
        # The ~330 launches of a step are captured ONCE into a hipGraph (ClipPipeline.capture: encoder, both decoders, fg
        # mask, gather, clustering rounds, all on one stream) and replayed per clip: the launch-bound tail of small kernels
        # no longer pays per-launch host latency.  Inputs are copied into the graph's static frame buffer (device-to-device).
        graph, lanes = None, []
        if args.graph:
            try:
                lanes = [pipe.capture(clips[0], overlap=bool(args.graph_overlap), n_clips=NC, lane=lane0 + k) for k in range(max(1, args.lanes))]
                graph = lanes[0]
                mark("capture done")
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                print("[bench] hipGraph capture failed (%r); falling back to eager launches" % (e,), file=sys.stderr)
                graph = None
                pipe.model.overlap_decoders = overlap
                torch.cuda.synchronize()
        pending = [None] * len(lanes)
        pending_batch = [None] * len(lanes)

        def step_graph(i):
            """Step i goes to lane i % L: first consume (read back) what that lane produced L steps ago, then enqueue the new
            batch on the lane's stream -- L steps are in flight."""
            k = i % len(lanes)
            m = None
            if pending[k] is not None:
                with torch.cuda.stream(lanes[k].stream):
                    m = read_back(pending[k], pending_batch[k], lanes[k])
            pending[k] = lanes[k].run_async(clips[i % len(clips)])
            pending_batch[k] = i % len(clips)
            return m

        def drain():
            m = None
            for k in range(len(lanes)):
                if pending[k] is not None:
                    with torch.cuda.stream(lanes[k].stream):
                        m = read_back(pending[k], pending_batch[k], lanes[k])
                    pending[k] = None
            return m

        run = step_graph if graph is not None else step
        for i in range(2 * max(1, len(lanes))):
            meta = run(i) or meta
            mark("pre-run %d done" % i)
        meta = drain() or meta
        sync()
        hip.profile_enable(graph is None)
        sampler = ClockSampler(device.index or 0).start()      # (a host thread reading two sysfs files every 10 ms: no GPU work)
        t0 = time.perf_counter()
        for i in range(steps):
            meta = run(i) or meta
        meta = drain() or meta
        sync()
        dt = time.perf_counter() - t0
        clock_timed = sampler.stop()
        mark("timed region done")
        determinism = None
        if graph is not None:
            checked = mismatching = 0
            for b, rows in bitsums.items():
                if rows:
                    t = torch.stack([r.to(device) for r in rows]).cpu()
                    checked += int(t.numel())
                    mismatching += int((t != t[0:1]).sum())
            determinism = {"clip_results_checked": checked, "mismatching": mismatching, "overflow_fallbacks": fallbacks[0],
                           "what": "int64 sum of the fp32 bit patterns of every clip's embedding map, every step since the first replay (pre-runs and "
                                   "timed region, %d lanes in flight), against the first result of the same input batch; a mismatch makes the bench "
                                   "exit non-zero without a line" % len(lanes)}
            if mismatching:
                print("[bench] FAILED: %d of %d clip results differ bitwise from the first result of their batch (%s): no line is printed"
                      % (mismatching, checked, precision), file=sys.stderr)
                sys.exit(5)
        hip.profile_enable(True)
        hip.profile_read()
        # Roofline pass: under stream concurrency the per-launch elapsed times overlap and are not additive, so the dominant
        # kernel is timed (same hipEvent pairs, same clips) over a few extra steps with every launch on one stream.
        pipe.model.set_lane(lane0)
        pipe.model.overlap_decoders = False
        for i in range(2):
            step(i)
        hip.profile_read()
        n_roof = max(2, min(5, steps))
        sampler = ClockSampler(device.index or 0).start()
        for i in range(n_roof):
            step(i)
        prof = hip.profile_read()
        clock_roof = sampler.stop()
        hip.profile_enable(False)
        pipe.model.overlap_decoders = overlap
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dict(dt=dt, meta=meta, determinism=determinism, prof=prof, n_roof=n_roof, lanes=lanes, graph=graph, clock_timed=clock_timed, clock_roof=clock_roof,
                    first_bitsum=int(bitsums[0][0][0].item()) if bitsums[0] else None)

    def k3_class(prof, precision):
        """(achieved TFLOP/s, peak, ms, flop, launches) of the 3x3x3 implicit-GEMM class in mode ``precision``"""
        peak = PEAK_MFMA_F32_TFLOPS if precision == "f32" else PEAK_MFMA_BF16_TFLOPS / PRODUCTS[precision]
        k3 = [prof[t] for t in hip.PROFILE_CONV_TAGS["conv3x3x3"] if t in prof]
        ms, fl, launches = sum(p[0] for p in k3), sum(p[1] for p in k3), sum(p[2] for p in k3)
        return ((fl / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0), peak, ms, fl, launches

    leg = run_leg(args.precision, args.steps, args.warmup, 0)
    dt, meta, determinism, prof, n_roof, lanes, graph = (leg[k] for k in ("dt", "meta", "determinism", "prof", "n_roof", "lanes", "graph"))
    clock_timed, clock_roof, first_bitsum = leg["clock_timed"], leg["clock_roof"], leg["first_bitsum"]
    # Reference-width legs: the same step, graph replay and lanes in the modes whose arithmetic is at least as wide as
    # the reference's fp32 -- bf16x6 (24 significand bits per operand, fp32's exponent range) and f32 (fp32-input MFMA) -- each over
    # a shorter timed region, so that the line carries their throughput and roofline fraction next to the default mode's.
    alt = {}
    if world == 1 and not args.no_alt_precision and args.workload == "davis":
        for j, prec in enumerate(p_ for p_ in ("bf16x6", "f32") if p_ != args.precision):
            a = run_leg(prec, args.alt_steps, 1, 100 * (j + 1))
            ach_a, peak_a, ms_a, _, _ = k3_class(a["prof"], prec)
            alt[prec] = {"value": round(args.alt_steps * NC / a["dt"], 4), "unit": "clips/s", "dtype": PRECISION_DTYPE[prec],
                         "operand_significand_bits": hip.PRECISION_INFO[prec]["operand_significand_bits"],
                         "steps": args.alt_steps, "ms_per_step": round(1e3 * a["dt"] / args.alt_steps, 3),
                         "determinism_mismatching": a["determinism"]["mismatching"] if a["determinism"] else None,
                         "roofline_3x3x3": {"achieved": round(ach_a, 2), "peak": round(peak_a, 1), "frac": round(ach_a / peak_a, 4), "unit": "TFLOP/s (fp32-equivalent)"}}
            del a
        pipe.model.set_precision(args.precision)
        pipe.model.set_lane(0)
    ranks = rank_devices(device, rank, world, use_dist)
    if use_dist and os.environ.get("STEMSEG_BENCH_BACKEND", "nccl") == "nccl":
        assert len({r["device"] for r in ranks}) == world, "ranks share a device: %s" % (ranks,)
    # BASELINE configs[3] in the same process group: one 64-frame sequence sharded over the ranks, both all-gathers (RCCL at N > 1)
    # inside its timed region -- the data-path collectives of the framework, which the weak-scaling value above never touches
    seq_leg = None
    if not args.no_sequence_leg and args.workload == "davis":
        try:
            pipe.model.set_precision(args.precision)
            seq_leg = run_sequence_leg(args, pipe, device, rank, world, use_dist, 64, max(1, args.sequence_steps), 1)
        except Exception as e:  # noqa: BLE001  (never lose the clip number because the attached leg failed; at N > 1 a failure here is fatal for the group anyway)
            if use_dist and world > 1:
                raise
            import traceback
            traceback.print_exc()
            seq_leg = {"value": None, "error": repr(e)}
        pipe.model.set_lane(0)

    if rank == 0:
        clips_total = args.steps * world * NC
        # dominant kernel: the 3x3x3 implicit-GEMM conv (all tile shapes)
        ach, peak, ms, fl, launches = k3_class(prof, args.precision)
        per_clip = 1.0 / (n_roof * NC)

        def cls(tags):
            sel = [prof[t] for t in tags if t in prof]
            m, f = sum(q[0] for q in sel) * per_clip, sum(q[1] for q in sel) * per_clip
            return {"ms_per_clip": round(m, 3), "gflop_per_clip": round(f / 1e9, 1), "tflops": round(f / m / 1e9, 1) if m > 0 else None,
                    "frac_of_mfma_peak": round(f / m / 1e9 / peak, 3) if m > 0 else None}
        breakdown = {name: cls(tags) for name, tags in hip.PROFILE_CONV_TAGS.items()}
        # whole step against the MFMA roof: every convolution FLOP of a clip (encoder + both decoders) over the TIMED region
        conv_flop_clip = sum(prof[t][1] for tags in hip.PROFILE_CONV_TAGS.values() for t in tags if t in prof) * per_clip
        whole_tf = conv_flop_clip * clips_total / world / dt / 1e12
        # the streaming kernels against the HBM roof: algorithmic bytes (inputs read once + outputs written once) / elapsed
        hbm = []
        for tag, name in sorted(hip.PROFILE_HBM_TAGS.items()):
            if tag in prof and prof[tag][0] > 0:
                m_, by, n_ = prof[tag]
                hbm.append({"kernel": name, "launches_per_clip": round(n_ * per_clip, 2), "mb_per_clip": round(by * per_clip / 1e6, 2),
                            "us_per_clip": round(1e3 * m_ * per_clip, 1), "gb_per_s": round(by / m_ / 1e6, 1),
                            "frac_of_hbm_peak": round(by / m_ / 1e6 / PEAK_HBM_GBPS, 3)})
        traffic, traffic_note = None, "not collected in this process: PMC counters need their own rocprofv3 --pmc passes (tools/gpu_round.sh pmc)"
        tfile = TRAFFIC_FILE if args.precision == "f32" else TRAFFIC_FILE.replace("_latest", "_%s_latest" % args.precision)
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic, traffic_note = tj["gb_per_launch_group"], "OFFLINE (not measured by this run), %s: %s" % (os.path.relpath(tfile, ROOT), tj["note"])
            except Exception as e:  # noqa: BLE001
                traffic_note = "could not read %s: %r" % (tfile, e)
        # the clock the chip actually sustained, sampled from sysfs by a host thread WHILE the eager roofline pass (and the timed region) ran
        # -- un-profiled, the same launches `achieved` is made of; None when the box exposes no sclk file
        sclk = clock_roof["sclk_ghz_mean"] if clock_roof else None
        res = {
            "metric": "clips/sec (T=8, 480p) embed+cluster" if args.workload == "davis" else "clips/sec (T=8, %dx%d) embed+cluster" % (H, W),
            "value": round(clips_total / dt, 4), "unit": "clips/s",
            "n_gpus": len(ranks), "ranks": ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": PRECISION_DTYPE[args.precision],
            "operand_significand_bits": hip.PRECISION_INFO[args.precision]["operand_significand_bits"],
            "data": "synthetic",
            "alt_precision": alt if alt else None,
            "config": {"workload": WL["name"], "clips_per_step": NC, "steps_in_flight": len(lanes) if graph is not None else 1, "random_init": "He-normal; seediness head gain 30 so clustering runs its rounds",
                       "last_clip": {"K": int(meta.K), "n_points": int(meta.n_points)}, "determinism": determinism,
                       # int64 sum of the fp32 bit patterns of clip 0's embedding map: the same for every --clips-per-step / --lanes / N
                       "first_clip_bitsum": first_bitsum,
                       "precision": dict(hip.PRECISION_INFO[args.precision], mode=args.precision, note=PRECISION_NOTE[args.precision]),
                       "encoder_plan_frames": int(args.plan_frames),
                       # weight folds at load (exact algebra on the model, every output of the path is still computed): FrozenBN into its convolution
                       # (the reference's own inference form, resnet.py:49-60) and -- unless --no-fold-conv4 -- the decoders' conv_4 into the head weights
                       # (1x1x1, no bias, no activation, feeding only the 1x1x1 heads: W_heads . W_conv4 as one fp64 product rounded once; outputs
                       # within fp32 round-off of the two-step form, tests/test_gpu_parity.py::test_conv4_folded_into_the_heads_vs_the_two_step_form)
                       "weight_folds": ["FrozenBN -> conv"] + ([] if args.no_fold_conv4 else (["decoder conv_4 -> head weights"] if args.no_linear_tail else
                                        ["decoder linear tail (conv_16, conv_8, conv_4, heads; up-sampling commutes with 1x1x1 convs) -> per-level head matrices"]))},
            "roofline": {"bound": "mfma", "kernel": "conv_igemm_kernel (3x3x3, %s)" % {"f32": "fp32 MFMA 32x32x2", "bf16x6": "bf16x6 on MFMA 32x32x16 bf16; peak = 2500/6 fp32-equivalent TFLOP/s", "f16x3": "f16x3 on MFMA 32x32x16 f16; peak = 2500/3 fp32-equivalent TFLOP/s"}[args.precision],
                         "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s" if args.precision == "f32" else "TFLOP/s (fp32-equivalent: conv FLOPs / time; the MFMA pipe issues %d 16-bit products per fp32 product)" % PRODUCTS.get(args.precision, 1),
                         "frac": round(ach / peak, 4), "achieved_vs_fp32_input_mfma_peak": round(ach / PEAK_MFMA_F32_TFLOPS, 3),
                         "achieved_vs_bf16x6_roof": round(ach / (PEAK_MFMA_BF16_TFLOPS / 6.0), 3),
                         "frac_note": "the roof is the 16-bit MFMA peak / products per fp32 product (f16x3 3, bf16x6 6): it doubles whenever a mode halves the "
                                      "matrix work, so frac is not comparable across modes -- the two achieved_vs_* keys restate it against the earlier roofs",
                         "traffic": traffic, "traffic_offline": True, "traffic_note": traffic_note,
                         # flat (the driver keeps scalar keys of this dict): the un-profiled shader clock sampled during the eager roofline pass
                         # and the timed region, frac restated at that clock, and the reference-width legs of this very run
                         "sclk_ghz_roofline_pass": sclk, "sclk_ghz_timed_region": clock_timed["sclk_ghz_mean"] if clock_timed else None,
                         "power_w_timed_region": clock_timed["power_w_mean"] if clock_timed else None,
                         "frac_at_sampled_clock": round(ach / (peak * sclk / 2.4), 4) if sclk else None,
                         "ref_width_f32_clips_per_s": alt["f32"]["value"] if "f32" in alt else None,
                         "ref_width_f32_frac_3x3x3": alt["f32"]["roofline_3x3x3"]["frac"] if "f32" in alt else None,
                         "ref_width_bf16x6_clips_per_s": alt["bf16x6"]["value"] if "bf16x6" in alt else None,
                         "ref_width_bf16x6_frac_3x3x3": alt["bf16x6"]["roofline_3x3x3"]["frac"] if "bf16x6" in alt else None,
                         "power_bound": {"nominal_clock_ghz": 2.4, "roofline_pass": clock_roof, "timed_region": clock_timed,
                                         "note": "the roof assumes 2.4 GHz; under these MFMA streams the chip clocks to its power budget.  The clock here "
                                                 "is polled from the driver's sysfs sclk file by a host thread every 10 ms while the region runs (the whole eager "
                                                 "step / the whole timed region, all kernels -- not one kernel's clock, and never a profiled pass); "
                                                 "frac_at_sampled_clock = achieved / (peak x sclk / 2.4).  The same launches on all-zero activations -- "
                                                 "identical instruction stream -- run 20-25 % faster (profiles/r05i_dvfs_zero_inputs.txt, DESIGN.md section 5f)"},
                         "launches": launches, "avg_launch_ms": round(ms / max(launches, 1), 4),
                         "how": "hipEvent pairs (in-library profiler, on the launch's own stream) around every tagged launch over %d eager "
                                "single-stream steps right after the timed region; conv launches include their split-K reduce; a launch = one "
                                "kernel launch of the conv (the planner may cut the block_4x conv in two)" % n_roof,
                         "hip_graph_replay_in_timed_region": graph is not None,
                         "whole_step": {"bound": "mfma", "gflop_per_clip": round(conv_flop_clip / 1e9, 1), "achieved": round(whole_tf, 2),
                                        "peak": round(peak, 1), "unit": "TFLOP/s per GPU", "frac": round(whole_tf / peak, 4),
                                        "how": "all convolution FLOPs of the clips processed / the timed region itself (graph replay, %d steps in flight)"
                                               % (len(lanes) if graph is not None else 1)},
                         "conv_classes_eager": breakdown,
                         "hbm_kernels_eager": {"peak_gb_per_s": PEAK_HBM_GBPS, "bytes": "algorithmic: inputs read once + outputs written once", "kernels": hbm}},
        }
        res["sequence"] = seq_leg
        # CPU baseline: measured at N = 1 (rank 0, outside every timed region) and cached under .bench_cache/; a line at N > 1 carries the cached
        # object of the N = 1 run on the same box (``cached_from_n1_run``) or, without one, measures a single timed run while the other ranks
        # wait at the closing barrier -- so the first SCALE line is as complete as the N = 1 line
        if not args.no_cpu_baseline and args.workload == "davis":
            cache = os.path.join(ROOT, ".bench_cache", "cpu_baseline.json")
            try:
                if world > 1 and os.path.exists(cache):
                    cb = json.load(open(cache))
                    cb["cached_from_n1_run"] = True
                    res["cpu_baseline"] = cb
                else:
                    pipe.model.set_lane(0)
                    gpu_out = pipe.step(clips[0][:T].contiguous())
                    torch.cuda.synchronize()
                    res["cpu_baseline"] = cpu_baseline({k: v for k, v in sd.items()}, clips[0][:T].cpu(), gpu_out, runs=3 if world == 1 else 1)
                    if world == 1:
                        os.makedirs(os.path.dirname(cache), exist_ok=True)
                        json.dump(res["cpu_baseline"], open(cache, "w"))
            except Exception as e:  # noqa: BLE001  (never lose the GPU number because the baseline leg failed)
                res["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        res.setdefault("cpu_baseline", None)      # (--no-cpu-baseline / other workloads)
        assert set(LINE_KEYS) <= set(res), sorted(set(LINE_KEYS) - set(res))
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()            # (rank 0 may have spent ~10 s on the CPU baseline: leave together)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
