"""Worker of tests/test_gpu_nccl.py (run as a script, one process, one GPU): the clip-parallel sequence path on a REAL RCCL process
group of one rank.  A world-1 ``nccl`` group runs the same device-tensor collectives as world 8 -- ``dist.all_gather`` of the
seediness planes and ``dist.all_gather_into_tensor`` of the label-code planes (stemseg_amd/pipeline.py, TorchComm) -- which no test
and no driver run had ever executed.  Prints ONE JSON line."""
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "stem-seg_amd")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def crc_of(track):
    return zlib.crc32(torch.cat([t.cpu() for t in track]).numpy().tobytes())


def golden_chain(tag):
    """A reference-generated chainer golden (tests/golden/chainer.npz) through run_sequence_sharded on the current group."""
    from stemseg_amd import config, pipeline
    from stemseg_amd.inference.clusterers import SequentialClustering
    from stemseg_amd.inference.online_chainer import OnlineChainer
    from tests.test_distributed_cpu import _case
    emb, bw, sd, fg, clips, overlap, exp = _case(tag)
    config.load_preset("davis")
    emb_d, bw_d, sd_d, fg_d = (torch.as_tensor(np.ascontiguousarray(a)).cuda() for a in (emb, bw, sd, fg))

    def embed(frames):
        idx = torch.as_tensor(frames, device="cuda")
        return emb_d[:, idx].contiguous(), bw_d[:, idx].contiguous(), sd_d[:, idx].contiguous()
    chainer = OnlineChainer(SequentialClustering(0.5, 0.3, 0.8, 2, [0.3, 0.3], "cuda:0"), 1.0)
    stats = {}
    (track, counts, life), _, _, _, meta = pipeline.run_sequence_sharded(
        fg.shape[0], embed, chainer, "davis", frame_overlap=overlap, fg_mask_fn=lambda entries, thr: fg_d, stats=stats)
    ok = len(track) == len(exp["track"]) and all(np.array_equal(l.cpu().numpy(), e) for l, e in zip(track, exp["track"]))
    ok = ok and sorted(counts.items()) == exp["counts"] and sorted(life.items()) == exp["life"]
    ok = ok and [m["instance_labels"] for m in meta] == exp["instance_labels"]
    return bool(ok), crc_of(track), stats


def real_sequence(pipe, frames, n, thr):
    """bench.py --sequence at reduced size: embed_many (hipGraph replays on two lanes) -> both exchanges -> chain."""
    from stemseg_amd import pipeline
    eh = pipe.model._model.embedding_head
    stats = {}
    (track, counts, _), _, _, _, _ = pipeline.run_sequence_sharded(
        n, None, pipe.tg.chainer, "davis", frame_overlap=4, seediness_thresh=thr, stats=stats, channel_split=(eh.embedding_size, eh.variance_channels),
        embed_many_fn=lambda my: pipe.embed_many(frames, my, batch=4, lanes=2), outputs_on_cpu=False)
    torch.cuda.synchronize()
    return crc_of(track), len(counts), stats


def main():
    from stemseg_amd import config, hip
    from stemseg_amd.modeling.inference_model import InferenceModel
    from stemseg_amd.pipeline import ClipPipeline
    from tests import synth
    hip.require_gpu()
    torch.cuda.set_device(0)
    config.load_preset("davis")
    config.cfg.MODEL.BACKBONE.TYPE = "R-50-FPN"
    model = InferenceModel()
    names = [(k, v.shape) for k, v in model._model.state_dict().items()]
    sd = synth.synth_state_dict(names, 11)
    sd["seediness_head.conv_out.weight"] = sd["seediness_head.conv_out.weight"] * 25.0
    model._model.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(model._model.state_dict()[k].shape) for k, v in sd.items()})
    pipe = ClipPipeline(model, seediness_thresh=0.4)
    model.overlap_decoders = False
    n = 36
    frames = (torch.from_numpy(synth.synth_frames(n, 96, 160, seed=11).astype(np.float32)).permute(0, 3, 1, 2) - 110.0).cuda().contiguous()
    # 1. without a process group: the buffers are used in place
    crc0, ids0, st0 = real_sequence(pipe, frames, n, 0.4)
    ok0, gcrc0, gst0 = golden_chain("seq20_ov4")
    assert st0["collectives_run"] == 0 and gst0["collectives_run"] == 0
    # 2. on a world-1 RCCL group: both collectives run on device tensors
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    warm = torch.ones(4, device="cuda")
    dist.all_reduce(warm)                                  # communicator set-up outside the checks below
    torch.cuda.synchronize()
    crc1, ids1, st1 = real_sequence(pipe, frames, n, 0.4)
    crc2, _, st2 = real_sequence(pipe, frames, n, 0.4)     # (again: the embed graphs replay while the RCCL watchdog thread is alive)
    res = {}
    for tag in ("seq20_ov4", "seq14_ov6", "seq8_single", "long"):
        ok, gcrc, gst = golden_chain(tag)
        res[tag] = dict(ok=ok, crc=gcrc, backend=gst["backend"], collectives_run=gst["collectives_run"], calls=gst["comm_calls"],
                        allgather_ms=round(gst["allgather_ms"], 4))
    dist.barrier()
    dist.destroy_process_group()
    out = dict(no_group=dict(crc=crc0, ids=ids0, golden_ok=ok0, golden_crc=gcrc0),
               nccl_world1=dict(crc=crc1, crc_again=crc2, ids=ids1, backend=st1["backend"], collectives_run=st1["collectives_run"], calls=st1["comm_calls"],
                                allgather_seediness_us=round(1e3 * st1["allgather_seediness_ms"], 1), allgather_codes_us=round(1e3 * st1["allgather_codes_ms"], 1)),
               goldens=res)
    print("NCCL_WORLD1 " + json.dumps(out))


if __name__ == "__main__":
    main()
