"""N "virtual ranks" of the clip-parallel sequence path inside ONE process (tests only): every rank is a thread, the two
collectives of stemseg_amd.pipeline.run_sequence_sharded are emulated with a barrier + shared slots.  On the GPU box (one
device) this runs the real HIP kernels of every rank's part -- own-clip clustering with label_start = 1, code planes, pair
tables, LUT gather -- with the plane / owner arithmetic of a world-N job."""
import threading

import torch


class ThreadComm(object):
    def __init__(self, rank, world, shared):
        self.rank, self.world, self.shared = rank, world, shared

    def _exchange(self, t):
        sh = self.shared
        sh["slots"][self.rank] = t.contiguous()
        if t.is_cuda:
            torch.cuda.synchronize()          # the other threads read this rank's tensor: make its producer kernels visible
        sh["barrier"].wait()
        got = [s.clone() for s in sh["slots"]]
        if t.is_cuda:
            torch.cuda.synchronize()
        sh["barrier"].wait()
        return got

    def all_gather(self, outs, t):
        for o, g in zip(outs, self._exchange(t)):
            o.copy_(g)

    def all_gather_into_tensor(self, out, t):
        for r, g in enumerate(self._exchange(t)):
            out[r].copy_(g)


def run_virtual_ranks(world, fn):
    """fn(comm) on ``world`` threads -> list of results by rank (exceptions re-raised)."""
    shared = {"slots": [None] * world, "barrier": threading.Barrier(world)}
    res, err = [None] * world, [None] * world

    def work(r):
        try:
            res[r] = fn(ThreadComm(r, world, shared))
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            shared["barrier"].abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return res
