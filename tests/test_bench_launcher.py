"""bench.py's own multi-rank launcher (``python bench.py --gpus N`` with no launcher environment re-executes itself under
torch.distributed.run, one rank per device) and the driver's form (``python -m torch.distributed.run ... bench.py --gpus N``),
exercised on CPU with the stub step (STEMSEG_BENCH_STUB=1: gloo, no GPU): the line must report the ranks the process group saw."""
import json
import os
import subprocess
import sys

from tests.conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    env["STEMSEG_BENCH_STUB"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and [x["rank"] for x in j["ranks"]] == [0, 1] and j["steps"] == 3 and j["stub"] is True
    assert "launching 2 ranks" in r.stderr


def test_bench_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and "launching" not in r.stderr


def test_bench_under_the_drivers_launcher_and_gpus_mismatch():
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(29900 + os.getpid() % 90), BENCH]
    r = subprocess.run(base + ["--gpus", "2", "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2
    # a launcher that started a different number of ranks than --gpus says: no line at all
    r = subprocess.run(base + ["--gpus", "4", "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
