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


def test_world_8_line_parses_with_every_key_of_the_n1_line():
    """The first 8-GPU run must be readable by whatever reads the N = 1 line: same top-level keys, the contract's roofline / cpu_baseline
    sub-keys, eight distinct ranks (stub step, gloo, the driver's own launcher form)."""
    import bench
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
            "--master-port", str(29700 + os.getpid() % 90), BENCH]
    r8 = subprocess.run(base + ["--gpus", "8", "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r8.returncode == 0, r8.stderr[-2000:]
    j8 = _line(r8.stdout)
    r1 = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=600)
    j1 = _line(r1.stdout)
    assert set(j1) == set(j8) and set(bench.LINE_KEYS) <= set(j8)
    assert j8["n_gpus"] == 8 and sorted(x["rank"] for x in j8["ranks"]) == list(range(8)) and j8["scaling"] == "weak"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in j8["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in j8["cpu_baseline"]


def test_clock_sampler_reads_sysfs_and_degrades_to_none(tmp_path, monkeypatch):
    """bench.ClockSampler: hwmon freq1_input (Hz) + power1_average (uW) of the amdgpu card; the '*' line of pp_dpm_sclk as fallback; None
    when the box exposes neither (this container)."""
    import time
    import bench
    card = tmp_path / "card0" / "device"
    hw = card / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("1950000000\n")
    (hw / "power1_average").write_text("1000000000\n")
    (card / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2100Mhz *\n")
    monkeypatch.setenv("STEMSEG_SYSFS_DRM", str(tmp_path))
    s = bench.ClockSampler(0, period_s=0.002).start()
    time.sleep(0.05)
    r = s.stop()
    assert r and r["sclk_ghz_mean"] == 1.95 and r["power_w_mean"] == 1000.0 and r["samples"] >= 3 and "freq1_input" in r["source"]
    (hw / "freq1_input").unlink()
    s = bench.ClockSampler(0, period_s=0.002).start()
    time.sleep(0.03)
    r = s.stop()
    assert r and r["sclk_ghz_mean"] == 2.1 and "pp_dpm_sclk" in r["source"]
    monkeypatch.setenv("STEMSEG_SYSFS_DRM", str(tmp_path / "nothing"))
    s = bench.ClockSampler(0).start()
    assert not s.available() and s.stop() is None
