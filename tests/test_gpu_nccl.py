"""The RCCL code path of the clip-parallel sequence driver, executed for real: a world-1 ``nccl`` process
group on the GPU box (tests/nccl_world1_worker.py, its own process: a process group is process-wide state, and RCCL's watchdog
thread must not meet the other tests' graph captures).  ``run_sequence_sharded`` then issues ``dist.all_gather`` on the seediness
planes and ``dist.all_gather_into_tensor`` on the label-code planes -- device tensors, RCCL kernels on the caller's stream -- exactly
as at world 8 (/root/reference/stemseg/inference/main.py:93-103 and online_chainer.py:193-236 are what the exchange stands in for).
Asserted: the reference-generated chainer goldens are reproduced through the collectives, and the real embed -> exchange -> chain flow
gives the same label checksum with and without the group."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_sequence_path_on_a_world_1_rccl_group():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "stem-seg_amd")] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_world1_worker.py")], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("NCCL_WORLD1 ")]
    assert r.returncode == 0 and line, "worker failed (rc %d):\n%s\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-5000:])
    out = json.loads(line[-1][len("NCCL_WORLD1 "):])
    print("[nccl] " + json.dumps(out["nccl_world1"]))
    w = out["nccl_world1"]
    assert w["backend"] == "nccl" and w["collectives_run"] == 2
    assert w["calls"]["all_gather"] >= 1 and w["calls"]["all_gather_into_tensor_nccl"] == 1 and w["calls"]["all_gather_into_tensor_list"] == 0
    assert w["crc"] == w["crc_again"] == out["no_group"]["crc"] and w["ids"] == out["no_group"]["ids"] and w["ids"] >= 2
    assert out["no_group"]["golden_ok"]
    for tag, g in out["goldens"].items():
        assert g["ok"] and g["backend"] == "nccl" and g["collectives_run"] == 2 and g["calls"]["all_gather_into_tensor_nccl"] == 1, (tag, g)
    assert out["goldens"]["seq20_ov4"]["crc"] == out["no_group"]["golden_crc"]
