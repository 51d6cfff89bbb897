#!/bin/bash
# round 4, call H: full GPU suite, smoke, default bench (with alt-precision legs + CPU baseline), soak-test sensitivity with the VALU stem
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider --durations=8 > gpurun_out/tests.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -2 | cut -c1-200
grep -E "^\[fullsize\]|^\[soak\]|^FAILED|^ERROR" gpurun_out/tests.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log | cut -c1-200
( time timeout 600 python bench.py ) > gpurun_out/bench_final.log 2>&1; echo "bench exit $?"; grep real gpurun_out/bench_final.log; grep "^{" gpurun_out/bench_final.log | tail -1 > gpurun_out/bench_final.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_final.json"))
print("davis", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["config"]["determinism"]["mismatching"], d["config"]["determinism"]["clip_results_checked"])
print("alt", json.dumps(d["alt_precision"]))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("parity_vs_hip_path"))
PY
STEMSEG_STEM=valu timeout 300 python -m pytest tests/test_gpu_soak.py -q -s -p no:cacheprovider 2>&1 | grep -E "soak|passed|failed" | cut -c1-200
