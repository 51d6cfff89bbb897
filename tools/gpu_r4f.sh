#!/bin/bash
# round 4, call F: MFMA stem -- parity, timing, soak
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -p no:cacheprovider -k "encoder or stem" 2>&1 | grep -E "\[stem\]|passed|failed|Error|error|assert" | tail -12
timeout 600 python -m pytest tests/test_gpu_bf16x6.py -q -x -s --timeout 300 -p no:cacheprovider -k "checkpoint_like" 2>&1 | grep -E "checkpoint-like|passed|failed|Error|assert" | cut -c1-200 | tail -12
timeout 300 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/bench_mfma_stem.log 2>&1; tail -1 gpurun_out/bench_mfma_stem.log > gpurun_out/bench_mfma_stem.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_mfma_stem.json"))
print("davis", d["value"], d["config"]["determinism"]["mismatching"], d["config"]["determinism"]["clip_results_checked"])
for k in d["roofline"]["hbm_kernels_eager"]["kernels"]:
    if "stem" in k["kernel"] or "maxpool" in k["kernel"]: print(k)
PY
STEMSEG_STEM=valu timeout 300 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('davis valu stem', d['value'], d['config']['determinism']['mismatching'])"
timeout 900 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 300 --precision bf16x6 --max-reports 2 > gpurun_out/soak4_ytvis_bf16x6.txt 2>&1; echo "exit $?"; grep -n "S0 \|RESULT" gpurun_out/soak4_ytvis_bf16x6.txt | cut -c1-250 | tail -6
timeout 900 python tools/soak_probe.py --workload davis --lanes 3 --reps 300 --max-reports 2 > gpurun_out/soak4_davis_f16x3.txt 2>&1; echo "exit $?"; grep -n "S0 \|RESULT" gpurun_out/soak4_davis_f16x3.txt | cut -c1-250 | tail -6
