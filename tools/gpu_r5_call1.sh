#!/bin/bash
# round 5, GPU call 1: the whole GPU suite on the reworked library (planning frames, pruned modes, canaries, RCCL world-1, T=16, lane
# fallback), smoke, the default bench line (sequence leg attached), what the planning frame count costs, the graph co-residency probe on
# both stem forms, a conv sweep with the four-wave candidate tiles, a steady-state kernel trace.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r05a}
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/${R}_device.txt; nproc >> gpurun_out/${R}_device.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/${R}_device.txt
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 -p no:cacheprovider --durations=12 > gpurun_out/${R}_gpu_tests.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|error" gpurun_out/${R}_gpu_tests.log | tail -3 | cut -c1-300
grep -E "^FAILED|^ERROR" gpurun_out/${R}_gpu_tests.log | head -40 | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log | cut -c1-200
timeout 600 python bench.py > gpurun_out/bench_davis.log 2>&1; echo "bench exit $?"; grep "^{" gpurun_out/bench_davis.log | tail -1 > gpurun_out/${R}_bench_davis.json; cut -c1-200 gpurun_out/${R}_bench_davis.json; tail -5 gpurun_out/bench_davis.log | grep -v "^{" | cut -c1-300
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/%s_bench_davis.json" % __import__("os").environ.get("ROUND", "r05a")))
    print("value", j["value"], "alt", {k: v["value"] for k, v in (j.get("alt_precision") or {}).items()}, "roof", j["roofline"]["frac"], "seq", (j.get("sequence") or {}).get("value"), (j.get("sequence") or {}).get("result", {}).get("label_checksum_crc32"))
    print("classes", {k: (v["ms_per_clip"], v["frac_of_mfma_peak"]) for k, v in j["roofline"]["conv_classes_eager"].items()})
    print("cpu", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("parity_vs_hip_path"))
except Exception as e:
    print("no bench line:", e)
PY
# what the planning frame count costs: plan 0 = every launch decides on its real shape (the round-4 behaviour)
for spec in "4 32" "4 0" "1 32" "1 8" "1 0" "2 32" "2 0"; do
  set -- $spec
  timeout 200 python bench.py --clips-per-step $1 --plan-frames $2 --steps 30 --no-cpu-baseline --no-alt-precision --no-sequence-leg > gpurun_out/ab.log 2>&1
  echo "clips/step $1 plan_frames $2: $(grep '^{' gpurun_out/ab.log | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])' 2>&1 | tail -1)"
done
for pf in 32 0; do
  timeout 200 python bench.py --sequence --frames 64 --steps 4 --warmup 1 --plan-frames $pf > gpurun_out/ab.log 2>&1
  echo "sequence64 plan_frames $pf: $(grep '^{' gpurun_out/ab.log | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["value"], j["result"]["label_checksum_crc32"])' 2>&1 | tail -1)"
done
EXP=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_exp.so
STEMSEG_HIP_LIB=$EXP STEMSEG_STEM=valu timeout 400 python tools/graph_corun_probe.py --rounds 200 > gpurun_out/${R}_graph_corun_valu_stem.txt 2>&1; echo "corun valu exit $?"; grep -E "victim|total" gpurun_out/${R}_graph_corun_valu_stem.txt | cut -c1-200
STEMSEG_HIP_LIB=$EXP timeout 400 python tools/graph_corun_probe.py --rounds 100 > gpurun_out/${R}_graph_corun_mfma_stem.txt 2>&1; echo "corun mfma exit $?"; grep -E "total" gpurun_out/${R}_graph_corun_mfma_stem.txt | cut -c1-200
PREC=f16x3 SWEEP_T=32 ONLY=enc timeout 500 python tools/conv_sweep.py > gpurun_out/${R}_conv_sweep_f16x3_enc_T32.txt 2>&1; echo "sweep exit $?"; cut -c1-400 gpurun_out/${R}_conv_sweep_f16x3_enc_T32.txt
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --no-sequence-leg --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_steady_state.txt 2>&1; head -30 gpurun_out/${R}_kernel_trace_steady_state.txt | cut -c1-200
rm -f gpurun_out/prof_graph/*.db gpucore.*
