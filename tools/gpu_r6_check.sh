#!/bin/bash
# round 6 working call: GPU suite (or a -k subset: TESTS_K), default bench line, steady-state kernel trace.  Usage: bash tools/gpu_r6_check.sh [tag]
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${1:-r06a}
if [[ -z "${SKIP_TESTS:-}" ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider ${TESTS_K:+-k "$TESTS_K"} > gpurun_out/${R}_gpu_tests.log 2>&1; echo "tests exit $?"
  grep -E "passed|failed|error" gpurun_out/${R}_gpu_tests.log | tail -2 | cut -c1-200
  grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/${R}_gpu_tests.log | head -20 | cut -c1-250
fi
if [[ -z "${SKIP_BENCH:-}" ]]; then
  timeout 600 python bench.py ${BENCH_ARGS:-} > gpurun_out/${R}_bench.log 2>&1; echo "bench exit $?"; grep "^{" gpurun_out/${R}_bench.log | tail -1 > gpurun_out/${R}_bench_davis.json
  python - <<PY
import json
try:
    j = json.load(open("gpurun_out/${R}_bench_davis.json"))
    r = j["roofline"]
    print("value", j["value"], "ms/step", j["ms_per_step"], "frac", r["frac"], "sclk", r.get("sclk_ghz_roofline_pass"), r.get("sclk_ghz_timed_region"), "W", r.get("power_w_timed_region"),
          "frac@clk", r.get("frac_at_sampled_clock"), "f32", r.get("ref_width_f32_clips_per_s"), "x6", r.get("ref_width_bf16x6_clips_per_s"))
    for k, v in r["conv_classes_eager"].items():
        print(" ", k, v)
    for h in r["hbm_kernels_eager"]["kernels"]:
        print("   %-40s %8.1f us/clip %7.1f GB/s %.3f" % (h["kernel"], h["us_per_clip"], h["gb_per_s"], h["frac_of_hbm_peak"]))
    print(" seq", (j.get("sequence") or {}).get("value"), "cpu", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("parity_vs_hip_path"))
    print(" bitsum", j["config"].get("first_clip_bitsum"), "det", j["config"]["determinism"]["mismatching"])
except Exception as e:
    print("bench line unreadable:", e)
PY
  tail -3 gpurun_out/${R}_bench.log | cut -c1-300 | grep -v "^{"
fi
if [[ -z "${SKIP_PROF:-}" ]]; then
  rm -rf gpurun_out/prof_graph
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --no-sequence-leg --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
  db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_steady_state.txt 2>&1; head -45 gpurun_out/${R}_kernel_trace_steady_state.txt | cut -c1-150
  rm -rf gpurun_out/prof_graph gpucore.*
fi
ls /sys/class/drm/ 2>/dev/null | head; for f in /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input /sys/class/drm/card*/device/pp_dpm_sclk; do echo "$f: $(cat $f 2>&1 | tr '\n' ' ' | cut -c1-120)"; done 2>/dev/null | head -8
