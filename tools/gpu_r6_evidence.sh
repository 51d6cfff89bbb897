#!/bin/bash
# round 6 evidence set on the shipped build (ONE gpurun call): GPU suite log, smoke, default bench line (alt-precision legs, attached sequence
# leg, CPU baseline, sampled clock), workload lines, sequence lines, functional N = 2 / 3, kernel trace, PMC passes on the block_4x conv and on
# the fused bottleneck tail, whole-step HBM PMC, lane soaks.  Usage: ROUND=r06 bash tools/gpu_r6_evidence.sh
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r06}
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/${R}_device.txt; nproc >> gpurun_out/${R}_device.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/${R}_device.txt
if [[ -z "${SKIP_TESTS:-}" ]]; then
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 -p no:cacheprovider --durations=10 > gpurun_out/${R}_gpu_tests.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|error" gpurun_out/${R}_gpu_tests.log | tail -2 | cut -c1-200
grep -E "^FAILED|^ERROR" gpurun_out/${R}_gpu_tests.log | head -20 | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log | cut -c1-200
fi
timeout 600 python bench.py > gpurun_out/bench_davis.log 2>&1; echo "bench exit $?"; grep "^{" gpurun_out/bench_davis.log | tail -1 > gpurun_out/${R}_bench_davis.json; cut -c1-160 gpurun_out/${R}_bench_davis.json
for wl in ytvis kitti; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_$wl.log 2>&1; echo "$wl exit $?"; grep "^{" gpurun_out/bench_$wl.log | tail -1 > gpurun_out/${R}_bench_$wl.json; cut -c1-150 gpurun_out/${R}_bench_$wl.json
done
for fr in 64 36; do
  timeout 300 python bench.py --sequence --frames $fr --steps 4 --warmup 1 > gpurun_out/bench_seq$fr.log 2>&1; echo "seq$fr exit $?"; grep "^{" gpurun_out/bench_seq$fr.log | tail -1 > gpurun_out/${R}_bench_seq$fr.json; cut -c1-150 gpurun_out/${R}_bench_seq$fr.json
done
for n in 2 3; do
  STEMSEG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus $n --steps 6 --warmup 2 --lanes 2 --no-alt-precision --sequence-steps 2 > gpurun_out/bench_n$n.log 2>&1; echo "n$n exit $?"
  grep "^{" gpurun_out/bench_n$n.log | tail -1 > gpurun_out/${R}_bench_davis_n${n}_gloo_functional.json
done
python - <<'PY'
import json, os
R = os.environ.get("ROUND", "r06")
for f in ("bench_davis", "bench_davis_n2_gloo_functional", "bench_davis_n3_gloo_functional", "bench_seq64", "bench_ytvis", "bench_kitti"):
    try:
        j = json.load(open("gpurun_out/%s_%s.json" % (R, f)))
        s = j.get("sequence") or j
        r = j.get("roofline") or {}
        cc = r.get("conv_classes_eager") or {}
        print(f, "n_gpus", j["n_gpus"], "value", j["value"], "| seq:", s.get("value"), "crc", (s.get("result") or {}).get("label_checksum_crc32"), "| cpu_baseline", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("cached_from_n1_run"),
              "| frac", r.get("frac"), "sclk", r.get("sclk_ghz_roofline_pass"), "| classes", {k: (v["ms_per_clip"], v["frac_of_mfma_peak"]) for k, v in cc.items()})
    except Exception as e:
        print(f, "unreadable:", e)
PY
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --no-sequence-leg --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_steady_state.txt 2>&1; head -6 gpurun_out/${R}_kernel_trace_steady_state.txt | cut -c1-170
rm -rf gpurun_out/prof_graph gpucore.*
PREC=f16x3 ROUND=${R}_f16x3 bash tools/gpu_round.sh pmc 2>&1 | grep -v rocprofv3 | tail -6
PREC=bf16x6 ROUND=${R}_bf16x6 bash tools/gpu_round.sh pmc 2>&1 | grep -v rocprofv3 | tail -4
bash tools/pmc_step.sh > gpurun_out/${R}_pmc_whole_step_hbm.txt 2>&1; tail -8 gpurun_out/${R}_pmc_whole_step_hbm.txt
KPAT="fused_tail_r1_kernel<stemseg::FusedTailR1Cfg<256" bash tools/gpu_r6_pmc_kernel.sh ${R}_fused_tail_r1 > /dev/null 2>&1; head -40 gpurun_out/${R}_fused_tail_r1_pmc_kernel.txt
for wl in davis ytvis; do
  timeout 600 python tools/soak_probe.py --workload $wl --lanes 3 --reps ${SOAK_REPS:-200} > gpurun_out/${R}_soak_${wl}.txt 2>&1; echo "soak $wl exit $?"; tail -1 gpurun_out/${R}_soak_${wl}.txt | cut -c1-200
done
