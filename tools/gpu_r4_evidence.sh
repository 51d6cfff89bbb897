#!/bin/bash
# round 4 evidence set on the shipped build (one gpurun call): GPU suite log, default bench line (alt-precision legs + CPU baseline),
# workload lines, kernel trace, PMC passes on the block_4x conv (f16x3 and bf16x6), whole-step HBM PMC, co-residency probes.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r04}
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/${R}_device.txt; nproc >> gpurun_out/${R}_device.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/${R}_device.txt
if [[ -z "${SKIP_TESTS:-}" ]]; then
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider --durations=8 > gpurun_out/${R}_gpu_tests.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|error" gpurun_out/${R}_gpu_tests.log | tail -2 | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log | cut -c1-200
fi
timeout 600 python bench.py > gpurun_out/bench_davis.log 2>&1; echo "bench exit $?"; grep "^{" gpurun_out/bench_davis.log | tail -1 > gpurun_out/${R}_bench_davis.json; cut -c1-160 gpurun_out/${R}_bench_davis.json
for wl in ytvis kitti; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_$wl.log 2>&1; echo "$wl exit $?"; grep "^{" gpurun_out/bench_$wl.log | tail -1 > gpurun_out/${R}_bench_$wl.json; cut -c1-150 gpurun_out/${R}_bench_$wl.json
done
for fr in 64 36; do
  timeout 300 python bench.py --sequence --frames $fr --steps 4 --warmup 1 > gpurun_out/bench_seq$fr.log 2>&1; echo "seq$fr exit $?"; grep "^{" gpurun_out/bench_seq$fr.log | tail -1 > gpurun_out/${R}_bench_seq$fr.json; cut -c1-150 gpurun_out/${R}_bench_seq$fr.json
done
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_steady_state.txt 2>&1; head -6 gpurun_out/${R}_kernel_trace_steady_state.txt | cut -c1-170
rm -f gpurun_out/prof_graph/*.db gpucore.*
PREC=f16x3 ROUND=${R}_f16x3 bash tools/gpu_round.sh pmc 2>&1 | grep -v rocprofv3 | tail -6
PREC=bf16x6 ROUND=${R}_bf16x6 bash tools/gpu_round.sh pmc 2>&1 | grep -v rocprofv3 | tail -4
bash tools/pmc_step.sh > gpurun_out/${R}_pmc_whole_step_hbm.txt 2>&1; tail -12 gpurun_out/${R}_pmc_whole_step_hbm.txt
if [[ -z "${SKIP_PROBES:-}" ]]; then
STEMSEG_STEM=valu timeout 600 python tools/stem_corun_probe.py --reps 100 --victims stem --aggressors none,k3_f16x3,k3_bf16x6,k2_f16x3,k1_f16x3_expand > gpurun_out/${R}_stem_corun_probe.txt 2>&1; grep victim gpurun_out/${R}_stem_corun_probe.txt | cut -c1-200
timeout 300 tools/microbench/bin/valu_corun_probe 100 > gpurun_out/${R}_valu_corun_probe.txt 2>&1; grep -c "0 of 100 launches wrong" gpurun_out/${R}_valu_corun_probe.txt
fi
