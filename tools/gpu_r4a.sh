#!/bin/bash
# round 4, call A: evidence on the SHIPPED f16x3 kernel (HEAD of round 3): kernel trace, PMC passes on the block_4x conv,
# whole-step HBM PMC, and the workload lines that round 3 left on the withdrawn build.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=r04a
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/device.txt
# 1. kernel trace (graph replay, one lane), steady state
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_steady_state.txt 2>&1; head -30 gpurun_out/${R}_kernel_trace_steady_state.txt | cut -c1-170
rm -f gpurun_out/prof_graph/*.db gpucore.*
# 2. PMC passes on the dominant conv (f16x3)
PREC=f16x3 ROUND=$R bash tools/gpu_round.sh pmc 2>&1 | grep -v rocprofv3 | tail -16
# 3. whole-step HBM traffic (f16x3, eager single lane)
bash tools/pmc_step.sh > gpurun_out/${R}_pmc_whole_step_hbm.txt 2>&1; tail -25 gpurun_out/${R}_pmc_whole_step_hbm.txt
# 4. workload lines
for wl in ytvis kitti; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_$wl.log 2>&1; echo "$wl exit $?"; tail -1 gpurun_out/bench_$wl.log > gpurun_out/${R}_bench_$wl.json; cut -c1-150 gpurun_out/${R}_bench_$wl.json
done
timeout 300 python bench.py --sequence --frames 64 --steps 3 --warmup 1 > gpurun_out/bench_seq64.log 2>&1; echo "seq64 exit $?"; tail -1 gpurun_out/bench_seq64.log > gpurun_out/${R}_bench_seq64.json; cut -c1-150 gpurun_out/${R}_bench_seq64.json
timeout 300 python bench.py --sequence --frames 36 --steps 3 --warmup 1 > gpurun_out/bench_seq36.log 2>&1; echo "seq36 exit $?"; tail -1 gpurun_out/bench_seq36.log > gpurun_out/${R}_bench_seq36.json; cut -c1-150 gpurun_out/${R}_bench_seq36.json
timeout 300 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/bench_davis.log 2>&1; echo "davis exit $?"; tail -1 gpurun_out/bench_davis.log > gpurun_out/${R}_bench_davis.json; cut -c1-150 gpurun_out/${R}_bench_davis.json
