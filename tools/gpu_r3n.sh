#!/bin/bash
# round 3 final artifacts on the shipped code (f16x3 default): bench lines of every workload, kernel trace of the bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 420 python bench.py > gpurun_out/r03_bench_davis.json 2> gpurun_out/r03_bench_davis.err; grep -o '"value": [0-9.]*' gpurun_out/r03_bench_davis.json | head -1
rm -rf gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1 --no-graph) > gpurun_out/prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_steady.py $db 3 > gpurun_out/r03_kernel_trace_steady_state.txt 2>&1; head -6 gpurun_out/r03_kernel_trace_steady_state.txt | cut -c1-150
find gpurun_out/prof -name "*kernel_stats*" | head -1 | while read f; do head -25 "$f" > gpurun_out/r03_rocprof_kernel_stats_head.csv; done
rm -rf gpurun_out/prof
for wl in ytvis kitti; do timeout 300 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_$wl.json 2>/dev/null; echo -n "$wl "; grep -o '"value": [0-9.]*' gpurun_out/r03_bench_$wl.json | head -1; done
for f in 64 36; do timeout 300 python bench.py --sequence --frames $f --steps 3 --warmup 1 > gpurun_out/r03_bench_seq$f.json 2>/dev/null; echo -n "seq$f "; grep -o '"value": [0-9.]*' gpurun_out/r03_bench_seq$f.json | head -1; done
