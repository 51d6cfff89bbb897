#!/bin/bash
# steady-state kernel trace of one bench workload (single lane, graph replay).  Usage: bash tools/gpu_trace_workload.sh <tag> <workload> [bench args]
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$1; W=$2; shift 2
rm -rf gpurun_out/prof_w
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_w -o $R -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --no-sequence-leg --lanes 1 "$@") > gpurun_out/prof_w.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_w -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_${W}.txt 2>&1; head -45 gpurun_out/${R}_kernel_trace_${W}.txt | cut -c1-160
rm -rf gpurun_out/prof_w gpucore.*
