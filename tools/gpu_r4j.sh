#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=r04j
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_steady_state.txt 2>&1; head -48 gpurun_out/${R}_kernel_trace_steady_state.txt | cut -c1-175
rm -f gpurun_out/prof_graph/*.db gpucore.*
