#!/bin/bash
# round 3, step j: kernel trace of the bench in the f16x3 mode (steady-state per-kernel table) + conv sweeps
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1 --no-graph) > gpurun_out/prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_steady.py $db 3 > gpurun_out/r03_f16x3_kernel_trace_steady_state.txt 2>&1; head -45 gpurun_out/r03_f16x3_kernel_trace_steady_state.txt | cut -c1-160
rm -rf gpurun_out/prof
PREC=f16x3 ONLY=enc SWEEP_T=32 timeout 300 python tools/conv_sweep.py > gpurun_out/r3j_sweep_enc_f16x3.txt 2>&1
PREC=f16x3 ONLY=dec timeout 300 python tools/conv_sweep.py > gpurun_out/r3j_sweep_dec_f16x3.txt 2>&1
