#!/bin/bash
# the two legs of the last evidence call that needed a fix: the kernel trace summary (tools/prof_steady.py delimits steps by the stem's first
# kernel, which changed name with the f16x3 stem) and the VALU-stem co-run probe (experiment library rebuilt for ABI 9)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=r05
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --no-sequence-leg --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_steady_state.txt 2>&1; head -4 gpurun_out/${R}_kernel_trace_steady_state.txt | cut -c1-200
rm -f gpurun_out/prof_graph/*.db gpucore.*
EXP=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_exp.so
STEMSEG_HIP_LIB=$EXP STEMSEG_STEM=valu timeout 400 python tools/graph_corun_probe.py --rounds 100 --aggressors k1,k1_bf16x6,k1_f32,k1_big,k1_wide,k2flat,k3,stream,stem --modes ee,gg,eg,ge > gpurun_out/${R}_graph_corun_valu_stem.txt 2>&1; echo "corun valu exit $?"; grep -E "total" gpurun_out/${R}_graph_corun_valu_stem.txt
