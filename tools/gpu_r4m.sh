#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'k3', r['conv_classes_eager']['conv3x3x3']['ms_per_clip'], 'k2', r['conv_classes_eager']['conv1x3x3']['ms_per_clip'], 'k1', r['conv_classes_eager']['conv1x1x1']['ms_per_clip'], 'mismatch', d['config']['determinism']['mismatching'])"; }
echo flat on; b; b
echo flat off; STEMSEG_X6_FLAT=0 b; STEMSEG_X6_FLAT=0 b
R=r04m
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/${R}_kernel_trace_steady_state.txt 2>&1; head -24 gpurun_out/${R}_kernel_trace_steady_state.txt | cut -c1-175
rm -f gpurun_out/prof_graph/*.db gpucore.*
