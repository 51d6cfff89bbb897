#!/bin/bash
# round 4, call D: stand-alone co-residency probe; split-mode kernel tests (per-channel scale, NaN-keeping ReLU); soak with word-level diagnostics
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 tools/microbench/bin/valu_corun_probe 300 > gpurun_out/valu_corun_probe.txt 2>&1; echo "probe exit $?"; cat gpurun_out/valu_corun_probe.txt | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_bf16x6.py -q -x -s --timeout 300 -p no:cacheprovider 2>&1 | grep -E "dynamic range|checkpoint-like|passed|failed|Error|error" | cut -c1-220 | tail -30
timeout 600 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 60 --precision bf16x6 --max-reports 3 > gpurun_out/soak3_ytvis_bf16x6.txt 2>&1; echo "exit $?"; grep -n "S0\|X1 \|neighbourhood\|got \|RESULT" gpurun_out/soak3_ytvis_bf16x6.txt | cut -c1-1500 | head -40
