#!/bin/bash
# final-build soak: 3 lanes x 300 rounds, davis f16x3 and ytvis bf16x6 (every lane-owned buffer vs the lone replay)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python tools/soak_probe.py --workload davis --lanes 3 --reps 300 --precision f16x3 > gpurun_out/soak6_davis_f16x3_final.txt 2>&1; tail -3 gpurun_out/soak6_davis_f16x3_final.txt | cut -c1-200
timeout 600 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 300 --precision f16x3 > gpurun_out/soak6_ytvis_f16x3_final.txt 2>&1; tail -3 gpurun_out/soak6_ytvis_f16x3_final.txt | cut -c1-200
timeout 600 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 300 --precision bf16x6 > gpurun_out/soak6_ytvis_bf16x6_final.txt 2>&1; tail -3 gpurun_out/soak6_ytvis_bf16x6_final.txt | cut -c1-200
