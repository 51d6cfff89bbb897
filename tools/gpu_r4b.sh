#!/bin/bash
# round 4, call B: lane-soak probe -- rate and location of the run-to-run differences under three lanes
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 150 > gpurun_out/soak_ytvis_f16x3.txt 2>&1; echo "exit $?"; tail -40 gpurun_out/soak_ytvis_f16x3.txt | cut -c1-200
timeout 600 python tools/soak_probe.py --workload davis --lanes 3 --reps 150 > gpurun_out/soak_davis_f16x3.txt 2>&1; echo "exit $?"; tail -30 gpurun_out/soak_davis_f16x3.txt | cut -c1-200
timeout 600 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 100 --precision bf16x6 > gpurun_out/soak_ytvis_bf16x6.txt 2>&1; echo "exit $?"; tail -20 gpurun_out/soak_ytvis_bf16x6.txt | cut -c1-200
