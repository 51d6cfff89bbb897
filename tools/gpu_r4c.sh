#!/bin/bash
# round 4, call C: stem kernel without AGPR parking -- encoder goldens, then the lane-soak probe in both split modes
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "encoder" -p no:cacheprovider 2>&1 | tail -2
timeout 600 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 300 --precision bf16x6 > gpurun_out/soak2_ytvis_bf16x6.txt 2>&1; echo "exit $?"; tail -25 gpurun_out/soak2_ytvis_bf16x6.txt | cut -c1-250
timeout 600 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 300 > gpurun_out/soak2_ytvis_f16x3.txt 2>&1; echo "exit $?"; tail -25 gpurun_out/soak2_ytvis_f16x3.txt | cut -c1-250
timeout 600 python tools/soak_probe.py --workload davis --lanes 3 --reps 300 > gpurun_out/soak2_davis_f16x3.txt 2>&1; echo "exit $?"; tail -25 gpurun_out/soak2_davis_f16x3.txt | cut -c1-250
