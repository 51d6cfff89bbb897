#!/bin/bash
# round 5, GPU call 3: co-residency probe with lone controls (both stem forms), the full GPU suite on the cleaned library, lane soaks.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r05c}
EXP=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_exp.so
STEMSEG_HIP_LIB=$EXP STEMSEG_STEM=valu timeout 500 python tools/graph_corun_probe.py --rounds 40 --aggressors k1,k1_f32,k1_bf16x6,k2,k2flat,stem --modes ee,gg > gpurun_out/${R}_graph_corun_valu_stem.txt 2>&1; echo "corun valu exit $?"; grep -E "victim|wrong words|aggressor output|total" gpurun_out/${R}_graph_corun_valu_stem.txt | cut -c1-330
STEMSEG_HIP_LIB=$EXP timeout 300 python tools/graph_corun_probe.py --rounds 40 --aggressors k1,k2,k2flat --modes ee,gg > gpurun_out/${R}_graph_corun_mfma_stem.txt 2>&1; echo "corun mfma exit $?"; grep -E "victim|aggressor output|total" gpurun_out/${R}_graph_corun_mfma_stem.txt | cut -c1-330
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 -p no:cacheprovider --durations=10 > gpurun_out/${R}_gpu_tests.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|error" gpurun_out/${R}_gpu_tests.log | tail -3 | cut -c1-300
grep -E "^FAILED|^ERROR" gpurun_out/${R}_gpu_tests.log | head -20 | cut -c1-250
grep -E "^\[soak\]|^\[invariance\]|^\[nccl\]" gpurun_out/${R}_gpu_tests.log | cut -c1-300
for wl in davis ytvis; do
  timeout 600 python tools/soak_probe.py --workload $wl --lanes 3 --reps 300 --small > gpurun_out/${R}_soak_${wl}.txt 2>&1; echo "soak $wl exit $?"; tail -3 gpurun_out/${R}_soak_${wl}.txt | cut -c1-300
done
