#!/bin/bash
# (the probe libraries: STEMSEG_BUILD_DEFINES="-DSS_EXPERIMENTS -DSS_PROBE=<n>" STEMSEG_BUILD_TAG=p<n> python stem-seg_amd/build.py, n = 1..5)
# timing probes of the split-staged chunk loop (SS_PROBE builds: results wrong by construction, times only)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
L=$PWD/stem-seg_amd/stemseg_amd/lib
for t in 32 8; do
  ONLY=enc SWEEP_T=$t timeout 200 python tools/ab_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/probe_p0_T$t.txt
  for n in 1 2 3 4 5; do
    STEMSEG_HIP_LIB=$L/libstemseg_hip_p$n.so ONLY=enc SWEEP_T=$t timeout 200 python tools/ab_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/probe_p${n}_T$t.txt
  done
  echo "== T=$t: us per launch: product | 1 no loads | 2 no split/LDS writes | 3 no MFMA | 4 no fragment reads | 5 no barriers"
  paste -d'|' <(cut -c1-30,44-55 gpurun_out/probe_p0_T$t.txt) <(cut -c44-55 gpurun_out/probe_p1_T$t.txt) <(cut -c44-55 gpurun_out/probe_p2_T$t.txt) <(cut -c44-55 gpurun_out/probe_p3_T$t.txt) <(cut -c44-55 gpurun_out/probe_p4_T$t.txt) <(cut -c44-55 gpurun_out/probe_p5_T$t.txt)
done
