#!/bin/bash
# round 5, GPU call 5: which property of the f16x3 1x1 kernel disturbs the VALU stem?  A/B builds of the experiment library.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r05f}
L=$PWD/stem-seg_amd/stemseg_amd/lib
for tag in exp expw3 expns expwm0; do
  echo "== library build: $tag"
  STEMSEG_HIP_LIB=$L/libstemseg_hip_$tag.so STEMSEG_STEM=valu timeout 300 python tools/graph_corun_probe.py --rounds 60 --aggressors k1,k1_big,k1_wide --modes ee > gpurun_out/${R}_corun_$tag.txt 2>&1; echo "exit $?"; grep -E "^victim|wrong words|total" gpurun_out/${R}_corun_$tag.txt | cut -c1-300
done
