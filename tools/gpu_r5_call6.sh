#!/bin/bash
# round 5, GPU call 6: the VALU stem compiled WITHOUT packed FMAs (encoder.hip with -fno-slp-vectorize: 448 v_fma_f32 instead of 224 v_pk_fma_f32;
# every other object identical to the experiment library) against the same f16x3 1x1 aggressor
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r05g}
L=$PWD/stem-seg_amd/stemseg_amd/lib
for tag in exp expnoslp; do
  echo "== library build: $tag"
  STEMSEG_HIP_LIB=$L/libstemseg_hip_$tag.so STEMSEG_STEM=valu timeout 300 python tools/graph_corun_probe.py --rounds 100 --aggressors k1 --modes ee,eg,ge,gg > gpurun_out/${R}_corun_$tag.txt 2>&1; echo "exit $?"; grep -E "^victim|wrong words|total" gpurun_out/${R}_corun_$tag.txt | cut -c1-300
done
