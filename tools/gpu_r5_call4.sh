#!/bin/bash
# round 5, GPU call 4: co-residency probe -- the VALU stem against 3x3 aggressors (probe fixed), and the PRODUCT's own VALU-bound kernels as victims
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r05d}
EXP=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_exp.so
STEMSEG_HIP_LIB=$EXP STEMSEG_STEM=valu timeout 500 python tools/graph_corun_probe.py --rounds 100 --aggressors k1,k1_bf16x6,k1_f32,k2flat,stream,stem,k3 --modes ee,gg,eg,ge > gpurun_out/${R}_graph_corun_valu_stem.txt 2>&1; echo "corun valu exit $?"; grep -E "^victim|wrong words|aggressor output|total" gpurun_out/${R}_graph_corun_valu_stem.txt | cut -c1-330
timeout 500 python tools/graph_corun_probe.py --rounds 40 --victims stem,heads,upsample,gn --aggressors k1,k2flat,k3 --modes ee,gg,eg > gpurun_out/${R}_graph_corun_product_kernels.txt 2>&1; echo "corun product exit $?"; grep -E "^victim|aggressor output|total" gpurun_out/${R}_graph_corun_product_kernels.txt | cut -c1-330
