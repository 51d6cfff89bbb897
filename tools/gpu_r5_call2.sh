#!/bin/bash
# round 5, GPU call 2: the tests that failed in call 1 (C-ABI conv3d did not forward the planning fields), the co-residency probe with
# more aggressor kinds, a conv sweep with the candidate tiles (tile_cfg 6 four-wave halves, 7 / 8 64-channel chunks).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r05b}
timeout 600 python -m pytest tests/test_gpu_invariance.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "conv_output or embeddings_are" > gpurun_out/${R}_gpu_tests_part.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|error" gpurun_out/${R}_gpu_tests_part.log | tail -2 | cut -c1-200; grep -E "^E  " gpurun_out/${R}_gpu_tests_part.log | head -8 | cut -c1-300
EXP=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_exp.so
STEMSEG_HIP_LIB=$EXP STEMSEG_STEM=valu timeout 500 python tools/graph_corun_probe.py --rounds 40 --aggressors k1,k1_f32,k1_bf16x6,k2,stream,stem,k3 --modes ee,gg > gpurun_out/${R}_graph_corun_valu_stem.txt 2>&1; echo "corun valu exit $?"; grep -E "victim|wrong words|total" gpurun_out/${R}_graph_corun_valu_stem.txt | cut -c1-260
STEMSEG_HIP_LIB=$EXP timeout 300 python tools/graph_corun_probe.py --rounds 40 --aggressors k1,k1_f32,k2,stem --modes ee,gg > gpurun_out/${R}_graph_corun_mfma_stem.txt 2>&1; echo "corun mfma exit $?"; grep -E "victim|total" gpurun_out/${R}_graph_corun_mfma_stem.txt | cut -c1-200
PREC=f16x3 SWEEP_T=32 REPS=12 timeout 600 python tools/conv_sweep.py > gpurun_out/${R}_conv_sweep_f16x3_T32.txt 2>&1; echo "sweep exit $?"; cut -c1-520 gpurun_out/${R}_conv_sweep_f16x3_T32.txt
