#!/bin/bash
# (record of a removed experiment: libstemseg_hip_prev.so was the previous commit's build)
# FPN top-down add fused into the lateral conv's epilogue (HEAD) against the previous build: encoder parity, kernel times, the step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
P=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_prev.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_invariance.py -m gpu -q -x -k "encoder or fpn or embeddings or conv" 2>&1 | tail -3
STEMSEG_HIP_LIB=$P timeout 200 python tools/ab_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/up_prev.txt
timeout 200 python tools/ab_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/up_new.txt
paste -d'|' <(cut -c1-70 gpurun_out/up_prev.txt) <(cut -c31-70 gpurun_out/up_new.txt)
for r in 1 2; do
  for L in $P ""; do
    STEMSEG_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('${L:-HEAD}'[-12:], d['value'], {k:v['ms_per_clip'] for k,v in r['conv_classes_eager'].items()}, r['frac'], round(sum(k['us_per_clip'] for k in r['hbm_kernels_eager']['kernels']),1))"
  done
done
