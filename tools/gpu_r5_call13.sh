#!/bin/bash
# (record: libstemseg_hip_ck16.so was the first build with the 16-channel chunks, against the then-product library)
# 16-channel chunks for the f16x3 1x3x3 tiles (no padded tap slot) against the product build: conv parity tests on the new library, then times
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
N=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_ck16.so
STEMSEG_HIP_LIB=$N timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_invariance.py -m gpu -q -x -k "conv or encoder" 2>&1 | tail -4
for r in 1 2; do
  ONLY=enc timeout 200 python tools/ab_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ck16_base_$r.txt
  STEMSEG_HIP_LIB=$N ONLY=enc timeout 200 python tools/ab_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ck16_new_$r.txt
done
paste -d'|' <(cut -c1-30,44-70 gpurun_out/ck16_base_1.txt) <(cut -c44-70 gpurun_out/ck16_new_1.txt) <(cut -c44-70 gpurun_out/ck16_base_2.txt) <(cut -c44-70 gpurun_out/ck16_new_2.txt) | grep "3x3\|#"
for r in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
  STEMSEG_HIP_LIB=$N timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
done
