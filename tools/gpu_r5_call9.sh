#!/bin/bash
# A/B: the mixed-precision split (HEAD) against the previous build (libstemseg_hip_base.so): bit-identity per shape + time, then the step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
L=stem-seg_amd/stemseg_amd/lib
for r in 1 2; do
  STEMSEG_HIP_LIB=$PWD/$L/libstemseg_hip_base.so timeout 300 python tools/ab_conv.py > gpurun_out/ab9_base_$r.txt 2>&1
  timeout 300 python tools/ab_conv.py > gpurun_out/ab9_new_$r.txt 2>&1
done
for r in 1 2; do
  STEMSEG_HIP_LIB=$PWD/$L/libstemseg_hip_base.so timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg > gpurun_out/ab9_bench_base_$r.json 2> gpurun_out/ab9_bench_base_$r.log
  timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg > gpurun_out/ab9_bench_new_$r.json 2> gpurun_out/ab9_bench_new_$r.log
done
paste -d'|' <(cut -c1-70 gpurun_out/ab9_base_2.txt) <(cut -c31-70 gpurun_out/ab9_new_2.txt)
grep -h -o '"value": [0-9.]*' gpurun_out/ab9_bench_*.json | head -4
