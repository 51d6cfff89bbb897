#!/bin/bash
# round 3, step h: f16x3 kernel tests + conv sweeps (bf16x6 vs f16x3) + bench in both modes
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16x6.py -q -x --timeout 600 -p no:cacheprovider -s > gpurun_out/r3h_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3h_tests.log
for P in bf16x6 f16x3; do
  PREC=$P ONLY=enc SWEEP_T=32 timeout 300 python tools/conv_sweep.py > gpurun_out/r3h_sweep_enc_$P.txt 2>&1
  PREC=$P ONLY=dec timeout 300 python tools/conv_sweep.py > gpurun_out/r3h_sweep_dec_$P.txt 2>&1
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --precision $P > gpurun_out/r3h_bench_$P.json 2> gpurun_out/r3h_bench_$P.err; tail -c 600 gpurun_out/r3h_bench_$P.json
done
