#!/bin/bash
# round 3, step l: full GPU suite with the f16x3 default, then with bf16x6 as default; bench lines
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1400 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r3l_tests_f16x3.log 2>&1; tail -4 gpurun_out/r3l_tests_f16x3.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r3l_bench_f16x3.json 2> gpurun_out/r3l_bench_f16x3.err; grep -o '"value": [0-9.]*' gpurun_out/r3l_bench_f16x3.json | head -1
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --precision bf16x6 > gpurun_out/r3l_bench_bf16x6.json 2> gpurun_out/r3l_bench_bf16x6.err; grep -o '"value": [0-9.]*' gpurun_out/r3l_bench_bf16x6.json | head -1
STEMSEG_PRECISION=bf16x6 timeout 1400 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r3l_tests_bf16x6.log 2>&1; tail -4 gpurun_out/r3l_tests_bf16x6.log
