#!/bin/bash
# round 3, second GPU call: the bf16x6 convolution mode -- kernel tests, flows, bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bf16x6.py -q -s --timeout 300 -p no:cacheprovider -x > gpurun_out/x6_kernel_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/x6_kernel_tests.log; grep "bf16x6\]" gpurun_out/x6_kernel_tests.log | tail -40; tail -3 gpurun_out/x6_kernel_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s --timeout 300 -p no:cacheprovider -k "bf16x6" > gpurun_out/x6_flow_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/x6_flow_tests.log; grep -E "bf16x3-labels|passed|failed|Error" gpurun_out/x6_flow_tests.log | tail -12
timeout 600 python bench.py --precision bf16x6 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_x6.log 2>&1; echo "exit $?" >> gpurun_out/bench_x6.log; tail -c 1500 gpurun_out/bench_x6.log
