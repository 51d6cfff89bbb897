#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
T="tests/test_gpu_parity.py::test_step_batch_shares_the_encoder_pass tests/test_gpu_parity.py::test_embed_many_batches_and_lanes_match_per_clip_embedding tests/test_gpu_parity.py::test_graphed_step_equals_eager"
for i in 1 2 3 4 5 6; do
  STEMSEG_PRECISION=bf16x6 timeout 600 python -m pytest $T -q --timeout 300 -p no:cacheprovider > gpurun_out/stress_x6_$i.log 2>&1
  echo "stress $i: $(tail -1 gpurun_out/stress_x6_$i.log)"
done
for i in 1 2 3; do
  STEMSEG_X6_PLANNER=1 STEMSEG_PRECISION=bf16x6 timeout 600 python -m pytest $T -q --timeout 300 -p no:cacheprovider > gpurun_out/stress_x6_plan_$i.log 2>&1
  echo "stress planner $i: $(tail -1 gpurun_out/stress_x6_plan_$i.log)"
done
STEMSEG_PRECISION=bf16x6 timeout 1400 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/tests_default_x6.log 2>&1; tail -4 gpurun_out/tests_default_x6.log
timeout 400 python bench.py --precision bf16x6 --steps 100 --warmup 3 --no-cpu-baseline > gpurun_out/bench_x6_soak.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/bench_x6_soak.log | head -1
timeout 300 python bench.py --precision bf16x6 --sequence --frames 64 --steps 5 --warmup 2 > gpurun_out/bench_seq64_x6.log 2>&1; grep -o '"value": [0-9.]*\|crc32": [0-9]*' gpurun_out/bench_seq64_x6.log | head -3
timeout 300 python bench.py --precision f32 --sequence --frames 64 --steps 3 --warmup 1 > gpurun_out/bench_seq64_f32.log 2>&1; grep -o '"value": [0-9.]*\|crc32": [0-9]*' gpurun_out/bench_seq64_f32.log | head -3
