#!/bin/bash
# round 3, step m: lane determinism x8 per precision, full GPU suite (f16x3 default), bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for p in f16x3 bf16x6; do echo "== lanes $p"; for i in 1 2 3 4 5 6 7 8; do STEMSEG_PRECISION=$p timeout 200 python -m pytest tests/test_gpu_parity.py -q -k "test_step_batch_shares_the_encoder_pass or test_embed_many_batches_and_lanes" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | sed "s/, 185 deselected in//" | tr "\n" " "; done; echo; done
timeout 1400 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r3m_tests_f16x3.log 2>&1; tail -3 gpurun_out/r3m_tests_f16x3.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r3m_bench_f16x3.json 2> gpurun_out/r3m_bench_f16x3.err; grep -o '"value": [0-9.]*' gpurun_out/r3m_bench_f16x3.json | head -1
