#!/bin/bash
# round 3: f16x3 with all MFMA operands from LDS (three staged weight planes): kernel tests, determinism monitor at 3 lanes, lane tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 200 python -m pytest tests/test_gpu_bf16x6.py -q -x --timeout 180 -p no:cacheprovider 2>&1 | tail -1
run() { timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], d['value'], d['config']['determinism']['mismatching'], d['config']['determinism']['clip_results_checked'])" "$@"; }
run --lanes 3 --steps 40
run --lanes 3 --steps 40
timeout 100 python -m pytest tests/test_gpu_parity.py -q -k "test_step_batch_shares_the_encoder_pass or test_embed_many_batches_and_lanes" -p no:cacheprovider 2>&1 | tail -1
