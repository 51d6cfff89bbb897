#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_bf16x6.py -q -x -p no:cacheprovider -k "flat_split" 2>&1 | tail -1
b() { timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], 'mismatch', d['config']['determinism']['mismatching'])" "$@"; }
b; b --lanes 2; b --lanes 4; b --clips-per-step 6; b --clips-per-step 8 --lanes 2; b --clips-per-step 2 --lanes 4
