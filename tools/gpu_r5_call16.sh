#!/bin/bash
# the f16x3 stem (space-to-depth + 4x4 conv) against the exact fp32-MFMA stem: unit + encoder parity, then the step, interleaved
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_invariance.py -m gpu -q -x -k "stem or encoder or embeddings or end_to_end" -s 2>&1 | grep -E "stem s2d|passed|failed|Error|error" | tail -12
for r in 1 2; do
  for f in "--fp32-stem" ""; do
    timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=[x for x in r['hbm_kernels_eager']['kernels'] if 'stem' in x['kernel']]; print('${f:-s2d-stem}', d['value'], k[0]['us_per_clip'] if k else None, round(sum(x['us_per_clip'] for x in r['hbm_kernels_eager']['kernels']),1), d['config']['determinism']['mismatching'])"
  done
done
