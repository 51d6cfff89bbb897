#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'k3', r['conv_classes_eager']['conv3x3x3']['ms_per_clip'], 'k2', r['conv_classes_eager']['conv1x3x3']['ms_per_clip'], 'k1', r['conv_classes_eager']['conv1x1x1']['ms_per_clip'], 'mismatch', d['config']['determinism']['mismatching'])"; }
for m in 1 2 3 0 1 2; do echo flat mode $m; STEMSEG_X6_FLAT=$m b; done
