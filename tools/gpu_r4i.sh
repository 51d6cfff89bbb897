#!/bin/bash
# round 4, call I: split-K threshold A/B
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(sys.argv[1:], d['value'], 'k3', r['conv_classes_eager']['conv3x3x3']['ms_per_clip'], 'k2', r['conv_classes_eager']['conv1x3x3']['ms_per_clip'], 'k1', r['conv_classes_eager']['conv1x1x1']['ms_per_clip'], 'mismatch', d['config']['determinism']['mismatching'])" "$@"; }
echo default; b; b
echo all640; STEMSEG_SPLITK_WGS=640 b
echo all320; STEMSEG_SPLITK_WGS=320 b
echo all1024; STEMSEG_SPLITK_WGS=1024 b
echo all256; STEMSEG_SPLITK_WGS=256 b
