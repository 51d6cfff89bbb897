#!/bin/bash
# A/B of library builds (STEMSEG_BUILD_TAG variants under stemseg_amd/lib): interleaved bench runs on one box.
# usage (GPU box): bash tools/ab_lib.sh "<tag> <tag> ..." [bench args]      ("" = the default library)
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
tags="$1"; shift
L=stem-seg_amd/stemseg_amd/lib
for rep in 1 2; do
  for t in default $tags; do
    lib=$L/libstemseg_hip.so; [[ $t != default ]] && lib=$L/libstemseg_hip_$t.so
    STEMSEG_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], 'mismatch', d['config']['determinism']['mismatching'], 'k3 frac', d['roofline']['frac'])"
  done
done
