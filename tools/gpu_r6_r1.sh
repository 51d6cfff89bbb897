#!/bin/bash
# R1 fused-tail form: parity tests, then encoder-alone A/B (kernel trace) of fuse_tail = 7|8 (16-column form) vs 7|16 (one wave per SIMD).
# Usage: bash tools/gpu_r6_r1.sh <tag>
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${1:-r06n}
if [[ -z "${SKIP_TESTS:-}" ]]; then
  timeout 900 python -m pytest tests/test_gpu_fused_tail.py -m gpu -q -x --timeout 600 -p no:cacheprovider -s > gpurun_out/${R}_tests.log 2>&1; echo "tests exit $?"
  grep -E "passed|failed|error|\[fused\]" gpurun_out/${R}_tests.log | tail -12 | cut -c1-220
  grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/${R}_tests.log | head -20 | cut -c1-250
fi
out=gpurun_out/${R}_r1_ab.txt; : > $out
for ft in ${FTS:-15 23 15 23}; do
  rm -rf gpurun_out/prof_ft
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_ft -o ft -- python $GRAFT_REPO_ROOT/tools/enc_bench.py --fuse-tail $ft ${ENC_ARGS:-}) > gpurun_out/prof_ft.log 2>&1
  echo "== fuse_tail $ft: $(grep '^encoder' gpurun_out/prof_ft.log)" >> $out
  db=$(find gpurun_out/prof_ft -name "*.db" | head -1)
  python tools/prof_steady.py $db 3 2>&1 | grep -E "steady|fused_tail|16, 4, 2, 1, 8, 16" | cut -c1-175 >> $out
done
rm -rf gpurun_out/prof_ft gpucore.*
cat $out
