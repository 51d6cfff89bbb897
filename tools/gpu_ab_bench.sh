#!/bin/bash
# Interleaved A/B of the default bench line (graph replay, 3 lanes) over bench.py argument sets and / or libraries.
# Usage: ARMS="name1|args1;name2|args2" [LIB_name=<tag>] REPS=3 bash tools/gpu_ab_bench.sh <tag>
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${1:-r06q}; out=gpurun_out/${R}_ab_bench.txt; : > $out
IFS=';' read -ra arms <<< "${ARMS}"
for rep in $(seq 1 ${REPS:-3}); do
  for arm in "${arms[@]}"; do
    name=${arm%%|*}; args=${arm#*|}
    libvar="LIB_$name"; lib=""; [[ -n "${!libvar:-}" ]] && lib=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_${!libvar}.so
    line=$(STEMSEG_HIP_LIB=$lib timeout 400 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg --steps ${STEPS:-60} $args 2>/dev/null | grep "^{" | tail -1)
    python - "$name" "$rep" <<PY >> $out
import json, sys
try:
    j = json.loads('''$line''')
    c = j["roofline"]["conv_classes_eager"]
    print("%-10s rep %s: %.2f clips/s  %.3f ms/step  3x3x3 %.3f 1x3x3 %.3f 1x1 %.3f ms/clip  bitsum %s  sclk %.3f" % (sys.argv[1], sys.argv[2], j["value"], j["ms_per_step"],
          c["conv3x3x3"]["ms_per_clip"], c["conv1x3x3"]["ms_per_clip"], c["conv1x1x1"]["ms_per_clip"], j["config"].get("first_clip_bitsum"), j["roofline"].get("sclk_ghz_timed_region") or 0))
except Exception as e:
    print(sys.argv[1], "rep", sys.argv[2], "unreadable:", e)
PY
  done
done
cat $out
