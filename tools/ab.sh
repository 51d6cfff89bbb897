#!/bin/bash
# A/B of library switches on the GPU box: one short bench per setting, value + conv-class breakdown per line.
#   gpurun -- 'bash tools/ab.sh "STEMSEG_GLDS=0" "STEMSEG_GLDS=1" ...'      (each argument: space-separated VAR=VALUE list)
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
i=0
for setting in "$@"; do
  i=$((i+1))
  log=gpurun_out/ab_$i.log
  env $setting timeout 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > $log 2>&1
  python - "$setting" $log <<'PY'
import json, sys
setting, log = sys.argv[1], sys.argv[2]
try:
    j = json.loads(open(log).read().strip().splitlines()[-1])
    c = j["roofline"]["conv_classes_eager"]
    print("%-40s %7.2f clips/s | 3x3x3 %.3f ms %5.1f TF | 1x3x3 %.3f ms %5.1f TF | 1x1x1 %.3f ms %5.1f TF | whole %.3f" % (
        setting, j["value"], c["conv3x3x3"]["ms_per_clip"], c["conv3x3x3"]["tflops"], c["conv1x3x3"]["ms_per_clip"], c["conv1x3x3"]["tflops"],
        c["conv1x1x1"]["ms_per_clip"], c["conv1x1x1"]["tflops"], j["roofline"]["whole_step"]["frac"]))
except Exception as e:
    print("%-40s FAILED (%r): %s" % (setting, e, open(log).read()[-400:].replace("\n", " | ")))
PY
done
