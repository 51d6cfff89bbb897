#!/bin/bash
# round 5, GPU call 7: what would building the product WITHOUT packed-fp32 VALU instructions cost?  (interleaved A/B on one box) -- and are the results bit-identical?
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
L=$PWD/stem-seg_amd/stemseg_amd/lib
for rep in 1 2 3; do
  for tag in "" _nopk; do
    STEMSEG_HIP_LIB=$L/libstemseg_hip$tag.so timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-sequence-leg --alt-steps 10 > gpurun_out/ab7.log 2>&1
    echo "lib${tag:-_default} rep $rep: $(grep '^{' gpurun_out/ab7.log | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); c=j["roofline"]["conv_classes_eager"]; print(j["value"], {k:v["value"] for k,v in j["alt_precision"].items()}, {k:c[k]["ms_per_clip"] for k in c}, sum(k["us_per_clip"] for k in j["roofline"]["hbm_kernels_eager"]["kernels"]))' 2>&1 | tail -1)"
  done
done
python - <<'PY'
import os, sys, zlib, subprocess, json
# bit-identity of the two builds: label checksum + embedding bit sums of a sequence run
for tag in ("", "_nopk"):
    env = dict(os.environ, STEMSEG_HIP_LIB=os.path.join(os.getcwd(), "stem-seg_amd/stemseg_amd/lib/libstemseg_hip%s.so" % tag))
    r = subprocess.run([sys.executable, "bench.py", "--sequence", "--frames", "36", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    j = json.loads(line[-1]) if line else {}
    print("lib%s sequence36:" % (tag or "_default"), j.get("value"), j.get("result", {}).get("label_checksum_crc32"), j.get("result", {}).get("fg_points"))
PY
