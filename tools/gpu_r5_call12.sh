#!/bin/bash
# A/B on one box: decoders clip by clip vs all clips of the step per launch (interleaved), then the launch count / tail of the batched build
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for r in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg --no-decoder-batch > gpurun_out/ab12_single_$r.json 2> gpurun_out/ab12_single_$r.log
  timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg > gpurun_out/ab12_batch_$r.json 2> gpurun_out/ab12_batch_$r.log
done
python - <<'PY'
import json
for n in ("single_1","batch_1","single_2","batch_2"):
    try:
        d=json.loads(open("gpurun_out/ab12_%s.json"%n).read().strip().splitlines()[-1])
        r=d["roofline"]; hk=r["hbm_kernels_eager"]["kernels"]
        print(n, d["value"], "3x3x3 ms/clip", r["conv_classes_eager"]["conv3x3x3"]["ms_per_clip"], "frac", r["frac"], "launches(3x3x3)", r["launches"], "tail us/clip", round(sum(k["us_per_clip"] for k in hk),1), d["config"]["determinism"]["mismatching"])
    except Exception as e:
        print(n, "ERR", e)
PY
