#!/bin/bash
# functional N = 2 / 3 runs of the sharded sequence path with the REAL kernels: ranks share the box's one GPU, gloo through the host
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python bench.py --sequence --frames 64 --steps 3 --warmup 1 > gpurun_out/seq64_n1.json 2>/dev/null
for n in 2 3; do
  STEMSEG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus $n --sequence --frames 64 --steps 3 --warmup 1 > gpurun_out/seq64_n${n}_gloo.json 2> gpurun_out/seq64_n${n}_gloo.err
  echo "n=$n exit $?"; tail -2 gpurun_out/seq64_n${n}_gloo.err | cut -c1-300
done
python - <<'PY'
import json
for f in ("seq64_n1", "seq64_n2_gloo", "seq64_n3_gloo"):
    try:
        j = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
        print(f, j["n_gpus"], j.get("ranks"), j["value"], j["result"]["label_checksum_crc32"], j["exchange"])
    except Exception as e:
        print(f, "FAILED", e)
PY
# weak-scaling default mode through the self-launcher, 2 ranks sharing the GPU (functional)
STEMSEG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/davis_n2_gloo.json 2> gpurun_out/davis_n2_gloo.err; echo "exit $?"; grep -o '"value": [0-9.]*\|"n_gpus": [0-9]*\|"ranks": [^]]*]' gpurun_out/davis_n2_gloo.json | head -4
