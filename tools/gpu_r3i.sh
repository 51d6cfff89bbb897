#!/bin/bash
# round 3, step i: f16x3 (pre-scaled low term) kernel tests, output distance to bf16x6 on a bench clip, bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16x6.py -q -x --timeout 600 -p no:cacheprovider -s > gpurun_out/r3i_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3i_tests.log
timeout 250 python tools/prec_diff.py 2>&1 | tail -4
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --precision f16x3 > gpurun_out/r3i_bench_f16x3.json 2> gpurun_out/r3i_bench_f16x3.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3i_bench_f16x3.json").read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["roofline"]["achieved"], d["roofline"]["frac"], d["config"]["last_clip"])
PY
