#!/bin/bash
# round 3 closing run: the bench line with the determinism monitor, smoke, lane tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 420 python bench.py > gpurun_out/r03_bench_davis.json 2> gpurun_out/r03_bench_davis.err; tail -3 gpurun_out/r03_bench_davis.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench_davis.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["achieved_vs_bf16x6_roof"], d["config"]["determinism"], d["cpu_baseline"]["parity_vs_hip_path"]["labels_identical_fraction_on_common_fg"])
PY
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -2
for i in 1 2 3; do timeout 200 python -m pytest tests/test_gpu_parity.py -q -k "test_step_batch_shares_the_encoder_pass or test_embed_many_batches_and_lanes" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tr "\n" " "; done; echo
