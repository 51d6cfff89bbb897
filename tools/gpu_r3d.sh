#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
T="tests/test_gpu_parity.py::test_step_batch_shares_the_encoder_pass tests/test_gpu_parity.py::test_embed_many_batches_and_lanes_match_per_clip_embedding"
for m in 0 1 2 4 8 15; do
  STEMSEG_X6_TILES=$m STEMSEG_PRECISION=bf16x6 timeout 600 python -m pytest $T -q --timeout 300 -p no:cacheprovider > gpurun_out/dbg_x6_$m.log 2>&1
  echo "tiles mask $m: $(tail -1 gpurun_out/dbg_x6_$m.log)"
done
REPS=10 PREC=bf16x6 ONLY=dec timeout 300 python tools/conv_sweep.py > gpurun_out/sweep_x6_dec.log 2>&1; cut -c1-200 gpurun_out/sweep_x6_dec.log
timeout 300 python bench.py --precision bf16x6 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_x6.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/bench_x6.log | head -1
python - <<'PY'
import json
for l in open('gpurun_out/bench_x6.log'):
    if l.startswith('{'):
        j=json.loads(l); print({k:(v['ms_per_clip'],v['tflops']) for k,v in j['roofline']['conv_classes_eager'].items()})
PY
