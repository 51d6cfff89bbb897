#!/bin/bash
# Round-end measurement set on the GPU box (one gpurun call): tests, smoke, bench (+ profile), sequence modes incl. the 2-rank
# functional check on one GPU (gloo), lane A/B, bf16x3, PMC passes on the dominant conv.
bash tools/gpu_call.sh
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python bench.py --sequence --frames 36 --steps 3 --warmup 1 > gpurun_out/bench_seq36.log 2>&1; echo "seq36 exit $?"; tail -1 gpurun_out/bench_seq36.log > gpurun_out/bench_seq36.json; cut -c1-120 gpurun_out/bench_seq36.json
STEMSEG_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --sequence --steps 2 --warmup 1 > gpurun_out/bench_seq64_2ranks_gloo.log 2>&1; echo "seq64 2 ranks (gloo, one GPU) exit $?"; tail -1 gpurun_out/bench_seq64_2ranks_gloo.log > gpurun_out/bench_seq64_2ranks_gloo.json
python - <<'PY'
import json
a = json.load(open("gpurun_out/bench_seq64.json")); b = json.load(open("gpurun_out/bench_seq64_2ranks_gloo.json"))
print("checksum world 1 %d, world 2 %d -> %s; all-gather %s bytes, %.2f ms (gloo through the host)" % (
    a["result"]["label_checksum_crc32"], b["result"]["label_checksum_crc32"], "IDENTICAL" if a["result"] == b["result"] else "DIFFERENT",
    b["exchange"]["bytes_received_per_rank"], b["exchange"]["ms_median"]))
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1; echo "torchrun N=1 (RCCL init) exit $?"; tail -1 gpurun_out/bench_torchrun1.log | cut -c1-110
BENCH_ARGS="--lanes 2" STEPS=15 bash tools/ab.sh "LANES=2"
BENCH_ARGS="--lanes 4" STEPS=15 bash tools/ab.sh "LANES=4"
timeout 300 python bench.py --precision bf16x3 --no-cpu-baseline > gpurun_out/bench_bf16x3.log 2>&1; tail -1 gpurun_out/bench_bf16x3.log > gpurun_out/bench_bf16x3.json; cut -c1-110 gpurun_out/bench_bf16x3.json
bash tools/gpu_round.sh pmc 2>&1 | grep -v rocprofv3 | tail -14
