#!/bin/bash
# round 3, first GPU call: GPU tests, smoke, bench on the three workloads, sequence mode (both partitionings)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider -x > gpurun_out/tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/tests.log; tail -4 gpurun_out/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_davis.log 2>&1; echo "exit $?" >> gpurun_out/bench_davis.log; tail -c 600 gpurun_out/bench_davis.log
for wl in ytvis kitti; do
  timeout 900 python bench.py --workload $wl --steps 8 --warmup 2 > gpurun_out/bench_$wl.log 2>&1; echo "exit $?" >> gpurun_out/bench_$wl.log; tail -c 400 gpurun_out/bench_$wl.log
done
for part in clips replicated; do
  timeout 600 python bench.py --sequence --frames 64 --partition $part --steps 5 --warmup 2 > gpurun_out/bench_seq64_$part.log 2>&1; echo "exit $?" >> gpurun_out/bench_seq64_$part.log; tail -c 500 gpurun_out/bench_seq64_$part.log
done
timeout 600 python bench.py --sequence --frames 36 --steps 5 --warmup 2 > gpurun_out/bench_seq36_clips.log 2>&1; echo "exit $?" >> gpurun_out/bench_seq36_clips.log; tail -c 300 gpurun_out/bench_seq36_clips.log
