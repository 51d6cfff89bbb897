#!/bin/bash
# round 3: artifacts on the current code -- PMC (bf16x6 block_4x), kernel trace of the bench, bench lines, full GPU suite
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/device.txt
export PREC=bf16x6
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_conv.py 3 0) > gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag exit $?"
done
python tools/pmc_summary.py r03_bf16x6 3 | tail -14
rm -rf gpurun_out/pmc_*/ gpucore.*
unset PREC
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1 --no-graph) > gpurun_out/prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_steady.py $db 3 > gpurun_out/r03_kernel_trace_steady_state.txt 2>&1; head -12 gpurun_out/r03_kernel_trace_steady_state.txt | cut -c1-150
find gpurun_out/prof -name "*kernel_stats*" | head -2 | while read f; do head -25 "$f" > gpurun_out/r03_rocprof_kernel_stats_head.csv; done
rm -rf gpurun_out/prof
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench_davis.json 2> gpurun_out/r03_bench_davis.err; grep -o '"value": [0-9.]*' gpurun_out/r03_bench_davis.json | head -1
timeout 600 python bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_davis_f32mfma.json 2>/dev/null; grep -o '"value": [0-9.]*' gpurun_out/r03_bench_davis_f32mfma.json | head -1
for wl in ytvis kitti; do timeout 600 python bench.py --workload $wl --steps 8 --warmup 2 > gpurun_out/r03_bench_$wl.json 2>/dev/null; grep -o '"value": [0-9.]*' gpurun_out/r03_bench_$wl.json | head -1; done
for f in 64 36; do timeout 600 python bench.py --sequence --frames $f --steps 5 --warmup 2 > gpurun_out/r03_bench_seq$f.json 2>/dev/null; grep -o '"value": [0-9.]*' gpurun_out/r03_bench_seq$f.json | head -1; done
timeout 1400 python -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider > gpurun_out/r03_gpu_tests.log 2>&1; tail -3 gpurun_out/r03_gpu_tests.log
