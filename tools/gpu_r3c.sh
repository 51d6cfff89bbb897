#!/bin/bash
# round 3, third GPU call: bf16x6 -- remaining kernel tests, per-layer sweep (f32 vs x6), PMC on block_4x, kernel trace of the bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bf16x6.py -q -s --timeout 300 -p no:cacheprovider > gpurun_out/x6_kernel_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/x6_kernel_tests.log; grep -E "passed|failed" gpurun_out/x6_kernel_tests.log | tail -3
REPS=10 PREC=bf16x6 ONLY=dec timeout 300 python tools/conv_sweep.py > gpurun_out/sweep_x6_dec.log 2>&1; cat gpurun_out/sweep_x6_dec.log | cut -c1-400
REPS=10 PREC=bf16x6 ONLY=enc SWEEP_T=32 timeout 300 python tools/conv_sweep.py > gpurun_out/sweep_x6_enc.log 2>&1; cut -c1-260 gpurun_out/sweep_x6_enc.log
REPS=10 PREC=f32 ONLY=enc SWEEP_T=32 timeout 300 python tools/conv_sweep.py > gpurun_out/sweep_f32_enc.log 2>&1
STEMSEG_X6_PLANNER=0 timeout 300 python bench.py --precision bf16x6 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_x6_noplan.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/bench_x6_noplan.log | head -1
timeout 300 python bench.py --precision bf16x6 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_x6.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/bench_x6.log | head -1
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  (cd /tmp && PREC=bf16x6 timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_conv.py 3 0) > gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag exit $?"; grep conv3d_k3 gpurun_out/pmc_$tag.log
done
python tools/pmc_summary.py r03_x6 3 | tail -12
rm -rf gpurun_out/pmc_*/ gpucore.*
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r03 -- python $GRAFT_REPO_ROOT/bench.py --precision bf16x6 --steps 5 --warmup 2 --no-cpu-baseline --lanes 1) > gpurun_out/prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/prof_steady_x6.txt 2>&1; head -45 gpurun_out/prof_steady_x6.txt | cut -c1-170
rm -rf gpurun_out/prof
