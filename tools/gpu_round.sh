#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): parity tests, smoke, a short bench and a rocprofv3 kernel trace of it.
# Everything lands in gpurun_out/ (merged back by gpurun).  Usage: tools/gpu_round.sh [tests|bench|prof|all]
set -u
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/device.txt
if [[ $what == all || $what == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider > gpurun_out/tests.log 2>&1
  echo "pytest exit $?" >> gpurun_out/tests.log
  tail -5 gpurun_out/tests.log
fi
if [[ $what == all || $what == smoke ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
fi
if [[ $what == all || $what == bench ]]; then
  timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log; tail -4 gpurun_out/bench.log
fi
if [[ $what == all || $what == prof ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1) > gpurun_out/prof.log 2>&1
  echo "prof exit $?" >> gpurun_out/prof.log
  db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/prof_steady.txt 2>&1; head -12 gpurun_out/prof_steady.txt; tail -3 gpurun_out/prof.log
fi
if [[ $what == pmc ]]; then
  # PMC passes on the dominant kernel only, counters in their own runs (no trace domains besides kernel-trace)
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40)
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_conv.py 3 ${PMC_CFG:-0}) > gpurun_out/pmc_$tag.log 2>&1
    echo "pmc $tag exit $?"; tail -2 gpurun_out/pmc_$tag.log
  done
  python tools/pmc_summary.py ${ROUND:-r02} 3 | tail -12
  rm -rf gpurun_out/pmc_*/ gpucore.*
fi
