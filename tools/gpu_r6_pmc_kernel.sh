#!/bin/bash
# PMC counters of one kernel family inside tools/enc_bench.py (separate rocprofv3 --pmc passes, kernel-trace only).
# Usage: KPAT="fused_tail_kernel<stemseg::FusedTailCfg<256" bash tools/gpu_r6_pmc_kernel.sh <out tag>     (STEMSEG_HIP_LIB / ENC_ARGS pass through)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${1:-r06e}; out=gpurun_out/${R}_pmc_kernel.txt; : > $out
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf gpurun_out/pmc_k
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_k -o pmc -- python $GRAFT_REPO_ROOT/tools/enc_bench.py --passes 2 ${ENC_ARGS:-}) > gpurun_out/pmc_k.log 2>&1
  db=$(find gpurun_out/pmc_k -name "*.db" | head -1)
  python - "$db" "${KPAT:-fused_tail_kernel}" >> $out <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
try:
    rows = con.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by counter_name", ("%" + sys.argv[2] + "%",)).fetchall()
except Exception as e:
    print("query failed:", e); rows = []
for name, v, n in rows:
    print("%-28s per launch %14.1f   (%d launches)" % (name, v / max(n, 1), n))
try:
    t = con.execute("select avg(end - start), count(*) from kernels where name like ?", ("%" + sys.argv[2] + "%",)).fetchone()
    print("%-28s %.1f us over %d launches (profiled pass)" % ("kernel time", t[0] / 1e3, t[1]))
except Exception as e:
    print("time query failed:", e)
PY
  tail -2 gpurun_out/pmc_k.log | grep -i "error\|fail" 
done
rm -rf gpurun_out/pmc_k gpucore.*
cat $out
