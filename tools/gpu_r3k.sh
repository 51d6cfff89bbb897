#!/bin/bash
# round 3, step k: PMC counters of the block_4x 3x3x3 conv in the f16x3 mode
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
export PREC=f16x3
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_conv.py 3 0) > gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag exit $?"
done
python tools/pmc_summary.py r03_f16x3 3 | tail -24
python - <<'PY'
import glob, sqlite3
for d in glob.glob("gpurun_out/pmc_SQ_INSTS_LDS*/**/*.db", recursive=True):
    con = sqlite3.connect(d)
    for row in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if "conv_igemm" in row[0]: print(row[1], row[2] / row[3])
PY
rm -rf gpurun_out/pmc_*/ gpucore.*
