#!/bin/bash
# round 5, GPU call 8: whole-step HBM (fixed script); the input-two-chunks-ahead build of the 1x1 tiles vs the shipped one; PMC on the layer-3 1x1 launches
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r05h}
L=$PWD/stem-seg_amd/stemseg_amd/lib
bash tools/pmc_step.sh > gpurun_out/r05_pmc_whole_step_hbm.txt 2>&1; tail -14 gpurun_out/r05_pmc_whole_step_hbm.txt
for tag in "" _ina; do
  STEMSEG_HIP_LIB=$L/libstemseg_hip$tag.so PREC=f16x3 SWEEP_T=32 REPS=15 ONLY=enc timeout 300 python tools/conv_sweep.py > gpurun_out/${R}_sweep$tag.txt 2>&1
  echo "== sweep lib${tag:-_default}"; grep -E "conv1|conv3|fpn_inner" gpurun_out/${R}_sweep$tag.txt | cut -c1-330
done
for rep in 1 2; do
  for tag in "" _ina; do
    STEMSEG_HIP_LIB=$L/libstemseg_hip$tag.so timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-sequence-leg --no-alt-precision > gpurun_out/ab8.log 2>&1
    echo "lib${tag:-_default} rep $rep: $(grep '^{' gpurun_out/ab8.log | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); c=j["roofline"]["conv_classes_eager"]; print(j["value"], {k:c[k]["ms_per_clip"] for k in c})' 2>&1 | tail -1)"
  done
done
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD|SQ)_[A-Z0-9_]+" | sort -u > gpurun_out/${R}_counters_available.txt; wc -l gpurun_out/${R}_counters_available.txt
for shape in l3conv1 l3conv3; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCC_HIT TCC_MISS"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-30)
    rm -rf gpurun_out/pmcs_$tag
    (cd /tmp && timeout 120 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmcs_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_shape.py $shape 5) > gpurun_out/pmcs_$tag.log 2>&1
    echo "pmc $shape $tag exit $?"
    python - "$shape" "$tag" <<'PY'
import glob, sqlite3, sys
shape, tag = sys.argv[1], sys.argv[2]
dbs = glob.glob("gpurun_out/pmcs_%s/**/*.db" % tag, recursive=True)
if not dbs:
    print("   no database"); sys.exit(0)
con = sqlite3.connect(dbs[0])
try:
    rows = con.execute("select counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%conv_igemm%' group by counter_name").fetchall()
except Exception as e:
    rows = []
    print("   query failed:", e)
for name, v, n in rows:
    print("   %-28s %.4g per launch (%d launches)" % (name, v / max(n, 1), n))
PY
    rm -rf gpurun_out/pmcs_$tag
  done
done > gpurun_out/${R}_pmc_layer3_1x1.txt 2>&1
cat gpurun_out/${R}_pmc_layer3_1x1.txt | cut -c1-200
