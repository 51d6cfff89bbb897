#!/bin/bash
# PMC passes (own runs, kernel-trace only) on the layer3 1x1 convolutions; per-kernel sums -> gpurun_out/r02_pmc_conv1x1_layer3.txt
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPS=5
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pk1_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_k1.py $REPS) > gpurun_out/pk1_$tag.log 2>&1
  echo "pmc $tag exit $?"; grep "us/launch" gpurun_out/pk1_$tag.log
done
python - <<'PY'
import glob, sqlite3, collections
REPS, WARM = 5, 2
vals = collections.defaultdict(dict)
for d in glob.glob("gpurun_out/pk1_*/"):
    for db in glob.glob(d + "**/*.db", recursive=True):
        con = sqlite3.connect(db)
        for name, cname, v, n in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if "conv_igemm" not in name:
                continue
            cfg = name.split("ConvCfg<")[1].split(">")[0]
            vals[cfg][cname] = (v, n)
lines = ["# rocprofv3 --pmc passes on tools/pmc_k1.py: the layer3 1x1 convolutions at the bench shape (V = 51 840 voxels), per launch",
         "# (counters summed over XCDs / SIMDs; each conv launched %d + %d times per pass)" % (WARM, REPS)]
for cfg, d in vals.items():
    n = max(v[1] for v in d.values())
    per = lambda k: d[k][0] / d[k][1] if k in d else float("nan")
    lines.append("ConvCfg<%s>  (%d launches)" % (cfg, n))
    if "FETCH_SIZE" in d:
        lines.append("   FETCH_SIZE x2 %.3f GB, WRITE_SIZE %.3f GB per launch" % (per("FETCH_SIZE") * 1024 * 2 / 1e9, per("WRITE_SIZE") * 1024 / 1e9))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        lines.append("   MFMA busy cycles / SIMD %.4g, active cycles / XCD %.4g -> MFMA-pipe utilisation %.1f %%; waves %.0f, MFMA instructions %.4g"
                     % (per("SQ_VALU_MFMA_BUSY_CYCLES") / 1024, per("GRBM_GUI_ACTIVE") / 8, 100 * (per("SQ_VALU_MFMA_BUSY_CYCLES") / 1024) / (per("GRBM_GUI_ACTIVE") / 8),
                        per("SQ_WAVES"), per("SQ_INSTS_MFMA")))
    if "SQ_WAVE_CYCLES" in d:
        wc = per("SQ_WAVE_CYCLES")
        lines.append("   wave time: parked (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %%"
                     % (100 * per("SQ_WAIT_ANY") / wc, 100 * per("SQ_WAIT_INST_ANY") / wc, 100 * per("SQ_ACTIVE_INST_ANY") / wc))
lines += [l.strip() for l in open(glob.glob("gpurun_out/pk1_FETCH_SIZE.log")[0]) if "us/launch" in l]
open("gpurun_out/r02_pmc_conv1x1_layer3.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf gpurun_out/pk1_*/ gpucore.*
