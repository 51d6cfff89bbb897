#!/bin/bash
# HBM traffic of the whole bench step (all kernels): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short
# single-lane bench run; prints per-kernel-family and total GB per clip.  Usage (on the GPU box): bash tools/pmc_step.sh
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_step_$c
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_step_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precision --no-sequence-leg --lanes 1 --no-graph --no-overlap) > gpurun_out/pmc_step_$c.log 2>&1
  echo "pmc $c exit $?"
done
python - <<'PY'
import glob, sqlite3, collections
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob("gpurun_out/pmc_step_%s/**/*.db" % c, recursive=True)[0]
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name=? group by kernel_name", (c,)).fetchall()
    fam = collections.defaultdict(float)
    for name, v, n in rows:
        key = "conv3x3x3" if "ConvCfg<3, 3, 3" in name else "conv1x3x3" if "ConvCfg<1, 3, 3" in name else "conv1x1x1" if "ConvCfg<1, 1, 1" in name else \
              "splitk_reduce" if "splitk" in name else name.split("(")[0].replace("void ", "").replace("stemseg::", "")[:40]
        fam[key] += v * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)      # KB -> bytes; x2 gfx950 fetch correction
    tot[c] = fam
# the run executes: 1 warm-up + 2 pre-runs... count clips from the stem kernel launches instead
db = glob.glob("gpurun_out/pmc_step_FETCH_SIZE/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
n_enc = con.execute("select count(*) from counters_collection where (kernel_name like '%stem_conv7x7%' or kernel_name like '%stem_s2d%') and counter_name='FETCH_SIZE'").fetchone()[0]
n_dec = con.execute("select count(*) from counters_collection where kernel_name like '%heads_kernel%' and counter_name='FETCH_SIZE'").fetchone()[0]
clips = n_enc * 4.0                                   # (bench default: 4 clips per encoder pass; the decoders take all of them per launch)
print("encoder passes %d, clips %.1f" % (n_enc, clips))
keys = sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]), key=lambda k: -(tot["FETCH_SIZE"].get(k, 0) + tot["WRITE_SIZE"].get(k, 0)))
print("%-44s %10s %10s   (GB per clip)" % ("kernel family", "fetch", "write"))
# One-time set-up is not step traffic: the hipMemsetAsync of *_init_workspace and torch.zeros of the zero-haloed FPN blocks run ONCE per
# workspace (first call of a shape / lane) -- the runtime's fill kernel -- and the weight packing once per precision.  They are listed
# apart (GB in total, not per clip) and left out of the per-clip sum (round 5 counted the fill kernel in: 0.44 of its 19.83 GB per clip).
SETUP = ("__amd_rocclr_fillBuffer", "pack_conv_weight", "absmax_rows", "canary_fill")
F = W = 0.0
setup = []
for k in keys:
    if any(k.startswith(x) or x in k for x in SETUP):
        setup.append((k, tot["FETCH_SIZE"].get(k, 0) / 1e9, tot["WRITE_SIZE"].get(k, 0) / 1e9))
        continue
    f, w = tot["FETCH_SIZE"].get(k, 0) / clips / 1e9, tot["WRITE_SIZE"].get(k, 0) / clips / 1e9
    F += f; W += w
    if f + w > 0.005:
        print("%-44s %10.3f %10.3f" % (k, f, w))
print("%-44s %10.3f %10.3f   total %.2f GB per clip" % ("ALL (steady state)", F, W, F + W))
for k, f, w in setup:
    print("one-time set-up  %-27s %10.3f %10.3f   (GB in the whole run, not per clip)" % (k, f, w))
PY
rm -rf gpurun_out/pmc_step_FETCH_SIZE gpurun_out/pmc_step_WRITE_SIZE gpucore.*
