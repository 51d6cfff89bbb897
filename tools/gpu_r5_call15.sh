#!/bin/bash
# the legs of the evidence script that a stale experiment library / the tools' view of the decoder workspaces broke, re-run on HEAD
cd "$(dirname "$0")/.." && mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=r05
bash tools/pmc_step.sh > gpurun_out/${R}_pmc_whole_step_hbm.txt 2>&1; tail -3 gpurun_out/${R}_pmc_whole_step_hbm.txt
EXP=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_exp.so
STEMSEG_HIP_LIB=$EXP STEMSEG_STEM=valu timeout 400 python tools/graph_corun_probe.py --rounds 100 --aggressors k1,k1_bf16x6,k1_f32,k1_big,k1_wide,k2flat,k3,stream,stem --modes ee,gg,eg,ge > gpurun_out/${R}_graph_corun_valu_stem.txt 2>&1; echo "corun valu exit $?"; grep -E "total" gpurun_out/${R}_graph_corun_valu_stem.txt
for wl in davis ytvis; do
  timeout 900 python tools/soak_probe.py --workload $wl --lanes 3 --reps 400 > gpurun_out/${R}_soak_${wl}.txt 2>&1; echo "soak $wl exit $?"; tail -1 gpurun_out/${R}_soak_${wl}.txt | cut -c1-200
done
