#!/bin/bash
# fused-tail A/B on the encoder alone: for each library in LIBS (tags of lib/libstemseg_hip_<tag>.so; "base" = the product library) a kernel
# trace of tools/enc_bench.py and the fused kernels' lines of it.  Usage: LIBS="base ftp1 ..." bash tools/gpu_r6_ft.sh <out tag>
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${1:-r06c}
out=gpurun_out/${R}_fused_tail_ab.txt; : > $out
for tag in ${LIBS:-base}; do
  lib=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_$tag.so; [[ $tag == base ]] && lib=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip.so
  extra=""; [[ $tag == nofuse ]] && { lib=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip.so; extra="--no-fuse"; }
  rm -rf gpurun_out/prof_ft
  (cd /tmp && STEMSEG_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_ft -o ft -- python $GRAFT_REPO_ROOT/tools/enc_bench.py $extra ${ENC_ARGS:-}) > gpurun_out/prof_ft.log 2>&1
  echo "== $tag: $(grep '^encoder' gpurun_out/prof_ft.log)" >> $out
  db=$(find gpurun_out/prof_ft -name "*.db" | head -1)
  python tools/prof_steady.py $db 3 2>&1 | grep -E "steady|fused_tail|1, 1, 1, 32|16, 4, 2, 1, 8, 16" | cut -c1-175 >> $out
done
rm -rf gpurun_out/prof_ft gpucore.*
cat $out
