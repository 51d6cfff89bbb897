#!/bin/bash
# (record of a removed experiment: libstemseg_hip_pp.so was the build with the ping-pong tiles behind tile_cfg 6; profiles/r05i_pingpong_tiles.txt)
# ping-pong tiles (tile_cfg 6) against the launcher's choice (0) and the plain big tile (1), same library
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export STEMSEG_HIP_LIB=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_pp.so
for c in 0 1 6; do CFG=$c timeout 300 python tools/ab_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/pp_cfg$c.txt; done
echo "== launcher's choice | big tile | ping-pong"
paste -d'|' <(cut -c1-70 gpurun_out/pp_cfg0.txt) <(cut -c31-70 gpurun_out/pp_cfg1.txt) <(cut -c31-70 gpurun_out/pp_cfg6.txt)
