#!/bin/bash
# round 4, call G: f16x3 with two staged weight planes (default build) vs three (lib _w3): kernel tests, bench A/B, soak; VALU-stem soak variations
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
W3=$PWD/stem-seg_amd/stemseg_amd/lib/libstemseg_hip_w3.so
timeout 600 python -m pytest tests/test_gpu_bf16x6.py -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2
b() { timeout 300 python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(sys.argv[1:], d['value'], 'k3', r['conv_classes_eager']['conv3x3x3']['ms_per_clip'], 'k2', r['conv_classes_eager']['conv1x3x3']['ms_per_clip'], 'k1', r['conv_classes_eager']['conv1x1x1']['ms_per_clip'], 'mismatch', d['config']['determinism']['mismatching'], d['config']['determinism']['clip_results_checked'])" "$@"; }
echo "two planes:"; b; b
echo "three planes:"; STEMSEG_HIP_LIB=$W3 b; STEMSEG_HIP_LIB=$W3 b
timeout 900 python tools/soak_probe.py --workload davis --lanes 3 --reps 400 --max-reports 2 > gpurun_out/soak5_davis_f16x3_w2.txt 2>&1; echo "exit $?"; grep -n "S0 \|RESULT" gpurun_out/soak5_davis_f16x3_w2.txt | cut -c1-250 | tail -4
timeout 900 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 400 --max-reports 2 > gpurun_out/soak5_ytvis_f16x3_w2.txt 2>&1; echo "exit $?"; grep -n "S0 \|RESULT" gpurun_out/soak5_ytvis_f16x3_w2.txt | cut -c1-250 | tail -4
# what the VALU stem needs to go wrong: two lanes; every lane on the same batch
STEMSEG_STEM=valu timeout 600 python tools/soak_probe.py --workload ytvis --lanes 2 --reps 300 --precision bf16x6 --max-reports 1 > gpurun_out/soak5_valu_2lanes.txt 2>&1; grep -n "RESULT" gpurun_out/soak5_valu_2lanes.txt | cut -c1-200
STEMSEG_STEM=valu timeout 600 python tools/soak_probe.py --workload ytvis --lanes 3 --reps 200 --precision bf16x6 --same-batch --max-reports 1 > gpurun_out/soak5_valu_samebatch.txt 2>&1; grep -n "RESULT" gpurun_out/soak5_valu_samebatch.txt | cut -c1-200
