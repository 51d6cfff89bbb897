#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "cluster" 2>&1 | tail -2
b() { timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; cl=[k for k in r['hbm_kernels_eager']['kernels'] if 'cluster' in k['kernel']]; print(d['value'], 'cluster us/clip', cl[0]['us_per_clip'] if cl else None, 'mismatch', d['config']['determinism']['mismatching'])"; }
for ppt in 1 2 4; do echo ppt $ppt; STEMSEG_CLUSTER_PPT=$ppt b; done
