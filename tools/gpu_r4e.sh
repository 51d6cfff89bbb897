#!/bin/bash
# round 4, call E: co-residency probes (stand-alone with an LDS-heavy aggressor; library kernels as victim / aggressor)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 tools/microbench/bin/valu_corun_probe 200 > gpurun_out/valu_corun_probe.txt 2>&1; echo "probe exit $?"; grep -v "aggressor none\|aggressor copy" gpurun_out/valu_corun_probe.txt | cut -c1-220
timeout 900 python tools/stem_corun_probe.py --reps 150 --victims stem > gpurun_out/stem_corun_probe.txt 2>&1; echo "corun exit $?"; cat gpurun_out/stem_corun_probe.txt | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_bf16x6.py -q -x -s --timeout 300 -p no:cacheprovider -k "checkpoint_like or overflow" 2>&1 | grep -E "checkpoint-like|passed|failed|Error|error|assert" | cut -c1-220 | tail -30
