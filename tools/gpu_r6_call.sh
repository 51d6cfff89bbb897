#!/bin/bash
# working call: a -k subset of the GPU suite, then interleaved bench A/B over environment switches, then other workloads' lines.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${1:-r06t}
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider ${TESTS_K:+-k "$TESTS_K"} > gpurun_out/${R}_gpu_tests.log 2>&1; echo "tests exit $?"
grep -E "passed|failed|error" gpurun_out/${R}_gpu_tests.log | tail -2 | cut -c1-200
grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/${R}_gpu_tests.log | head -20 | cut -c1-250
for rep in 1 2 3; do
  for arm in ${ENV_ARMS:-}; do
    line=$(env ${arm//,/ } timeout 400 python bench.py --no-cpu-baseline --no-alt-precision --no-sequence-leg --steps 60 2>/dev/null | grep "^{" | tail -1)
    echo "$line" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); h={k['kernel']:k['us_per_clip'] for k in j['roofline']['hbm_kernels_eager']['kernels']}
print('%-40s rep $rep: %.2f clips/s bitsum %s fpn_add %.1f us/clip' % ('$arm', j['value'], j['config'].get('first_clip_bitsum'), h.get('upsample2x_add (FPN top-down)',0)))" 2>&1 | tail -1
  done
done
for w in ${WORKLOADS:-}; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-alt-precision 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${R}_bench_$w.json
  python -c "
import json
j=json.load(open('gpurun_out/${R}_bench_$w.json')); print('$w', j['value'], {k:(v['ms_per_clip'],v['frac_of_mfma_peak']) for k,v in j['roofline']['conv_classes_eager'].items()})"
done
