#!/bin/bash
# One gpurun call's worth of checks (what the driver runs at round end, plus the artefacts under profiles/):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_call.sh'
# Optional: ROUND=r02 (artefact prefix), SKIP_PROF=1, SKIP_TESTS=1, TESTS="-k expr"
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r02}
if [[ -z "${SKIP_TESTS:-}" ]]; then
timeout 1200 python -m pytest tests -m gpu -q -s --timeout 600 -p no:cacheprovider --durations=8 ${TESTS:-} > gpurun_out/tests.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|error" gpurun_out/tests.log | tail -2 | cut -c1-200
grep -E "^\[fullsize\]|^\[bf16x3-labels\]|^FAILED|^ERROR" gpurun_out/tests.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log | cut -c1-200
fi
timeout 400 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_final.log > gpurun_out/bench_final.json; cut -c1-160 gpurun_out/bench_final.json
timeout 300 python bench.py --sequence --steps 3 --warmup 1 > gpurun_out/bench_seq64.log 2>&1; echo "bench --sequence exit $?"; tail -1 gpurun_out/bench_seq64.log > gpurun_out/bench_seq64.json; cut -c1-200 gpurun_out/bench_seq64.json
if [[ -z "${SKIP_PROF:-}" ]]; then
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/prof_graph_steady.txt 2>&1; head -4 gpurun_out/prof_graph_steady.txt | cut -c1-170
rm -f gpurun_out/prof_graph/*.db gpucore.*
fi
