mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/tests.log | cut -c1-200
for pr in f32 bf16x3; do
STEMSEG_BENCH_WATCHDOG=100 timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision $pr > gpurun_out/bench_$pr.log 2>&1; echo "bench $pr exit $?"; tail -1 gpurun_out/bench_$pr.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['launches'])"
done
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/prof_graph.log 2>&1; echo "prof graph exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 3 > gpurun_out/prof_graph_steady.txt 2>&1; head -12 gpurun_out/prof_graph_steady.txt | cut -c1-170
rm -f gpurun_out/prof_graph/*.db gpucore.*
