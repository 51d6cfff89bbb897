mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
STEMSEG_BENCH_WATCHDOG=150 timeout 200 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_final.log > gpurun_out/bench_final.json; cut -c1-200 gpurun_out/bench_final.json
STEMSEG_BENCH_WATCHDOG=100 timeout 150 python bench.py --clips-per-step 1 --no-cpu-baseline > gpurun_out/bench_nc1.log 2>&1; tail -1 gpurun_out/bench_nc1.log | cut -c1-160
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/prof_graph.log 2>&1; echo "prof graph exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/prof_graph_steady.txt 2>&1; head -3 gpurun_out/prof_graph_steady.txt | cut -c1-170
rm -f gpurun_out/prof_graph/*.db gpucore.*
