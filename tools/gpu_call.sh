mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 250 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_final.log > gpurun_out/bench_final.json; python -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['conv_classes_eager'], d['cpu_baseline']['value'])"
STEMSEG_BENCH_WATCHDOG=100 timeout 150 python bench.py --precision bf16x3 --no-cpu-baseline > gpurun_out/bench_bf16x3.log 2>&1; echo "bench bf exit $?"; tail -1 gpurun_out/bench_bf16x3.log > gpurun_out/bench_bf16x3.json; cut -c1-120 gpurun_out/bench_bf16x3.json
rm -rf gpurun_out/prof_graph
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lanes 1) > gpurun_out/prof_graph.log 2>&1; echo "prof graph exit $?"
db=$(find gpurun_out/prof_graph -name "*.db" | head -1); python tools/prof_steady.py $db 2 > gpurun_out/prof_graph_steady.txt 2>&1; head -3 gpurun_out/prof_graph_steady.txt | cut -c1-170
rm -f gpurun_out/prof_graph/*.db gpucore.*
