mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for pl in 1 0 1 0; do
STEMSEG_PLANNER=$pl STEMSEG_BENCH_WATCHDOG=100 timeout 150 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/bench_s.log 2>&1; echo -n "planner=$pl lanes=3: "; tail -1 gpurun_out/bench_s.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
