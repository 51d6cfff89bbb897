mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/tests.log | cut -c1-200
for pr in f32 bf16x3; do
STEMSEG_BENCH_WATCHDOG=100 timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision $pr > gpurun_out/bench_$pr.log 2>&1; echo "bench $pr exit $?"; tail -1 gpurun_out/bench_$pr.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['launches'])"
done
rm -f gpucore.*
