mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider > gpurun_out/tests.log 2>&1; echo "tests exit $?"; tail -2 gpurun_out/tests.log | cut -c1-250
STEMSEG_BENCH_WATCHDOG=100 timeout 150 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/bench_e.log 2>&1; tail -1 gpurun_out/bench_e.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['conv_classes_eager'])"
