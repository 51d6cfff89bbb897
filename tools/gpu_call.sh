mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for db in 0 1; do
STEMSEG_DB_ALL=$db STEMSEG_BENCH_WATCHDOG=100 timeout 150 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_dball$db.log 2>&1; echo "bench DB_ALL=$db exit $?"; tail -1 gpurun_out/bench_dball$db.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['conv_classes_eager'])"
done
STEMSEG_DB_ALL=1 timeout 300 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider -k "conv" > gpurun_out/tests_db.log 2>&1; echo "tests(DB_ALL) exit $?"; tail -2 gpurun_out/tests_db.log | cut -c1-200
