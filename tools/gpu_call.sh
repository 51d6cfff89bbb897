mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
S=$(date +%s); timeout 250 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench exit $? in $(( $(date +%s) - S )) s"; tail -1 gpurun_out/bench_final.log > gpurun_out/bench_final.json; python -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['conv_classes_eager'], d['cpu_baseline']['value'])"
