mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider -k "encoder_full_size" > gpurun_out/tests_e.log 2>&1; echo "tests exit $?"; grep -E "parity\]|passed|failed|Error" gpurun_out/tests_e.log | cut -c1-200
