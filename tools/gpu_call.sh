mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider -k "sequence_end_to_end" > gpurun_out/tests_s.log 2>&1; echo "tests exit $?"; grep -E "parity\] sequence|passed|failed|Error|error|assert" gpurun_out/tests_s.log | cut -c1-250 | head -20
