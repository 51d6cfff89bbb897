mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider -k "mask or preprocess or inference_model or track" > gpurun_out/tests_new.log 2>&1; echo "tests exit $?"; grep -E "parity\] (masks|preprocess)|passed|failed|Error" gpurun_out/tests_new.log | cut -c1-220
rm -f gpucore.*
