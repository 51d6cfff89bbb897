mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider -k "kitti_shape or ytvis_full" > gpurun_out/tests_cfg.log 2>&1; echo "tests exit $?"; grep -E "parity\]|passed|failed|Error" gpurun_out/tests_cfg.log | cut -c1-220
