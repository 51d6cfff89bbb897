#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_shared_convs.py -q -x -p no:cacheprovider 2>&1 | tail -15
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_shared_convs.py 2>&1 | tail -4
