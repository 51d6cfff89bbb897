mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 200 python -m pytest tests -m gpu -q -s --timeout 150 -p no:cacheprovider -k "graphed or cluster" > gpurun_out/tests_graph.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/tests_graph.log | cut -c1-300
rm -f gpucore.*
