set -o pipefail
O=gpurun_out/split_j
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -4 $O/tests.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread > $O/tests2.log 2>&1
rc=$?; tail -4 $O/tests2.log; echo "pytest (second run) rc=$rc"
