# Whole GPU suite + smoke + the mel bench lines after the log-mel kernel work.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_n
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
cp gpurun_out/parity_report.txt $O/
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
