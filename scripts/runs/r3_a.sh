# Round 3, first run: the whole GPU suite (every measured distance is reported), then the default bench line.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_a
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread > $O/tests.log 2>&1
echo "pytest rc=$?" | tee -a $O/tests.log
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
tail -30 $O/tests.log
