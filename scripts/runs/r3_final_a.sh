# Round 3 evidence, part A: the whole GPU suite, smoke, the default bench line (with its CPU baseline), rocprofv3 kernel
# stats of the same command, the HBM counters of the decode attention (one counter per pass), the idle time between
# attention launches from a kernel trace.  Nothing further runs after a bad exit.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_final_a
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
cp gpurun_out/parity_report.txt $O/
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 700 python bench.py > $O/bench_b2048.json 2> $O/bench_b2048.err || { tail -5 $O/bench_b2048.err; exit 1; }
cut -c1-200 $O/bench_b2048.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_b2048_under_rocprof.json 2> $O/trace.err; rc=$?
python3 $R/scripts/kernel_stats.py /tmp/prof/trace > $O/bench_b2048_kernel_stats.csv
head -4 $O/bench_b2048_kernel_stats.csv | cut -c1-160
echo "rocprofv3 kernel-trace rc=$rc"
[ $rc -eq 0 ] || { tail -3 $O/trace.err; exit 1; }
rm -rf /tmp/prof/trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 500 rocprofv3 --pmc $c --kernel-include-regex decode_attn -d /tmp/prof/attn_$c -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/attn_$c.json 2> $O/attn_$c.err; rc=$?
  echo "pmc $c rc=$rc"
  [ $rc -eq 0 ] || { tail -3 $O/attn_$c.err; exit 1; }
done
F=$(find /tmp/prof/attn_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/prof/attn_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/scripts/pmc_summary.py $F $W decode_attn "bench.py defaults: 1.7B bf16, 2048 x 30 s clips, 448 greedy tokens, decoded as two halves of 1024 (one launch = one half, each half's attention on a stream of its own confined to 160 CUs)" > $O/pmc_decode_attn_b2048.json || exit 1
tail -4 $O/pmc_decode_attn_b2048.json
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 1 --warmup 0 --max-tokens 24 > $O/gaps_bench.json 2> $O/gaps_trace.err; rc=$?
[ $rc -eq 0 ] || { tail -3 $O/gaps_trace.err; exit 1; }
python3 $R/scripts/trace_gaps.py /tmp/prof/trace decode_attn | tee $O/gaps.txt
