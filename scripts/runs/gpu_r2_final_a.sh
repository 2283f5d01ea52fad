# Measurement set, part A: the whole GPU test suite, the default bench line (with its CPU baseline), rocprofv3 kernel
# stats of the same command and the HBM traffic counters of the decode attention (one counter per pass).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_a
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > $O/bench_b2048.json 2> $O/bench_b2048.err || { tail -5 $O/bench_b2048.err; exit 1; }
echo bench done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_b2048_under_rocprof.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
python3 $R/scripts/kernel_stats.py /tmp/prof/trace > $O/bench_b2048_kernel_stats.csv || exit 1
head -5 $O/bench_b2048_kernel_stats.csv | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 500 rocprofv3 --pmc $c --kernel-include-regex decode_attn -d /tmp/prof/attn_$c -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/attn_$c.json 2> $O/attn_$c.err || { tail -5 $O/attn_$c.err; exit 1; }
  echo "pmc $c done"
done
F=$(find /tmp/prof/attn_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/prof/attn_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/scripts/pmc_summary.py $F $W decode_attn "bench.py defaults: 1.7B bf16, 2048 x 30 s clips, 448 greedy tokens, decoded as two halves of 1024 (one launch = one half, attention stream on 160 CUs)" > $O/pmc_decode_attn_b2048.json || exit 1
tail -4 $O/pmc_decode_attn_b2048.json
