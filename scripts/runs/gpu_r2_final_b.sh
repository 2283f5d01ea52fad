# Measurement set, part B: rocprofv3 kernel stats of the default bench command and the HBM traffic counters of the decode
# attention (one counter per pass).  The summaries are written even if the profiled program ends badly; nothing further
# runs after a bad exit.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_b2048_under_rocprof.json 2> $O/trace.err; rc=$?
python3 $R/scripts/kernel_stats.py /tmp/prof/trace > $O/bench_b2048_kernel_stats.csv
head -4 $O/bench_b2048_kernel_stats.csv | cut -c1-160
echo "rocprofv3 kernel-trace rc=$rc"
[ $rc -eq 0 ] || { tail -3 $O/trace.err; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 500 rocprofv3 --pmc $c --kernel-include-regex decode_attn -d /tmp/prof/attn_$c -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/attn_$c.json 2> $O/attn_$c.err; rc=$?
  echo "pmc $c rc=$rc"
  [ $rc -eq 0 ] || { tail -3 $O/attn_$c.err; break; }
done
F=$(find /tmp/prof/attn_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/prof/attn_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 $R/scripts/pmc_summary.py $F $W decode_attn "bench.py defaults: 1.7B bf16, 2048 x 30 s clips, 448 greedy tokens, decoded as two halves of 1024 (one launch = one half, attention stream on 160 CUs)" > $O/pmc_decode_attn_b2048.json
tail -4 $O/pmc_decode_attn_b2048.json
[ $rc -eq 0 ]
