# Measurement set: full GPU test suite, the default bench line (with CPU baseline), the labelled variants,
# other configurations, then rocprofv3 kernel stats + PMC traffic of the same default command.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final3
mkdir -p $O
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py > $O/bench_b2048.json 2> $O/bench_b2048.err || exit 1
timeout -k 10 400 python bench.py --max-tokens 96 --no-cpu-baseline > $O/bench_b2048_tokens96.json 2> $O/e1.err || exit 1
timeout -k 10 400 python bench.py --length-dist u60-150+straggler --no-cpu-baseline > $O/bench_b2048_len_retire.json 2> $O/e2.err || exit 1
Q3ASR_RETIRE=0 timeout -k 10 400 python bench.py --length-dist u60-150+straggler --no-cpu-baseline > $O/bench_b2048_len_noretire.json 2> $O/e3.err || exit 1
timeout -k 10 300 python bench.py --workload mel --batch 1024 --steps 10 --warmup 3 > $O/bench_mel_b1024.json 2> $O/e4.err || exit 1
timeout -k 10 300 python bench.py --preset 0.6b --batch 128 --no-cpu-baseline > $O/bench_0.6b_b128.json 2> $O/e5.err || exit 1
timeout -k 10 300 python bench.py --preset 0.6b --weights q4 --batch 256 --no-cpu-baseline > $O/bench_0.6b_q4_b256.json 2> $O/e6.err || exit 1
timeout -k 10 400 python bench.py --workload ragged --utterances 512 --no-cpu-baseline > $O/bench_ragged_u512.json 2> $O/e7.err || exit 1
Q3ASR_DECODE_ATT_CUS=0 timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_b2048_one_stream.json 2> $O/e8.err || exit 1
echo benches done
