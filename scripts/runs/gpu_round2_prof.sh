# rocprofv3 evidence: kernel-trace stats of the default bench command, then one counter per pass
# (FETCH_SIZE, WRITE_SIZE) for the decode attention and the log-mel kernel.  Outputs under gpurun_out/prof3/.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof3
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof3/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/prof3/trace_bench.json 2> $R/gpurun_out/prof3/trace.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --kernel-include-regex decode_attn -d $R/gpurun_out/prof3/attn_$c -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $R/gpurun_out/prof3/attn_$c.json 2> $R/gpurun_out/prof3/attn_$c.err || exit 1
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-include-regex mel_kernel -d $R/gpurun_out/prof3/mel_$c -o p --output-format csv -- python3 $R/bench.py --workload mel --batch 1024 --steps 1 --warmup 0 > $R/gpurun_out/prof3/mel_$c.json 2> $R/gpurun_out/prof3/mel_$c.err || exit 1
done
find $R/gpurun_out/prof3 -name "*.csv" | head -20
