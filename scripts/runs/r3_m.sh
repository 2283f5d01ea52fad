# Kernel trace of the stand-alone log-mel stage (1024 clips).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_m
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --workload mel --batch 1024 --steps 20 --warmup 3 > $O/bench_mel_under_rocprof.json 2> $O/trace.err; rc=$?
python3 $R/scripts/kernel_stats.py /tmp/prof/trace > $O/mel_kernel_stats.csv
head -8 $O/mel_kernel_stats.csv | cut -c1-200
echo rc=$rc
