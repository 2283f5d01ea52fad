# Stand-alone log-mel stage, the committed bench lines: speech-like, noise, a quarter of every clip silent, and the
# two-pass form (raw rows, then a clamp-and-scale pass over everything) for comparison; then a kernel trace.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_p
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_mel.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
for a in speechlike noise gapped; do
timeout -k 10 200 python bench.py --workload mel --batch 1024 --audio $a --steps 50 --warmup 5 > $O/bench_mel_${a}_b1024.json 2> $O/mel.err || { tail -5 $O/mel.err; exit 1; }
done
Q3ASR_MEL_TWO_PASS=1 timeout -k 10 200 python bench.py --workload mel --batch 1024 --steps 50 --warmup 5 > $O/bench_mel_b1024_two_pass.json 2> $O/mel.err || { tail -5 $O/mel.err; exit 1; }
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_mel_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --workload mel --batch 1024 --steps 20 --warmup 3 > $O/bench_mel_under_rocprof.json 2> $O/trace.err; rc=$?
python3 $R/scripts/kernel_stats.py /tmp/prof/trace > $O/bench_mel_b1024_kernel_stats.csv
head -6 $O/bench_mel_b1024_kernel_stats.csv | cut -c1-220
echo rc=$rc
