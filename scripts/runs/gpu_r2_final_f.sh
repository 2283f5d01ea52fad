# Measurement set on the final build: the whole GPU test suite, the default bench line with its CPU baseline, rocprofv3
# kernel stats of the same command, and the two labelled variants of the headline.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_f
mkdir -p $O
cd $R
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_ms"))
PY
}
run bench_b2048 X=1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_b2048_under_rocprof.json 2> $O/trace.err; rc=$?
python3 $R/scripts/kernel_stats.py /tmp/prof/trace > $O/bench_b2048_kernel_stats.csv
head -3 $O/bench_b2048_kernel_stats.csv | cut -c1-140
echo "rocprofv3 rc=$rc"
[ $rc -eq 0 ] || exit 1
cd $R
run bench_b2048_tokens96 X=1 --max-tokens 96 --no-cpu-baseline
run bench_b2048_len_u60-150_straggler_retire X=1 --length-dist u60-150+straggler --no-cpu-baseline
echo done
