# Round 3, run e: two attention streams as the default of the two-half decode: split-decode tests, then the headline
# bench with two streams (default) and with one (Q3ASR_ATT_STREAMS=1), same box, same process order.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_f
mkdir -p $O
cd $R
timeout -k 10 800 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_retire.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_ms"))
PY
}
run bench_default X=1 --steps 2 --warmup 1
run bench_one_mm_stream Q3ASR_MM_STREAMS=1 --steps 2 --warmup 1
run bench_default_again X=1 --steps 2 --warmup 1
run bench_tokens96 X=1 --steps 3 --warmup 1 --max-tokens 96
