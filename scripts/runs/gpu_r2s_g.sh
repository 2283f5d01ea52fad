# Two attention streams on the attention's CUs: split-decode tests with it on, then the headline bench with and without.
set -o pipefail
O=gpurun_out/r2s_g
mkdir -p $O
Q3ASR_ATT_STREAMS=2 timeout -k 10 400 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_retire.py tests/test_gpu_fullsize.py -m gpu -x -q -k "split or large_group or retire" > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])
PY
}
run att2 Q3ASR_ATT_STREAMS=2
run att1 X=1
