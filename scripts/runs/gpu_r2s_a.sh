# Shared LM head at the step boundary: split-decode tests, then the headline bench with and without it.
set -o pipefail
O=gpurun_out/r2s_a
mkdir -p $O
timeout -k 10 500 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_retire.py -m gpu -x -q --timeout 170 --timeout-method thread > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["achieved"], [ (k["name"],k["ms_per_step"]) for k in d["kernels"][:7]])
PY
}
run lm_shared X=1
run lm_whole Q3ASR_LM_SPLIT=0
run lm_shared_lead12 Q3ASR_LM_SPLIT_LEAD=12
