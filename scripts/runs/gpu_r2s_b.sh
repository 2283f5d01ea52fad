# Diagnose the split-decode test with the host-side trace, then the remaining split tests and the A/B bench.
set -o pipefail
O=gpurun_out/r2s_b
mkdir -p $O
Q3ASR_TRACE=1 timeout -k 10 120 python -u -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "split_decode_equals or large_group" --timeout 60 --timeout-method thread > $O/trace.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/trace.log; tail -30 $O/trace.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_retire.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -4 $O/tests.log
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
