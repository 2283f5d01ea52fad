# Round 3, run h: RMSNorm walking rows with the grid's stride: model / retire tests, then the 96-token bench line.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_i
mkdir -p $O
cd $R
timeout -k 10 800 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_retire.py tests/test_gpu_golden.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), [(k["name"], k["ms_per_step"]) for k in d["kernels"] if "norm" in k["name"] or "rope" in k["name"]])
PY
}
run tokens96 X=1 --steps 3 --warmup 1 --max-tokens 96
run tokens448 X=1 --steps 2 --warmup 1
