# Prefill attention with the two query heads of a KV head per workgroup: parity tests, then the prefill stage with and without.
set -o pipefail
O=gpurun_out/r2s_i
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --max-tokens 4 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"]["stage_ms_per_step_rank0"], [(k["name"],k["ms_per_step"],k["rate"]) for k in d["kernels"] if "flash" in k["name"]])
PY
}
run pair X=1
run single Q3ASR_FLASH_PAIR=0
