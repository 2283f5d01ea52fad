# Decode attention with K and V rows two tiles ahead (two workgroups per CU) against the default.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_s
rm -rf $O; mkdir -p $O
cd $R
Q3ASR_ATTN_DEEP=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py -x -q --timeout 300 --timeout-method thread > $O/tests_deep.log 2>&1 || { tail -30 $O/tests_deep.log; exit 1; }
tail -1 $O/tests_deep.log
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_ms"))
PY
}
run b128 X=1 --no-cpu-baseline --preset 0.6b --batch 128 --steps 3 --warmup 1
run b128_deep Q3ASR_ATTN_DEEP=1 --no-cpu-baseline --preset 0.6b --batch 128 --steps 3 --warmup 1
run b2048_deep Q3ASR_ATTN_DEEP=1 --no-cpu-baseline --steps 2 --warmup 1
