# Decode attention: four workgroups per CU (128 registers, 37 KB of LDS) against three.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_q
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
Q3ASR_ATTN_OCC4=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py -x -q --timeout 300 --timeout-method thread > $O/tests_occ4.log 2>&1 || { tail -30 $O/tests_occ4.log; exit 1; }
tail -1 $O/tests_occ4.log
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_ms"))
PY
}
run occ3 X=1 --no-cpu-baseline --steps 2 --warmup 1
run occ4 Q3ASR_ATTN_OCC4=1 --no-cpu-baseline --steps 2 --warmup 1
