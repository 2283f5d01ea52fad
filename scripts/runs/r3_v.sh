# After lowering the split threshold to 640: the affected tests, then the ragged workload and two group sizes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_v
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_retire.py -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"), d["config"].get("decode_attention_cus"))
PY
}
run bench_ragged_u512 X=1 --no-cpu-baseline --workload ragged --utterances 512 --steps 2 --warmup 1
run bench_b768 X=1 --no-cpu-baseline --batch 768 --steps 2 --warmup 1
run bench_b1024 X=1 --no-cpu-baseline --batch 1024 --steps 2 --warmup 1
