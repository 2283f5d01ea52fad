# 0.6B bf16 x 128 clips and 0.6B q4 x 256: the eight-wave decode attention (one workgroup per CU) against the four-wave one.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_r
rm -rf $O; mkdir -p $O
cd $R
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_ms"))
PY
}
run b128_w4 X=1 --no-cpu-baseline --preset 0.6b --batch 128 --steps 3 --warmup 1
run b128_w8 Q3ASR_ATTN_WAVES=8 --no-cpu-baseline --preset 0.6b --batch 128 --steps 3 --warmup 1
run q4_b256_w4 X=1 --no-cpu-baseline --preset 0.6b --weights q4 --batch 256 --steps 3 --warmup 1
run q4_b256_w8 Q3ASR_ATTN_WAVES=8 --no-cpu-baseline --preset 0.6b --weights q4 --batch 256 --steps 3 --warmup 1
