# Where the two-half decode starts to pay: one stream against the split, by model and group size.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_u2
rm -rf $O; mkdir -p $O
cd $R
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0")["decode"], (d.get("roofline") or {}).get("achieved"), d["config"].get("decode_attention_cus"))
PY
}
for b in 256 384 512; do
run 17b_b${b}_one X=1 --no-cpu-baseline --batch $b --steps 1 --warmup 1
run 17b_b${b}_split Q3ASR_DECODE_ATT_CUS=160 --no-cpu-baseline --batch $b --steps 1 --warmup 1
done
for b in 256 512; do
run 06b_b${b}_one X=1 --no-cpu-baseline --preset 0.6b --batch $b --steps 2 --warmup 1
run 06b_b${b}_split Q3ASR_DECODE_ATT_CUS=160 --no-cpu-baseline --preset 0.6b --batch $b --steps 2 --warmup 1
run 06bq4_b${b}_one X=1 --no-cpu-baseline --preset 0.6b --weights q4 --batch $b --steps 2 --warmup 1
run 06bq4_b${b}_split Q3ASR_DECODE_ATT_CUS=160 --no-cpu-baseline --preset 0.6b --weights q4 --batch $b --steps 2 --warmup 1
done
