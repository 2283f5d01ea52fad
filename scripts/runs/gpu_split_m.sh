set -o pipefail
O=gpurun_out/split_m
mkdir -p $O
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --max-tokens 96 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["config"]["decode_attention_cus"])
PY
}
run 06b_bf16_att192 Q3ASR_DECODE_ATT_CUS=192 --preset 0.6b
run 06b_bf16_att176 Q3ASR_DECODE_ATT_CUS=176 --preset 0.6b
run 17b_q4_att128 Q3ASR_DECODE_ATT_CUS=128 --weights q4
