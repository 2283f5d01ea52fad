# One stream against the split decode for the other model / weight-format combinations (2048 clips, 96 tokens).
set -o pipefail
O=gpurun_out/split_l
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
run 06b_bf16_cu0 Q3ASR_DECODE_ATT_CUS=0 --preset 0.6b
run 06b_bf16_split Q3ASR_VERBOSE=0 --preset 0.6b
run 17b_q4_cu0 Q3ASR_DECODE_ATT_CUS=0 --weights q4
run 17b_q4_split Q3ASR_VERBOSE=0 --weights q4
run 06b_q4_cu0 Q3ASR_DECODE_ATT_CUS=0 --preset 0.6b --weights q4
run 06b_q4_split Q3ASR_VERBOSE=0 --preset 0.6b --weights q4
