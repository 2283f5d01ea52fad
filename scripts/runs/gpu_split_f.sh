# Where the split decode starts to pay: smaller groups, and the mixed-length workload (rows retire, halves merge).
set -o pipefail
O=gpurun_out/split_f
mkdir -p $O
run() {  # name, env, args...
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["frac"])
PY
}
run b1024_cu0 Q3ASR_DECODE_ATT_CUS=0 --batch 1024
run b1024_cu160 Q3ASR_DECODE_ATT_CUS=160 --batch 1024
run b512_cu0 Q3ASR_DECODE_ATT_CUS=0 --batch 512
run b512_cu160 Q3ASR_DECODE_ATT_CUS=160 --batch 512
run len_cu0 Q3ASR_DECODE_ATT_CUS=0 --length-dist u60-150+straggler
run len_cu160 Q3ASR_DECODE_ATT_CUS=160 --length-dist u60-150+straggler
run len_cu160_m1024 "Q3ASR_DECODE_ATT_CUS=160 Q3ASR_DECODE_MERGE_ROWS=1024" --length-dist u60-150+straggler
