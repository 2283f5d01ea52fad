# Short outputs are bound by the projections' stream: 96 tokens per clip with 128 + 128 CUs against 160 + 96.
set -o pipefail
O=gpurun_out/r2s_h
mkdir -p $O
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])
PY
}
run t96_att128 Q3ASR_DECODE_ATT_CUS=128 --max-tokens 96
run t96_att160 X=1 --max-tokens 96
run t96_att144 Q3ASR_DECODE_ATT_CUS=144 --max-tokens 96
