# 2048 clips: the projections' table plan against plans measured per candidate, at 448 and at 96 tokens.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_y
rm -rf $O; mkdir -p $O
cd $R
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0")["decode"], (d.get("roofline") or {}).get("achieved"), d["config"].get("decode_attention_cus"))
PY
  grep "decode plan" $O/$name.err | sort -u | head -6
}
run t96_rule "Q3ASR_VERBOSE=1" --no-cpu-baseline --steps 1 --warmup 1 --max-tokens 96
run t96_auto "Q3ASR_VERBOSE=1 Q3ASR_DECODE_AUTOTUNE=1" --no-cpu-baseline --steps 1 --warmup 1 --max-tokens 96
