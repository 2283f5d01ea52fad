# Attention share of the CUs at 2048 clips: 168 and 176 against the default 160.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_w
rm -rf $O; mkdir -p $O
cd $R
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"), d["config"].get("decode_attention_cus"))
PY
}
run att160 X=1 --no-cpu-baseline --steps 1 --warmup 1
run att168 Q3ASR_DECODE_ATT_CUS=168 --no-cpu-baseline --steps 1 --warmup 1
run att176 Q3ASR_DECODE_ATT_CUS=176 --no-cpu-baseline --steps 1 --warmup 1
run att152 Q3ASR_DECODE_ATT_CUS=152 --no-cpu-baseline --steps 1 --warmup 1
