# Round 3, run k: what the decode attention and the projections reach on their shares of the CUs with the other share idle
# (Q3ASR_SPLIT_SERIAL=1), against the default side-by-side run; 96 tokens (short contexts) and 448.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_k
mkdir -p $O
cd $R
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), [(k["name"][:24], k["launches_per_step"], k["ms_per_step"], k["rate"]) for k in d["kernels"][:6]])
PY
}
run serial448 "Q3ASR_SPLIT_SERIAL=1 Q3ASR_ATT_STREAMS=1" --steps 1 --warmup 1 --profile-stride 5
run default448 Q3ASR_ATT_STREAMS=1 --steps 1 --warmup 1 --profile-stride 5
