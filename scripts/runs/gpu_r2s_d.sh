# Share of the compute units between the attention stream and the projections' stream, now that the LM head is shared.
set -o pipefail
O=gpurun_out/r2s_d
mkdir -p $O
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["achieved"], [ (k["name"],k["ms_per_step"]) for k in d["kernels"][:6]])
PY
}
run att176 Q3ASR_DECODE_ATT_CUS=176
run att168 Q3ASR_DECODE_ATT_CUS=168
env Q3ASR_DECODE_ATT_CUS=0 timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_b2048_one_stream.json 2> $O/one.err || exit 1
