# Measurement set on the final build, remaining lines: LM head whole, one stream, scripted lengths without retirement,
# then the other configurations.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_g
mkdir -p $O
cd $R
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"))
PY
}
run bench_b2048_one_stream Q3ASR_DECODE_ATT_CUS=0
run bench_b2048_lm_head_whole Q3ASR_LM_SPLIT=0
run bench_b2048_len_u60-150_straggler_noretire Q3ASR_RETIRE=0 --length-dist u60-150+straggler
run bench_0.6b_b128 X=1 --preset 0.6b --batch 128
run bench_0.6b_q4_b256 X=1 --preset 0.6b --weights q4 --batch 256
run bench_ragged_u512 X=1 --workload ragged --utterances 512
echo done
