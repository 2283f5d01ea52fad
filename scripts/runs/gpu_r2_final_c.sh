# Measurement set, part C: the full-size tests (incl. the headline configuration), the default bench line with its CPU
# baseline, then the labelled variants and the other configurations.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_c
mkdir -p $O
cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_retire.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("traffic"))
PY
}
run bench_b2048 X=1
run bench_b2048_tokens96 X=1 --max-tokens 96 --no-cpu-baseline
run bench_b2048_len_u60-150_straggler_retire X=1 --length-dist u60-150+straggler --no-cpu-baseline
run bench_b2048_len_u60-150_straggler_noretire Q3ASR_RETIRE=0 --length-dist u60-150+straggler --no-cpu-baseline
run bench_b2048_one_stream Q3ASR_DECODE_ATT_CUS=0 --no-cpu-baseline
run bench_b2048_lm_head_whole Q3ASR_LM_SPLIT=0 --no-cpu-baseline
run bench_ragged_u512 X=1 --workload ragged --utterances 512 --no-cpu-baseline
run bench_0.6b_b128 X=1 --preset 0.6b --batch 128 --no-cpu-baseline
run bench_0.6b_q4_b256 X=1 --preset 0.6b --weights q4 --batch 256 --no-cpu-baseline
run bench_mel_b1024 X=1 --workload mel --batch 1024 --steps 10 --warmup 3
echo benches done
