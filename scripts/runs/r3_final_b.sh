# Round 3 evidence, part B: rocprofv3 kernel stats of the default command with both halves' attention on ONE stream
# (launches do not overlap: the per-launch average that the bench's sampled, exclusive launches are compared with),
# then the other bench lines of the round.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_final_b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
export Q3ASR_ATT_STREAMS=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_b2048_att_streams1_under_rocprof.json 2> $O/trace.err; rc=$?
unset Q3ASR_ATT_STREAMS
python3 $R/scripts/kernel_stats.py /tmp/prof/trace > $O/bench_b2048_att_streams1_kernel_stats.csv
head -3 $O/bench_b2048_att_streams1_kernel_stats.csv | cut -c1-160
echo "rocprofv3 kernel-trace rc=$rc"
[ $rc -eq 0 ] || { tail -3 $O/trace.err; exit 1; }
rm -rf /tmp/prof
cd $R
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["config"].get("stage_ms_per_step_rank0"), (d.get("roofline") or {}).get("achieved"))
PY
}
run bench_b2048_tokens96 X=1 --no-cpu-baseline --steps 3 --warmup 1 --max-tokens 96
run bench_b2048_len_u60-150_straggler_retire X=1 --no-cpu-baseline --steps 3 --warmup 1 --length-dist u60-150+straggler
run bench_b2048_one_stream Q3ASR_DECODE_ATT_CUS=0 --no-cpu-baseline --steps 2 --warmup 1
run bench_0.6b_b128 X=1 --no-cpu-baseline --preset 0.6b --batch 128 --steps 3 --warmup 1
run bench_0.6b_q4_b256 X=1 --no-cpu-baseline --preset 0.6b --weights q4 --batch 256 --steps 3 --warmup 1
run bench_ragged_u512 X=1 --no-cpu-baseline --workload ragged --utterances 512 --steps 2 --warmup 1
run bench_c1 X=1 --workload c1 --steps 3 --warmup 1
run bench_b2048_steps20 X=1 --no-cpu-baseline --steps 20 --warmup 5
