# Headline workload (448 tokens) single stream against the split decode with measured plans.
set -o pipefail
O=gpurun_out/split_e
mkdir -p $O
for cu in 0 160 144 128; do
  Q3ASR_DECODE_ATT_CUS=$cu timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/bench_cu$cu.json 2> $O/bench_cu$cu.err || { echo "bench cu=$cu failed"; tail -5 $O/bench_cu$cu.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_cu$cu.json").read().strip().splitlines()[-1])
print("att_cus $cu:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
