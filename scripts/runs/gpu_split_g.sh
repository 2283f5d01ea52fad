set -o pipefail
O=gpurun_out/split_g
mkdir -p $O
for w in 2 1 3; do
  Q3ASR_DECODE_ATT_CUS=160 Q3ASR_ATTN_WGS_PER_CU=$w timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/bench_w$w.json 2> $O/bench_w$w.err || { echo "bench w=$w failed"; tail -5 $O/bench_w$w.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_w$w.json").read().strip().splitlines()[-1])
print("wgs/cu $w:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k in d["kernels"][:5]: print("   ", k["name"], round(1000*k["ms_per_step"]/k["launches_per_step"],1))
PY
done
