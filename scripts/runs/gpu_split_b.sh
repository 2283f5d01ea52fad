# Attention alone on a stream confined to n CUs (no concurrent kernels): per-CU capability of the kernel.
set -o pipefail
O=gpurun_out/split_b
mkdir -p $O
for cu in 248 192 160 128 96; do
  Q3ASR_DECODE_ATT_CUS=0 Q3ASR_DECODE_ATT_ONLY_CUS=$cu timeout -k 10 200 python bench.py --no-cpu-baseline --max-tokens 48 --steps 1 --warmup 1 > $O/bench_only$cu.json 2> $O/bench_only$cu.err || { echo "bench cu=$cu failed"; tail -5 $O/bench_only$cu.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_only$cu.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("attention alone on $cu CUs:", d["config"]["stage_ms_per_step_rank0"]["decode"], "ms decode, frac", r["frac"], "avg ms", r["avg_launch_ms"], "=> %.1f GB/s per CU" % (r["achieved"] / $cu))
PY
done
