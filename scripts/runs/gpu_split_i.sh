# PV on MFMA: correctness, then the headline single-stream and split.
set -o pipefail
O=gpurun_out/split_i
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -8 $O/tests.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
for cu in 0 160 128; do
  Q3ASR_DECODE_ATT_CUS=$cu timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/bench_cu$cu.json 2> $O/bench_cu$cu.err || { echo "bench cu=$cu failed"; tail -5 $O/bench_cu$cu.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_cu$cu.json").read().strip().splitlines()[-1])
print("att_cus $cu:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
