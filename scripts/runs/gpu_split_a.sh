# Split decode: correctness tests, then the headline bench with several attention CU shares (max_tokens 96 for speed).
set -o pipefail
O=gpurun_out/split_a
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_retire.py -m gpu -x -q --timeout 170 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -5 $O/tests.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
for cu in 0 160 144 176 128; do
  Q3ASR_DECODE_ATT_CUS=$cu Q3ASR_VERBOSE=1 timeout -k 10 200 python bench.py --no-cpu-baseline --max-tokens 96 --steps 1 --warmup 1 > $O/bench_t96_cu$cu.json 2> $O/bench_t96_cu$cu.err || { echo "bench cu=$cu failed"; tail -5 $O/bench_t96_cu$cu.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_t96_cu$cu.json").read().strip().splitlines()[-1])
print("att_cus $cu:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
