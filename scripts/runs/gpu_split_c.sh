# K rows through LDS: correctness, then single-stream and split decode at several attention CU shares.
set -o pipefail
O=gpurun_out/split_c
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_retire.py tests/test_gpu_golden.py tests/test_gpu_ops.py -m gpu -x -q --timeout 170 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -5 $O/tests.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
for cu in 0 160 144 128; do
  Q3ASR_DECODE_ATT_CUS=$cu timeout -k 10 200 python bench.py --no-cpu-baseline --max-tokens 96 --steps 1 --warmup 1 > $O/bench_t96_cu$cu.json 2> $O/bench_t96_cu$cu.err || { echo "bench cu=$cu failed"; tail -5 $O/bench_t96_cu$cu.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_t96_cu$cu.json").read().strip().splitlines()[-1])
print("att_cus $cu:", d["value"], d["config"]["stage_ms_per_step_rank0"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
