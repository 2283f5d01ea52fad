# Full GPU suite and the default bench line with the split decode as the library's own choice.
set -o pipefail
O=gpurun_out/split_h
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -5 $O/tests.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_b2048.json 2> $O/bench_b2048.err || { tail -5 $O/bench_b2048.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench_b2048.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["stage_ms_per_step_rank0"], d["config"]["decode_attention_cus"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
