# Attention reductions on v_permlane16_swap: tests that cover the attention kernels, then the headline bench.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_o
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_b2048.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench_b2048.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("stage_ms_per_step_rank0"), d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])
PY
