# Encoder passes trimmed to whole GEMM rounds: encoder / full-size tests, then the encoder stage time at 2048 clips.
set -o pipefail
O=gpurun_out/r2s_c
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --max-tokens 4 > $O/enc.json 2> $O/enc.err || { tail -5 $O/enc.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/enc.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["stage_ms_per_step_rank0"], [(k["name"],k["ms_per_step"],k["rate"]) for k in d["kernels"][:12]])
PY
