timeout -k 10 400 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -x -q --timeout 170 --timeout-method thread > gpurun_out/r2_t10.log 2>&1; echo "rc=$?" >> gpurun_out/r2_t10.log; tail -3 gpurun_out/r2_t10.log
grep -q "rc=0" gpurun_out/r2_t10.log || exit 1
timeout -k 10 300 python bench.py --batch 512 --max-tokens 16 --no-cpu-baseline > gpurun_out/r2_enc3_b512.json 2> gpurun_out/r2_enc3_b512.err || exit 1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2_enc3_b512.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["stage_ms_per_step_rank0"])
for k in d["kernels"]: print(k)
PY
