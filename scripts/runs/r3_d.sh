# Round 3, run d: log-mel with the clamp / scale inside the one launch (against the two-launch form), mel parity tests,
# and the c1 line (the reference's test clip, 0.6B, batch 1, GPU beside the CPU restatement).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_d
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_mel.py tests/test_zz_no_alloc.py tests/test_gpu_golden.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $O/tests.log; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  local name=$1 envs=$2; shift 2
  env $envs timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
print("$name:", d["value"], d["ms_per_step"], d["config"].get("stage_ms_per_step_rank0"), [(k["name"], k["ms_per_step"], k["rate"]) for k in (d.get("kernels") or [])[:3]])
PY
}
run bench_mel_b1024 X=1 --workload mel --batch 1024 --steps 20 --warmup 3
run bench_mel_b1024_two_pass Q3ASR_MEL_TWO_PASS=1 --workload mel --batch 1024 --steps 20 --warmup 3
run bench_mel_noise_b1024 X=1 --workload mel --batch 1024 --steps 20 --warmup 3 --audio noise


run bench_mel_gapped_b1024 X=1 --workload mel --batch 1024 --steps 20 --warmup 3 --audio gapped
