# Round 3, run j: even group sizes for requests above max_batch: the whole GPU suite, then the ragged bench line.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_j
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python bench.py --no-cpu-baseline --workload ragged --utterances 512 --steps 2 --warmup 1 > $O/bench_ragged_u512.json 2> $O/bench_ragged_u512.err || { tail -5 $O/bench_ragged_u512.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench_ragged_u512.json").read().strip().splitlines()[-1])
print("ragged:", d["value"], d["config"].get("stage_ms_per_step_rank0"))
PY
