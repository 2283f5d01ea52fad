# Log-mel stage: tests, then the stand-alone stage.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_l
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_mel.py tests/test_zz_no_alloc.py tests/test_gpu_golden.py -x -q --timeout 120 --timeout-method thread > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
for a in speechlike gapped; do for i in 1 2; do
timeout -k 10 200 python bench.py --workload mel --batch 1024 --audio $a --steps 20 --warmup 3 > $O/mel_b1024_${a}_$i.json 2> $O/mel.err || { tail -5 $O/mel.err; exit 1; }
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/mel_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
