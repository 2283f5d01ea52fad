# Measurement set, part D: the whole GPU test suite on the final tree, then one counter pass for the matrix-pipe
# utilisation of every kernel (512 clips, 16 tokens).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_d
mkdir -p $O
cd $R
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/tests.log; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof/mfma -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --batch 512 --max-tokens 16 --steps 1 --warmup 0 > $O/mfma_bench.json 2> $O/mfma.err; rc=$?
echo "pmc rc=$rc"
[ $rc -eq 0 ] || { tail -3 $O/mfma.err; exit 1; }
F=$(find /tmp/prof/mfma -name "*counter_collection.csv" | head -1)
python3 $R/scripts/pmc_mfma_util.py $F "1.7b, 512 clips x 30 s, max_tokens 16 (encoder + prefill + 15 decode steps)" > $O/pmc_mfma_util_b512.json
python3 - <<PY
import json
d=json.load(open("$O/pmc_mfma_util_b512.json"))
for k in d["kernels"][:14]: print(k["kernel"], k["launches"], k["mfma_util"])
PY
