set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r2_t2.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r2_t2.log; tail -5 gpurun_out/r2_t2.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -u -m pytest tests/test_gpu_golden.py -q -s --timeout 180 --timeout-method thread > gpurun_out/r2_golden.log 2>&1 || exit 1
for cfg in "1.7b bf16" "1.7b q4" "0.6b bf16" "0.6b q4"; do set -- $cfg; timeout -k 10 300 python scripts/plan_table.py --preset $1 --weights $2 > gpurun_out/plan2_$1_$2.json 2> gpurun_out/plan2_$1_$2.err || exit 1; done
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --backend gloo --batch 8 --max-tokens 8 --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/r2_launcher.json 2> gpurun_out/r2_launcher.err || exit 1
timeout -k 10 400 python bench.py --batch 2048 --length-dist u60-150+straggler --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_len_retire.json 2> gpurun_out/r2_len_retire.err || exit 1
Q3ASR_RETIRE=0 timeout -k 10 400 python bench.py --batch 2048 --length-dist u60-150+straggler --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_len_noretire.json 2> gpurun_out/r2_len_noretire.err || exit 1
echo done
