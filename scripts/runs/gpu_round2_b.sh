set -o pipefail
timeout -k 10 900 python bench.py > gpurun_out/r2_bench_b2048.json 2> gpurun_out/r2_bench_b2048.err || exit 1
timeout -k 10 400 python bench.py --max-tokens 96 --no-cpu-baseline > gpurun_out/r2_bench_b2048_t96.json 2> gpurun_out/r2_bench_b2048_t96.err || exit 1
echo done
