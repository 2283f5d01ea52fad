set -o pipefail
timeout -k 10 100 python scripts/bench_gemm256.py || exit 1
for so in build_variants/*.so; do Q3LIB=$so timeout -k 10 100 python scripts/bench_gemm256.py || exit 1; done
timeout -k 10 100 python scripts/bench_gemm256.py
