# build_variant.sh NAME -DFLAG...: a build of libq3asr.so whose gemm256.hip is compiled with extra defines -> build_variants/libq3asr_NAME.so
set -e
cd "$(dirname "$0")/../qwen3-asr-swift_amd"
name=$1; shift
mkdir -p ../build_variants build
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c csrc/gemm256.hip -o build/gemm256_$name.o
objs=$(ls build/*.o | grep -v gemm256)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../build_variants/libq3asr_$name.so $objs build/gemm256_$name.o
echo built build_variants/libq3asr_$name.so
