// Bare HBM read loops (plain, deeper, non-temporal, contiguous chunk per block, rows interleaved between blocks)
// timed with HIP events.
//   hipcc -O3 --offload-arch=gfx950 -o hbm_read hbm_read.hip && ./hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ p, size_t n, uint4* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i + (U - 1) * stride < n; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (NT) { const unsigned* q = (const unsigned*)(p + i + j * stride); v[j].x = __builtin_nontemporal_load(q); v[j].y = __builtin_nontemporal_load(q + 1); v[j].z = __builtin_nontemporal_load(q + 2); v[j].w = __builtin_nontemporal_load(q + 3); }
            else v[j] = p[i + j * stride];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
    }
    if (acc.x == 0x9E3779B9u && acc.y == 0x7F4A7C15u && acc.z == 1 && acc.w == 2) *sink = acc;
}
// contiguous chunk per block
template <int U>
__global__ __launch_bounds__(256) void rd_chunk(const uint4* __restrict__ p, size_t n, uint4* sink) {
    const size_t per = n / gridDim.x;
    const uint4* q = p + per * blockIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        uint4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = q[i + j * 256];
#pragma unroll
        for (int j = 0; j < U; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
    }
    if (acc.x == 0x9E3779B9u && acc.y == 0x7F4A7C15u && acc.z == 1 && acc.w == 2) *sink = acc;
}
// Non-temporal 16-byte load.
__device__ __forceinline__ uint4 ntload(const uint4* a) {
    const unsigned* q = (const unsigned*)a;
    return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2),
                      __builtin_nontemporal_load(q + 3));
}
// The decode attention's pattern: every block streams its own contiguous chunk with non-temporal loads.
// ILV = 1: chunk per block.  ILV = 8: eight neighbouring blocks share a region of eight chunks and take every
// eighth 256-byte row of it (a cache laid out [position][head][128] instead of [head][position][128]).
template <int U, int ILV>
__global__ __launch_bounds__(256) void rd_rows(const uint4* __restrict__ p, size_t n, uint4* sink) {
    const size_t per = n / gridDim.x;                       // 16-byte units per block
    const size_t region = (size_t)(blockIdx.x / ILV) * ILV * per;
    const int h = blockIdx.x % ILV;
    uint4 acc = make_uint4(0, 0, 0, 0);
    // 256 threads = 16 rows of 16 units (256 bytes) per step
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    const size_t rows = per / 16;
    for (size_t r = row; r + (U - 1) * 16 < rows; r += U * 16) {
        uint4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = ntload(p + region + ((r + j * 16) * ILV + h) * 16 + col);
#pragma unroll
        for (int j = 0; j < U; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
    }
    if (acc.x == 0x9E3779B9u && acc.y == 0x7F4A7C15u && acc.z == 1 && acc.w == 2) *sink = acc;
}
// The decode attention's load instructions without its arithmetic: a block owns a chunk, its four waves take
// 8 KB tiles round-robin; a tile is four 16-byte-per-lane loads in MFMA fragment order (16 rows of 256 bytes, lane
// (i, g) reads 16 bytes at row i, offset 16 g + 64 ks) and sixteen 4-byte-per-lane loads (one 256-byte row each).
// DEPTH tiles are requested before the first is consumed.  WIDE = true replaces the sixteen narrow loads by four
// 16-byte-per-lane ones.
template <int DEPTH, bool WIDE>
__global__ __launch_bounds__(256) void rd_tiles(const uint4* __restrict__ p, size_t n, uint4* sink) {
    const size_t per = n / gridDim.x;
    const char* base = (const char*)(p + per * blockIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int ntiles = (int)(per * 16 / 8192);
    unsigned acc = 0;
    for (int t = wave; t + 4 * (DEPTH - 1) < ntiles; t += 4 * DEPTH) {
        uint4 k[DEPTH][4];
        unsigned v[DEPTH][16];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* tb = base + (size_t)(t + 4 * d) * 8192;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) k[d][ks] = ntload((const uint4*)(tb + li * 256 + lg * 16 + ks * 64));
            if (WIDE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint4 w = ntload((const uint4*)(tb + 4096 + j * 1024 + lane * 16));
                    v[d][4 * j] = w.x; v[d][4 * j + 1] = w.y; v[d][4 * j + 2] = w.z; v[d][4 * j + 3] = w.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[d][j] = __builtin_nontemporal_load((const unsigned*)(tb + 4096 + j * 256) + lane);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc ^= k[d][ks].x ^ k[d][ks].y ^ k[d][ks].z ^ k[d][ks].w;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc ^= v[d][j];
        }
    }
    if (acc == 0x9E3779B9u) *sink = make_uint4(acc, 0, 0, 0);
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipEventRecord(a); for (int r = 0; r < 5; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    const size_t bytes = 8ull << 30, n = bytes / 16;
    uint4* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        float m1 = timeit([&] { hipLaunchKernelGGL((rd<8, false>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m2 = timeit([&] { hipLaunchKernelGGL((rd<16, false>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m3 = timeit([&] { hipLaunchKernelGGL((rd<8, true>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m4 = timeit([&] { hipLaunchKernelGGL((rd_chunk<8>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m5 = timeit([&] { hipLaunchKernelGGL((rd<4, false>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        printf("blocks %5d: U8 %.0f  U16 %.0f  U8nt %.0f  chunk8 %.0f  U4 %.0f GB/s\n", blocks, bytes / m1 / 1e6, bytes / m2 / 1e6, bytes / m3 / 1e6, bytes / m4 / 1e6, bytes / m5 / 1e6);
    }
    for (int blocks : {16384, 32768}) {   // 8 GiB / 16384 = 512 KB per block, about one (utterance, head) item's K + V
        float m1 = timeit([&] { hipLaunchKernelGGL((rd_rows<8, 1>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m2 = timeit([&] { hipLaunchKernelGGL((rd_rows<8, 8>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m3 = timeit([&] { hipLaunchKernelGGL((rd_rows<4, 1>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m4 = timeit([&] { hipLaunchKernelGGL((rd_rows<4, 8>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        printf("blocks %5d, non-temporal rows: chunk U8 %.0f  interleaved-by-8 U8 %.0f  chunk U4 %.0f  interleaved-by-8 U4 %.0f GB/s\n", blocks,
               bytes / m1 / 1e6, bytes / m2 / 1e6, bytes / m3 / 1e6, bytes / m4 / 1e6);
    }
    for (int blocks : {16384, 32768}) {
        float m1 = timeit([&] { hipLaunchKernelGGL((rd_tiles<1, false>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m2 = timeit([&] { hipLaunchKernelGGL((rd_tiles<2, false>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m3 = timeit([&] { hipLaunchKernelGGL((rd_tiles<1, true>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        float m4 = timeit([&] { hipLaunchKernelGGL((rd_tiles<2, true>), dim3(blocks), dim3(256), 0, 0, d, n, d); });
        printf("blocks %5d, attention tiles: narrow V depth 1 %.0f  depth 2 %.0f  wide V depth 1 %.0f  depth 2 %.0f GB/s\n", blocks,
               bytes / m1 / 1e6, bytes / m2 / 1e6, bytes / m3 / 1e6, bytes / m4 / 1e6);
    }
    // the same with the occupancy of the attention kernel: dynamic LDS caps the blocks per CU (160 KB / 50 KB = 3)
    CK(hipFuncSetAttribute((const void*)rd_tiles<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
    CK(hipFuncSetAttribute((const void*)rd_tiles<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
    for (int lds_kb : {50, 38, 30}) {
        const int blocks = 16384;
        float m1 = timeit([&] { hipLaunchKernelGGL((rd_tiles<1, false>), dim3(blocks), dim3(256), lds_kb << 10, 0, d, n, d); });
        float m2 = timeit([&] { hipLaunchKernelGGL((rd_tiles<2, false>), dim3(blocks), dim3(256), lds_kb << 10, 0, d, n, d); });
        printf("attention tiles, %d blocks per CU: depth 1 %.0f  depth 2 %.0f GB/s\n", 160 / lds_kb, bytes / m1 / 1e6, bytes / m2 / 1e6);
    }
    CK(hipGetLastError());
    return 0;
}
