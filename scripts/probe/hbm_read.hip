// Bare HBM read loops (plain, deeper, non-temporal, contiguous chunk per block) timed with HIP events.
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
    CK(hipGetLastError());
    return 0;
}
