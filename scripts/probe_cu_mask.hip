// Probe for CU-masked streams on MI355X: (1) which (XCC, SE, CU) a mask bit selects, (2) streaming-read bandwidth
// against the number of CUs a stream may use, (3) the same read stream beside an MFMA-bound kernel on the other CUs.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/probe_cu_mask.hip -o gpurun_out/probe_cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include <map>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void where_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID, bits 0..3
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    // stay a little so that the blocks spread over every CU the queue may use
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}

// 256 threads, each wave keeps 16 x 1 KB requests in flight (like the decode attention's two tiles)
__global__ __launch_bounds__(256, 3) void read_kernel(const u32x4* __restrict__ src, size_t n_vec, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + 15 * stride < n_vec; i += 16 * stride) {
        u32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

// MFMA-bound filler: 8 waves per workgroup, registers only
__global__ __launch_bounds__(512) void mfma_kernel(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    if (s == 12345.f) out[0] = s;
}

static hipStream_t masked_stream(const std::vector<int>& bits) {
    std::vector<uint32_t> mask(8, 0);
    for (int b : bits) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, mask.data()));
    return s;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    unsigned* d_where;
    CK(hipMalloc(&d_where, 4096 * 8));
    // (1) what a mask bit selects
    auto where = [&](const std::vector<int>& bits, const char* label) {
        hipStream_t s = masked_stream(bits);
        CK(hipMemsetAsync(d_where, 0xff, 4096 * 8, s));
        hipLaunchKernelGGL(where_kernel, dim3(2048), dim3(64), 0, s, d_where);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned> h(4096);
        CK(hipMemcpy(h.data(), d_where, 4096 * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int i = 0; i < 2048; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1];
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[xcc].insert(se * 100 + sh * 16 + cu);
        }
        printf("%s:", label);
        size_t total = 0;
        for (auto& kv : per_xcc) {
            printf(" xcc%u[%zu:", kv.first, kv.second.size());
            total += kv.second.size();
            int shown = 0;
            for (unsigned v : kv.second) if (shown++ < 6) printf(" %u", v);
            printf("]");
        }
        printf(" -> %zu distinct CUs\n", total);
        CK(hipStreamDestroy(s));
    };
    where({0}, "bit 0");
    where({1}, "bit 1");
    where({8}, "bit 8");
    where({0, 1, 2, 3, 4, 5, 6, 7}, "bits 0-7");
    { std::vector<int> b; for (int i = 0; i < 96; ++i) b.push_back(i); where(b, "bits 0-95"); }
    { std::vector<int> b; for (int i = 0; i < cus; ++i) b.push_back(i); where(b, "all bits"); }

    // (2) read bandwidth against the CU count (bits 0 .. n-1), alone
    const size_t bytes = (size_t)8 << 30;
    u32x4* src;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    unsigned* sink;
    CK(hipMalloc(&sink, 64));
    float* fout;
    CK(hipMalloc(&fout, 64));
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    const int ns[] = {32, 64, 96, 128, 160, 192, 256};
    for (int n : ns) {
        if (n > cus) continue;
        std::vector<int> ba, bb;
        for (int i = 0; i < cus; ++i) (i < n ? ba : bb).push_back(i);
        hipStream_t sa = masked_stream(ba);
        const int grid = 3 * n;
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, sa, src, bytes / 16, sink);
        CK(hipStreamSynchronize(sa));
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, sa, src, bytes / 16, sink);
        CK(hipEventRecord(e1, sa));
        CK(hipStreamSynchronize(sa));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("read on %3d CUs alone: %.2f TB/s", n, bytes / ms * 1e-9);
        if (!bb.empty()) {
            // (3) beside an MFMA-bound kernel on the other CUs
            hipStream_t sb = masked_stream(bb);
            const int mg = (int)bb.size();
            const int iters = 200000;
            hipLaunchKernelGGL(mfma_kernel, dim3(mg), dim3(512), 0, sb, fout, 1000);
            CK(hipStreamSynchronize(sb));
            CK(hipEventRecord(f0, sb));
            hipLaunchKernelGGL(mfma_kernel, dim3(mg), dim3(512), 0, sb, fout, iters);
            CK(hipEventRecord(f1, sb));
            CK(hipStreamSynchronize(sb));
            float ms_alone;
            CK(hipEventElapsedTime(&ms_alone, f0, f1));
            CK(hipEventRecord(f0, sb));
            hipLaunchKernelGGL(mfma_kernel, dim3(mg), dim3(512), 0, sb, fout, iters);
            CK(hipEventRecord(f1, sb));
            CK(hipEventRecord(e0, sa));
            for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, sa, src, bytes / 16, sink);
            CK(hipEventRecord(e1, sa));
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
            float ms_r, ms_m;
            CK(hipEventElapsedTime(&ms_r, e0, e1));
            CK(hipEventElapsedTime(&ms_m, f0, f1));
            const double fl = 2.0 * 16 * 16 * 32 * 8 * 8 * (double)iters * mg;
            printf(" | beside MFMA on %3d CUs: read %.2f TB/s, MFMA %.0f TF/s (alone %.0f)", mg, 4.0 * bytes / ms_r * 1e-9, fl / ms_m * 1e-9,
                   fl / ms_alone * 1e-9);
            CK(hipStreamDestroy(sb));
        }
        printf("\n");
        CK(hipStreamDestroy(sa));
    }
    return 0;
}
