// Probe for CU-masked streams on MI355X: (1) which (XCC, SE, CU) a mask bit selects, (2) streaming-read bandwidth
// against the number of CUs a stream may use, (3) the same read stream beside an MFMA-bound kernel on the other CUs,
// (4) "chain": the decode loop's event chain between two masked streams, with the streams destroyed and re-created
// under another mask between rounds while the events live on.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/probe_cu_mask.hip -o gpurun_out/probe_cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
template <int NL, int WPC>
__global__ __launch_bounds__(256, WPC) void read_kernel_t(const u32x4* __restrict__ src, size_t n_vec, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (NL - 1) * stride < n_vec; i += NL * stride) {
        u32x4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < NL; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}
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

// The decode attention's request pattern without its arithmetic: per wave and 16-key tile, the K rows as four 16-byte
// loads per lane (16 rows x 64 B per instruction) and the V rows either as sixteen 4-byte loads per lane (one 256 B
// row per instruction, MODE 0) or as four 16-byte loads per lane (four rows per instruction, MODE 1).  Two tiles in flight.
template <int MODE>
__global__ __launch_bounds__(256, 3) void pattern_kernel(const unsigned char* __restrict__ kbase, const unsigned char* __restrict__ vbase,
                                                          int tiles_per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const size_t wave_id = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned char* kp = kbase + wave_id * tiles_per_wave * 4096;
    const unsigned char* vp = vbase + wave_id * tiles_per_wave * 4096;
    u32x4 acc = {0, 0, 0, 0};
    unsigned acc1 = 0;
    u32x4 kf[2][4];
    u32x4 v4[2][4];
    unsigned v1[2][16];
    auto load = [&](int b, int t) {
        const unsigned char* k = kp + (size_t)t * 4096 + li * 256 + lg * 16;
        if (MODE != 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[b][ks] = __builtin_nontemporal_load((const u32x4*)(k + 64 * ks));
        }
        if (MODE == 2) {   // K rows coalesced as well: every 128-byte line is touched by one instruction only
            const unsigned char* k2 = kp + (size_t)t * 4096 + (4 * lg) * 256 + li * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) kf[b][i] = __builtin_nontemporal_load((const u32x4*)(k2 + 256 * i));
        }
        if (MODE == 0) {
            const unsigned char* v = vp + (size_t)t * 4096 + lane * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) v1[b][r] = __builtin_nontemporal_load((const unsigned*)(v + 256 * r));
        } else {
            const unsigned char* v = vp + (size_t)t * 4096 + (4 * lg) * 256 + li * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) v4[b][i] = __builtin_nontemporal_load((const u32x4*)(v + 256 * i));
        }
    };
    auto use = [&](int b) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc ^= kf[b][ks];
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1 ^= v1[b][r];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc ^= v4[b][i];
        }
    };
    load(0, 0);
    for (int t = 0; t + 1 < tiles_per_wave; t += 2) {
        load(1, t + 1);
        use(0);
        if (t + 2 < tiles_per_wave) load(0, t + 2);
        use(1);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ acc1) == 0x12345u) sink[0] = 1;
}

// The same bytes requested as 16-byte-per-lane loads that land in LDS without passing through registers
// (two tiles of K and V per wave = 16 KB, 64 KB per workgroup), then read back from LDS.
__global__ __launch_bounds__(256, 2) void lds_pattern_kernel(const unsigned char* __restrict__ kbase, const unsigned char* __restrict__ vbase,
                                                              int tiles_per_wave, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wave_id = (size_t)blockIdx.x * 4 + wave;
    const unsigned char* kp = kbase + wave_id * tiles_per_wave * 4096 + lane * 16;
    const unsigned char* vp = vbase + wave_id * tiles_per_wave * 4096 + lane * 16;
    unsigned char* mine = lds + wave * 16384;
    u32x4 acc = {0, 0, 0, 0};
    auto load = [&](int b, int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kp + (size_t)t * 4096 + 1024 * i),
                                             (__attribute__((address_space(3))) void*)(mine + b * 8192 + 1024 * i), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vp + (size_t)t * 4096 + 1024 * i),
                                             (__attribute__((address_space(3))) void*)(mine + b * 8192 + 4096 + 1024 * i), 16, 0, 0);
        }
    };
    auto use = [&](int b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= *(const u32x4*)(mine + b * 8192 + 1024 * i + lane * 16);
    };
    load(0, 0);
    for (int t = 0; t + 1 < tiles_per_wave; t += 2) {
        load(1, t + 1);
        __builtin_amdgcn_s_waitcnt(0x0f70 | 8);   // vmcnt(8): the older tile has landed
        use(0);
        if (t + 2 < tiles_per_wave) load(0, t + 2);
        if (t + 2 < tiles_per_wave) __builtin_amdgcn_s_waitcnt(0x0f70 | 8);
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        use(1);
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

__global__ void tick_kernel(int* p, int spin) {
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(20);
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}

#include <csignal>
#include <unistd.h>
static volatile int g_round = -1, g_iter = -1, g_phase = 0;
static void on_alarm(int) {
    char b[128];
    int n = snprintf(b, sizeof b, "STALLED in round %d, iteration %d, phase %d\n", g_round, g_iter, g_phase);
    (void)!write(1, b, n);
    _exit(3);
}

// The decode loop's pattern: a plain stream s0 forks two masked streams A and M; per iteration M records "pre", A
// waits for it, runs, records "attn", M waits for it and runs; every 8th iteration M copies a word to pageable host
// memory and is synchronised; at the end both join s0.  Between rounds A and M are destroyed and re-created with the
// other share while the events (last recorded on the destroyed streams) are kept, as a model did that was asked for
// another share.  keep = 1: the streams of both shares are created once and kept instead.
static int chain_probe(int rounds, int keep) {
    signal(SIGALRM, on_alarm);
    const int layers = 28, shares[2] = {160, 64};
    hipStream_t s0; CK(hipStreamCreate(&s0));
    std::vector<hipEvent_t> ev(2 * layers + 2);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    auto mk = [&](int lo, int hi) { std::vector<uint32_t> m(8, 0); for (int i = lo; i < hi; ++i) m[i >> 5] |= 1u << (i & 31);
                                    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m.data())); return s; };
    hipStream_t kept[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    for (int r = 0; r < rounds; ++r) {
        g_round = r; g_iter = -1; g_phase = 1; alarm(20);
        const int sh = shares[r & 1];
        hipStream_t A, M;
        if (keep) { if (!kept[r & 1][0]) { kept[r & 1][0] = mk(0, sh); kept[r & 1][1] = mk(sh, 256); } A = kept[r & 1][0]; M = kept[r & 1][1]; }
        else { A = mk(0, sh); M = mk(sh, 256); }
        g_phase = 2;
        CK(hipEventRecord(ev[2 * layers], s0)); CK(hipStreamWaitEvent(A, ev[2 * layers], 0)); CK(hipStreamWaitEvent(M, ev[2 * layers], 0));
        int host = 0;
        for (int it = 1; it <= 24; ++it) {
            g_iter = it;
            for (int l = 0; l < layers; ++l) {
                hipLaunchKernelGGL(tick_kernel, dim3(8), dim3(64), 0, M, d + 1, 2); CK(hipEventRecord(ev[l], M));
                CK(hipStreamWaitEvent(A, ev[l], 0)); hipLaunchKernelGGL(tick_kernel, dim3(8), dim3(64), 0, A, d + 2, 2); CK(hipEventRecord(ev[layers + l], A));
                CK(hipStreamWaitEvent(M, ev[layers + l], 0)); hipLaunchKernelGGL(tick_kernel, dim3(8), dim3(64), 0, M, d, 2);
            }
            if (it % 8 == 0) { g_phase = 3; CK(hipMemcpyAsync(&host, d, 4, hipMemcpyDeviceToHost, M)); CK(hipStreamSynchronize(M)); g_phase = 2; }
        }
        g_phase = 4;
        CK(hipEventRecord(ev[2 * layers + 1], M)); CK(hipStreamWaitEvent(s0, ev[2 * layers + 1], 0));
        CK(hipEventRecord(ev[2 * layers + 1], A)); CK(hipStreamWaitEvent(s0, ev[2 * layers + 1], 0));
        CK(hipStreamSynchronize(s0));
        g_phase = 5;
        if (!keep) { CK(hipStreamDestroy(A)); CK(hipStreamDestroy(M)); }
        printf("round %d (share %d, %s): %d ticks\n", r, sh, keep ? "streams kept" : "streams destroyed afterwards", host);
    }
    alarm(0);
    return 0;
}

static hipStream_t masked_stream(const std::vector<int>& bits) {
    std::vector<uint32_t> mask(8, 0);
    for (int b : bits) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t s = nullptr;
    printf("[mask %zu bits] ", bits.size());
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    printf("[stream %p] ", (void*)s);
    return s;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc >= 4 && !strcmp(argv[1], "try")) {   // try <mask words> <bits set>: does stream creation work at all?
        const int words = atoi(argv[2]), nbits = atoi(argv[3]);
        std::vector<uint32_t> mask(words, 0);
        for (int b = 0; b < nbits; ++b) mask[b >> 5] |= 1u << (b & 31);
        int rv = 0, dv = 0;
        CK(hipRuntimeGetVersion(&rv));
        CK(hipDriverGetVersion(&dv));
        printf("runtime %d driver %d: mask of %d words, %d bits ... ", rv, dv, words, nbits);
        hipStream_t s = nullptr;
        CK(hipStreamCreate(&s));
        printf("plain stream ok; ");
        hipStream_t sm = nullptr;
        hipError_t e = hipExtStreamCreateWithCUMask(&sm, (uint32_t)words, mask.data());
        printf("masked stream: %s\n", hipGetErrorString(e));
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "chain")) return chain_probe(atoi(argv[2]), argc >= 4 ? atoi(argv[3]) : 0);   // chain <rounds> [keep]
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
    if (argc >= 4 && !strcmp(argv[1], "where")) {   // where <first bit> <last bit> [step]
        std::vector<int> b;
        const int step = argc >= 5 ? atoi(argv[4]) : 1;
        for (int i = atoi(argv[2]); i <= atoi(argv[3]); i += step) b.push_back(i);
        char label[64];
        snprintf(label, sizeof label, "bits %s-%s step %d", argv[2], argv[3], step);
        where(b, label);
        return 0;
    }

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
    if (argc >= 2 && !strcmp(argv[1], "pattern")) {   // the attention's load pattern against the CU count
        for (int n : {96, 128, 160, 248}) {
            std::vector<int> ba;
            for (int i = 0; i < n; ++i) ba.push_back(i);
            hipStream_t sa = masked_stream(ba);
            printf("\n");
            const int tiles = 28;                       // ~450 keys per (utterance, head) item, as a workgroup of the decode kernel sees
            const int grid = (int)((bytes / 2) / ((size_t)tiles * 4096 * 4));
            auto run = [&](auto kern, const char* label) {
                const unsigned char* kb = (const unsigned char*)src;
                const unsigned char* vb = kb + bytes / 2;
                const int ldsb = (void*)kern == (void*)lds_pattern_kernel ? 65536 : 0;
                if (ldsb) CK(hipFuncSetAttribute((const void*)lds_pattern_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, sa, kb, vb, tiles, sink);
                CK(hipStreamSynchronize(sa));
                CK(hipEventRecord(e0, sa));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, sa, kb, vb, tiles, sink);
                CK(hipEventRecord(e1, sa));
                CK(hipStreamSynchronize(sa));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double moved = (double)grid * 4 * tiles * 8192;
                printf("  %3d CUs, %s: %.2f TB/s = %.1f GB/s per CU\n", n, label, moved / ms * 1e-9, moved / ms * 1e-6 / n);
            };
            run(pattern_kernel<0>, "V as 16 x 4 B per lane ");
            run(pattern_kernel<1>, "V as  4 x 16 B per lane");
            run(pattern_kernel<2>, "K and V as 4 x 16 B per lane, whole rows");
            run(lds_pattern_kernel, "K and V as 16 B per lane straight into LDS (2 workgroups/CU)");
            CK(hipStreamDestroy(sa));
        }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "inflight")) {   // bandwidth against the bytes in flight per CU
        for (int n : {96, 128, 160}) {
            std::vector<int> ba;
            for (int i = 0; i < n; ++i) ba.push_back(i);
            hipStream_t sa = masked_stream(ba);
            printf("\n");
            auto run = [&](auto kern, int nl, int wpc) {
                const int grid = wpc * n;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, sa, src, bytes / 16, sink);
                CK(hipStreamSynchronize(sa));
                CK(hipEventRecord(e0, sa));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, sa, src, bytes / 16, sink);
                CK(hipEventRecord(e1, sa));
                CK(hipStreamSynchronize(sa));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("  %3d CUs, %2d x 16 B per lane, %d workgroups/CU (%3d KB in flight per CU): %.2f TB/s = %.1f GB/s per CU\n", n, nl, wpc,
                       nl * 16 * 256 * wpc / 1024, bytes / ms * 1e-9, bytes / ms * 1e-6 / n);
            };
            run(read_kernel_t<4, 4>, 4, 4);
            run(read_kernel_t<8, 4>, 8, 4);
            run(read_kernel_t<8, 8>, 8, 8);
            run(read_kernel_t<16, 3>, 16, 3);
            run(read_kernel_t<16, 4>, 16, 4);
            run(read_kernel_t<16, 6>, 16, 6);
            run(read_kernel_t<32, 2>, 32, 2);
            run(read_kernel_t<32, 3>, 32, 3);
            run(read_kernel_t<48, 2>, 48, 2);
            CK(hipStreamDestroy(sa));
        }
        return 0;
    }
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
