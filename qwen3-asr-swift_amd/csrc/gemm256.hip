// 256 x 256 x 64 bf16 MFMA GEMM for gfx950 (large row counts):  C[M,N] = A[M,K] * W[N,K]^T.
//
// One 512-thread workgroup (8 waves as 2 x 4) per tile, one workgroup per CU.  LDS holds two K tiles;
// each K tile is four 16 KB staging units (A0, A1, W0, W1: 128 rows x 64 k, XOR-swizzled 16-byte
// slots) filled by 16-byte global_load_lds.  The K loop runs 8 phases per two K tiles; a phase
// reads one register sub-tile from LDS, issues the loads of one staging unit, and runs the 16 MFMAs
// of one 64 x 32 quadrant of the wave's 128 x 64 output between two raw barriers:
//
//   phase      LDS reads (this K tile)     quadrant (A unit, W unit)     staging unit issued
//   0 / 4      W0 (4), then A0 (8)         (0, 0)                        A1 of the other buffer
//   1 / 5      W1 (4)                      (0, 1)                        W0 of this buffer, next tile
//   2 / 6      A1 (8)                      (1, 1)                        A0 of this buffer, next tile
//   3 / 7      --  (W0 still in VGPRs)     (1, 0)                        W1 of this buffer, next tile
//
// A unit is re-staged two phases after its last read (W0: one phase, its reads are retired by a
// counted lgkmcnt before the phase's first barrier), loads are waited for with a counted vmcnt once
// per K tile (three units stay in flight across the barriers) and a unit is first read one phase
// after that wait.  Waves 4-7 run one barrier behind waves 0-3, so one half of the workgroup reads
// LDS and issues loads while the other half is in its MFMA cluster.
//
// Weight fragments are the MFMA "A" operand and activation fragments the "B" operand, so an
// accumulator register holds consecutive output columns of one row; weight rows are assigned to
// MFMA rows so that a lane owns 16 contiguous columns (two staging units x two fragments x 4).
#include "gemm.h"
#include "prof.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

namespace {

constexpr int UNIT = 128 * 128;  // bytes of one staging unit
constexpr int BUF = 4 * UNIT;    // A0 | A1 | W0 | W1
constexpr int LDS_BYTES = 2 * BUF;

// AMODE A_CONV3X3S2: implicit-GEMM 3x3 stride-2 pad-1 convolution over an NHWC image.  K runs over 9 taps x Cin
// channels, each tap padded to whole 64-channel K tiles (for Cin = 480: 8 tiles, the last one half empty, +6.7 % MFMA
// work) so that a K tile never straddles two taps: the tap's pixel offset and the channel offset are then uniform and
// ride in the scalar offset of the buffer loads, a lane's own patch origin is a fixed per-lane offset exactly like a
// plain row, and padding or channel-tail lanes are sent out of the descriptor's range (they read as zero).
template <int OUT, int AMODE>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = uwave >> 2, wc = uwave & 3;
    const int li = lane & 15, lg = lane >> 4;

    // ---- workgroup -> tiles.  The grid is persistent (launch_o caps it at the CU count, a multiple of 8): this
    // workgroup computes the tiles of the virtual block indices blockIdx.x, blockIdx.x + gridDim.x, ...  A virtual
    // index maps to a tile by the XCD-contiguous remap (virtual index & 7 is the XCD the workgroup runs on), then
    // groups of 8 M-tiles share each N-tile. ----
    const int tiles_m = (p.M + 255) >> 8, tiles_n = (p.N + 255) >> 8, total = tiles_m * tiles_n;
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;

    constexpr bool CONV = AMODE == A_CONV3X3S2;
    const int tpt = CONV ? (p.Cin + 63) >> 6 : 1;            // K tiles per tap
    const int tpt_magic = CONV ? (65536 + tpt - 1) / tpt : 0;  // kt / tpt == (kt * magic) >> 16 for kt < 4096
    const int nk_total = CONV ? 9 * tpt : (p.K + 63) >> 6;
    const int per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt0 = blockIdx.z * per, kt1 = min(nk_total, kt0 + per);

    // ---- staging sources.  Thread t fills 16-byte slots t and 512 + t of every unit: LDS row
    // q*64 + (t >> 3), slot t & 7, which holds source k-chunk (t & 7) ^ ((row >> 1) & 7). ----
    const int srow = tid >> 3;
    const int koff = ((tid & 7) ^ ((srow >> 1) & 7)) * 8;
    // Both operands are read through buffer descriptors based at the tile's first row: a load needs a fixed
    // 32-bit per-lane byte offset plus a scalar K offset, and no vector ALU work inside the K loop.
    // Offsets at or beyond BUF_RANGE read as zero (K tail, and the tiles past the end that keep the load counts
    // uniform).
    constexpr int BUF_RANGE = 0x40000000, BUF_OOB = 0x7f000000;
    int m0 = 0, n0 = 0, tn = 0;
    int a_off[2][2], w_off[2][2];  // [q][unit], in bytes
    unsigned a_valid[2][2];        // CONV: bit t set when tap t of the lane's output position lies inside the image
    __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
    auto setup = [&](int b) {
        const int q8 = total >> 3, r8 = total & 7, x = b & 7;
        const int L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (b >> 3);
        const int grp = L / per_group, first_m = grp * GROUP;
        const int gsz = min(GROUP, tiles_m - first_m);
        const int tm = first_m + (L % per_group) % gsz;
        tn = (L % per_group) / gsz;
        m0 = tm << 8;
        n0 = tn << 8;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // A unit u, LDS row q*64 + srow  <-  tile row q*128 + u*64 + srow   (q = reading wave's row half)
                if constexpr (CONV) {
                    const int hw = p.Hout * p.Wout, img = p.Hin * p.Win * p.Cin;
                    const int chunk0 = m0 / hw;                       // the descriptor is based one row and one pixel before this chunk
                    const int mm = min(m0 + q * 128 + u * 64 + srow, p.M - 1);
                    const int chunk = mm / hw, rem = mm - chunk * hw;
                    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                    const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
                    const int wv = min(p.Win, p.win_valid ? p.win_valid[chunk] : p.Win);
                    a_off[q][u] = ((chunk - chunk0) * img + ((iy0 + 1) * p.Win + ix0 + 1) * p.Cin + koff) * 2;
                    unsigned valid = 0;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                        if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < wv) valid |= 1u << t;
                    }
                    a_valid[q][u] = valid;
                } else {
                    const int m = min(m0 + q * 128 + u * 64 + srow, p.M - 1) - m0;
                    a_off[q][u] = (m * (int)p.lda + koff) * 2;
                }
                // W unit u, LDS row rho = q*64 + srow = wc*32 + j*16 + i  <-  tile row wc*64 + (i>>2)*16 + u*8 + j*4 + (i&3)
                const int rho = q * 64 + srow, i = rho & 15, j = (rho >> 4) & 1, c = rho >> 5;
                const int n = min(n0 + c * 64 + (i >> 2) * 16 + u * 8 + j * 4 + (i & 3), p.N - 1) - n0;
                w_off[q][u] = (n * p.K + koff) * 2;
            }
        if constexpr (CONV)
            a_rsrc = uniform_buffer(p.A + (int64_t)(m0 / (p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin - (int64_t)(p.Win + 1) * p.Cin, BUF_RANGE);
        else
            a_rsrc = uniform_buffer(p.A + (int64_t)m0 * p.lda, BUF_RANGE);
        w_rsrc = uniform_buffer((const bf16_t*)p.W + (int64_t)n0 * p.K, BUF_RANGE);
    };
    const int lds_lane0 = uwave * 64 * 16;

    // unit: 0 = A0, 1 = A1, 2 = W0, 3 = W1
    auto stage = [&](int buf, int unit, int kt) {
        char* dst = smem + buf * BUF + unit * UNIT + lds_lane0;
        const bool past = kt >= kt1;  // wave-uniform
        if constexpr (CONV) {
            const int tap = (kt * tpt_magic) >> 16, ci0 = (kt - tap * tpt) * 64;   // wave-uniform
            const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;
            const bool ch_out = ci0 + koff >= p.Cin;                                // channel tail of the tap's last K tile
            const int a_s = past ? 0 : (((kh - 1) * p.Win + kw - 1) * p.Cin + ci0 + (p.Win + 1) * p.Cin) * 2;
            const int w_s = past ? 0 : (tap * p.Cin + ci0) * 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (unit < 2) {
                    const bool ok = !past && !ch_out && ((a_valid[q][unit] >> tap) & 1u);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, LDS_PTR(dst + q * 512 * 16), 16, ok ? a_off[q][unit] : BUF_OOB, a_s, 0, 0);
                } else {
                    const bool ok = !past && !ch_out;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(dst + q * 512 * 16), 16, ok ? w_off[q][unit - 2] : BUF_OOB, w_s, 0, 0);
                }
            }
            return;
        }
        const int kbase = kt * 64;
        const bool tail = kbase + 64 > p.K;
        const bool masked = past || (tail && kbase + koff >= p.K);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int v = masked ? BUF_OOB : (unit < 2 ? a_off[q][unit] : w_off[q][unit - 2]);
            if (unit < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, LDS_PTR(dst + q * 512 * 16), 16, v, past ? 0 : kbase * 2, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(dst + q * 512 * 16), 16, v, past ? 0 : kbase * 2, 0, 0);
        }
    };

    // ---- fragment read offsets within a unit (ks = 1 is the same address with bit 6 flipped) ----
    const int sw = (li >> 1) & 7;
    const int a_rd = (wr * 64 + li) * 128 + ((lg ^ (sw & 3)) << 4) + ((sw >> 2) << 6);
    const int w_rd = (wc * 32 + li) * 128 + ((lg ^ (sw & 3)) << 4) + ((sw >> 2) << 6);

    f32x4 acc[2][4][2][2];  // [A unit][fragment][W unit][fragment]

    const int nt = kt1 - kt0;
    // K tile 0 of a tile complete in buffer 0, three units of its K tile 1 in flight
    auto prefetch = [&]() {
        stage(0, 2, kt0); stage(0, 0, kt0); stage(0, 3, kt0); stage(0, 1, kt0);
        stage(1, 2, kt0 + 1); stage(1, 0, kt0 + 1); stage(1, 3, kt0 + 1);
    };
    int vb = blockIdx.x;
    setup(vb);
    if (nt > 0) prefetch();
    for (; vb < total; vb += gridDim.x) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][f][u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (nt > 0) {
        // everything outstanding is waited for: the loads requested ahead of the previous tile's epilogue and that
        // epilogue's own loads and stores (loads and stores do not retire in one order, so no counted wait here)
        if (vb == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // first tile: only its own loads are out
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();

        bf16x8 af[4][2], w0f[2][2], w1f[2][2];
        const int iters = (nt + 1) >> 1;
        for (int it = 0; it < iters; ++it) {
            const int te = kt0 + 2 * it;
            bool tail = false;
#pragma unroll
            for (int ph = 0; ph < 8; ++ph) {
                const int buf = ph >> 2, sub = ph & 3;
                if (ph == 4 && te + 1 >= kt1) { tail = true; break; }
                const char* sb = smem + buf * BUF;
                if (sub == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) w0f[j][ks] = *(const bf16x8*)(sb + 2 * UNIT + ((w_rd + j * 2048) ^ (ks << 6)));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) af[f][ks] = *(const bf16x8*)(sb + ((a_rd + f * 2048) ^ (ks << 6)));
                } else if (sub == 1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) w1f[j][ks] = *(const bf16x8*)(sb + 3 * UNIT + ((w_rd + j * 2048) ^ (ks << 6)));
                } else if (sub == 2) {
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) af[f][ks] = *(const bf16x8*)(sb + UNIT + ((a_rd + f * 2048) ^ (ks << 6)));
                }
                switch (ph) {
                    case 0: stage(1, 1, te + 1); break;
                    case 1: stage(0, 2, te + 2); break;
                    case 2: stage(0, 0, te + 2); break;
                    case 3: stage(0, 3, te + 2); break;
                    case 4: stage(0, 1, te + 2); break;
                    case 5: stage(1, 2, te + 3); break;
                    case 6: stage(1, 0, te + 3); break;
                    default: stage(1, 3, te + 3); break;
                }
                if (sub == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                if (sub == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_setprio(1);
                const int ua = sub >> 1, uw = (sub == 1 || sub == 2) ? 1 : 0;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[ua][f][uw][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uw ? w1f[j][ks] : w0f[j][ks], af[f][ks],
                                                                                       acc[ua][f][uw][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
            }
            if (tail) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wr == 0) __builtin_amdgcn_s_barrier();
    }

    // The next tile's first loads are requested before this tile's epilogue, which then runs under them (one
    // workgroup fills the CU, so nothing else would cover the epilogue's memory traffic).
    const int e_m0 = m0, e_n0 = n0, e_tn = tn;
    if (vb + (int)gridDim.x < total) {
        setup(vb + gridDim.x);
        if (nt > 0) prefetch();
    }

    // ---- epilogue: lane (li, lg) owns rows e_m0 + wr*128 + a*64 + f*16 + li, columns c0 .. c0 + 15 ----
    const int c0 = e_n0 + wc * 64 + lg * 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int m = e_m0 + wr * 128 + a * 64 + f * 16 + li;
            const bool row_ok = m < p.M;
            int64_t orow = m;
            if (row_ok && p.rowmap) orow = p.rowmap[m];
            float v[16];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[u * 8 + j * 4 + r] = acc[a][f][u][j][r];
            if constexpr (OUT == OUT_ARGMAX) {
                float best = -INFINITY;
                int bidx = 0x7fffffff;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (c0 + e < p.N && v[e] > best) { best = v[e]; bidx = c0 + e + p.col_base; }
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float ov = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bidx, o, 64);
                    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
                }
                if (row_ok && lg == 0) {
                    const int part = p.part_base + e_tn * 4 + wc;
                    p.part_val[(int64_t)m * p.nparts + part] = best;
                    p.part_idx[(int64_t)m * p.nparts + part] = bidx;
                }
                continue;
            }
            if (!row_ok || orow < 0) continue;
            if constexpr (OUT == OUT_SLAB) {
                float* o = (float*)p.out + blockIdx.z * p.slab_stride + orow * p.ldo + c0;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (c0 + b * 4 < p.N) *(float4*)(o + b * 4) = make_float4(v[b * 4], v[b * 4 + 1], v[b * 4 + 2], v[b * 4 + 3]);
                continue;
            }
            if constexpr (OUT == OUT_SWIGLU) {
                // weight rows come in blocks of 16: 8 gate rows then the 8 matching up rows
                if (c0 < p.N) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float g = v[e], up = v[8 + e];
                        y[e] = g / (1.0f + __expf(-g)) * up;
                    }
                    bf16_t* o = (bf16_t*)p.out + orow * p.ldo + (c0 >> 1);
                    *(uint4*)o = make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]),
                                            pack_bf16x2(y[6], y[7]));
                }
                continue;
            }
            const float* rt = p.rowtab ? p.rowtab + (int64_t)(m % p.rowtab_period) * p.N : nullptr;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = c0 + b * 4;
                if (c >= p.N) continue;
                float x[4] = {v[b * 4], v[b * 4 + 1], v[b * 4 + 2], v[b * 4 + 3]};
                if (p.bias) {
                    const float4 bb = *(const float4*)(p.bias + c);
                    x[0] += bb.x; x[1] += bb.y; x[2] += bb.z; x[3] += bb.w;
                }
                if (rt) {
                    const float4 tt = *(const float4*)(rt + c);
                    x[0] += tt.x; x[1] += tt.y; x[2] += tt.z; x[3] += tt.w;
                }
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = gelu_erf(x[r]);
                }
                if constexpr (OUT == OUT_BF16) {
                    bf16_t* o = (bf16_t*)p.out + orow * p.ldo + c;
                    *(uint2*)o = make_uint2(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]));
                } else if constexpr (OUT == OUT_F32) {
                    float* o = (float*)p.out + orow * p.ldo + c;
                    *(float4*)o = make_float4(x[0], x[1], x[2], x[3]);
                } else {  // OUT_F32_ACC
                    float* o = (float*)p.out + orow * p.ldo + c;
                    float4 old = *(float4*)o;
                    *(float4*)o = make_float4(old.x + x[0], old.y + x[1], old.z + x[2], old.w + x[3]);
                }
            }
        }
    }
}

// One workgroup per CU, rounded down to a multiple of 8 so that a workgroup's virtual block indices all map to its XCD.
int persistent_grid(int cus) { return std::max(8, (cus > 0 ? std::min(cus, device_cus()) : device_cus()) & ~7); }

template <int OUT, int AMODE = A_PLAIN>
void launch_o(const GemmParams& p, hipStream_t s) {
    const int tiles = cdiv(p.M, 256) * cdiv(p.N, 256);
    static PerDeviceOnce once;
    once.run([&] { HIP_CHECK(hipFuncSetAttribute((const void*)gemm256_kernel<OUT, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); });
    static const std::string name = std::string(AMODE == A_PLAIN ? "gemm256_kernel<" : "gemm256_conv_kernel<") + std::to_string(OUT) + ">";
    {
        ProfScope ps(name.c_str(), 2.0 * p.M * p.N * p.K, s);
        hipLaunchKernelGGL((gemm256_kernel<OUT, AMODE>), dim3(std::min(tiles, persistent_grid(p.cus)), 1, p.ksplit), dim3(512), LDS_BYTES, s, p);
    }
    HIP_CHECK(hipGetLastError());
}

}  // namespace

void launch_gemm256_conv(const GemmParams& p, hipStream_t s) {
    // per-lane offsets are taken from the tile's first chunk: the chunks a 256-row tile can touch must stay inside 1 GB
    const int64_t chunks_per_tile = 256 / (p.Hout * p.Wout) + 2;
    Q3_REQUIRE(p.Hout * p.Wout > 0 && (chunks_per_tile + 1) * p.Hin * p.Win * p.Cin * 2 < (1ll << 30) && 257ll * p.K * 2 < (1ll << 30) &&
                   p.ksplit == 1 && (p.Cin + 63) / 64 * 9 < 4096,
               "gemm256: convolution shape outside the kernel's addressing range");
    launch_o<OUT_BF16, A_CONV3X3S2>(p, s);
}

void launch_gemm256(const GemmParams& p, GemmOut out, hipStream_t s) {
    // staging offsets are byte offsets from a tile's first row and must stay inside the descriptors' 1 GB range
    Q3_REQUIRE((256 * p.lda + p.K) * 2 < (1ll << 30) && 257ll * p.K * 2 < (1ll << 30), "gemm256: row stride too large for the buffer range");
    switch (out) {
        case OUT_BF16: launch_o<OUT_BF16>(p, s); break;
        case OUT_F32: launch_o<OUT_F32>(p, s); break;
        case OUT_F32_ACC: launch_o<OUT_F32_ACC>(p, s); break;
        case OUT_SWIGLU: launch_o<OUT_SWIGLU>(p, s); break;
        case OUT_ARGMAX: launch_o<OUT_ARGMAX>(p, s); break;
        case OUT_SLAB: launch_o<OUT_SLAB>(p, s); break;
    }
}
