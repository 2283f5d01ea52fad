// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T, fp32 accumulate, fused epilogues.
//
// Tile BM x BN x 64 per 256-thread workgroup (4 waves as 2 x 2), mfma_f32_16x16x32_bf16.
// Both operands are K-contiguous in memory; tiles are staged global -> LDS by 16-byte
// global_load_lds (LDS image lane-linear, XOR swizzle applied on the source chunk and on the
// fragment read), double buffered, one barrier per K tile.
//
// The weight fragment is fed as the MFMA "A" operand and the activation fragment as "B", so an
// accumulator register holds consecutive OUTPUT COLUMNS of one row.  Weight rows are assigned to
// MFMA rows through a permutation that makes every lane own 4*NT contiguous columns of a row, so
// the epilogue stores 16-byte vectors.
#include "gemm.h"
#include "prof.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

namespace {

// LDS ring depth: the 128x128 tile keeps two 32 KB stages (two workgroups per CU); the smaller
// tiles, used when few rows meet wide weights (decode), keep four so that three tiles of weights
// are in flight per workgroup.
constexpr int gemm_stages(int mt, int nt) { return (mt == 4 && nt == 4) ? 2 : (mt == 2 && nt == 2) ? 4 : 3; }

struct ConvRow {
    int64_t base;    // element offset of the row's patch origin (input row iy0, column ix0, channel 0): may point before the image
    unsigned valid;  // bit t set: tap t = kh * 3 + kw of this output position lies inside the (valid part of the) image
    int tap, ci;     // kernel tap and input channel of this thread's 8-channel chunk in the K tile staged next
};

template <int MT, int NT, int AMODE, int OUT, int WFMT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int AP = BM * 8 / 256, WP = BN * 8 / 256;  // 16-byte chunks per thread per tile
    constexpr int STAGES = gemm_stages(MT, NT);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int li = lane & 15, lg = lane >> 4;

    // ---- workgroup -> tile: XCD-contiguous remap, then groups of 8 M-tiles share each N-tile ----
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int L;
    {
        const int nwg = gridDim.x, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int grp = L / per_group, first_m = grp * GROUP;
    const int gsz = min(GROUP, tiles_m - first_m);
    const int tm = first_m + (L % per_group) % gsz, tn = (L % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- K range of this split ----
    const int nk_total = (p.K + 63) >> 6;
    const int per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt0 = blockIdx.z * per, kt1 = min(nk_total, kt0 + per);

    // ---- per-thread source descriptors ----
    // Plain rows and bf16 weights are read through buffer descriptors based at the tile's first row: a
    // load then needs a 32-bit per-lane byte offset (fixed) plus a scalar K offset, and no vector ALU work
    // inside the K loop.  Offsets at or beyond BUF_RANGE read as zero (used for the K tail).
    constexpr int BUF_RANGE = 0x40000000, BUF_OOB = 0x7f000000;
    const auto a_rsrc = uniform_buffer(p.A + (AMODE == A_PLAIN ? (int64_t)m0 * p.lda : 0), BUF_RANGE);
    const auto w_rsrc = uniform_buffer((const bf16_t*)p.W + (WFMT == W_BF16 ? (int64_t)n0 * p.K : 0), BUF_RANGE);
    int a_voff[AP];
    ConvRow crow[AP];
    int a_koff[AP];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int c = q * 256 + tid, row = c >> 3, cc = c & 7;
        const int sc = cc ^ ((row >> 1) & 7);
        a_koff[q] = sc * 8;
        const int m = min(m0 + row, p.M - 1);
        if constexpr (AMODE == A_PLAIN) {
            a_voff[q] = ((m - m0) * (int)p.lda + sc * 8) * 2;
        } else {
            const int hw = p.Hout * p.Wout;
            const int chunk = m / hw, rem = m - chunk * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
            const int wv = min(p.Win, p.win_valid ? p.win_valid[chunk] : p.Win);
            crow[q].base = ((int64_t)chunk * p.Hin + iy0) * p.Win * p.Cin + (int64_t)ix0 * p.Cin;
            unsigned valid = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < wv) valid |= 1u << t;
            }
            crow[q].valid = valid;
            // K is ordered (kh, kw, cin); Cin is a multiple of 8, so a 16-byte chunk never straddles two taps.  The K
            // tiles are staged in ascending order, so (tap, ci) advance by 64 channels per stage call: no division in the loop.
            const int k0 = kt0 * 64 + a_koff[q];
            crow[q].tap = k0 / p.Cin;
            crow[q].ci = k0 - crow[q].tap * p.Cin;
        }
    }
    int w_voff[WP];
    int w_koff[WP];
    if constexpr (WFMT == W_BF16) {
#pragma unroll
        for (int q = 0; q < WP; ++q) {
            const int c = q * 256 + tid, row = c >> 3, cc = c & 7;
            const int f = ((row >> 1) & 1) | (((row / (4 * NT)) & 3) << 1);
            const int sc = cc ^ f;
            w_koff[q] = sc * 8;
            const int n = min(n0 + row, p.N - 1);
            w_voff[q] = ((n - n0) * p.K + sc * 8) * 2;
        }
    }

    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE;
        char* sw = sa + A_BYTES;
        const int kbase = kt * 64;
        const bool tail = kbase + 64 > p.K;  // wave-uniform: only the last K tile of a ragged K masks lanes
        if constexpr (AMODE == A_PLAIN) {
#pragma unroll
            for (int q = 0; q < AP; ++q) {
                const int v = (tail && kbase + a_koff[q] >= p.K) ? BUF_OOB : a_voff[q];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, LDS_PTR(sa + (q * 256 + uwave * 64) * 16), 16, v, kbase * 2, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < AP; ++q) {
                const int tap = crow[q].tap, ci = crow[q].ci;
                const bool ok = tap < 9 && ((crow[q].valid >> tap) & 1u);       // tap >= 9: past the end of K
                const int kh = (tap * 11) >> 5;                                  // tap / 3 for tap < 32
                const int toff = (kh * p.Win + (tap - 3 * kh)) * p.Cin + ci;
                const bf16_t* src = ok ? p.A + crow[q].base + toff : p.zero;
                __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sa + (q * 256 + uwave * 64) * 16), 16, 0, 0);
                int nci = ci + 64, ntap = tap;
                while (nci >= p.Cin) { nci -= p.Cin; ++ntap; }                   // at most once when Cin >= 64
                crow[q].ci = nci;
                crow[q].tap = ntap;
            }
        }
        if constexpr (WFMT == W_BF16) {
#pragma unroll
            for (int q = 0; q < WP; ++q) {
                const int v = (tail && kbase + w_koff[q] >= p.K) ? BUF_OOB : w_voff[q];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(sw + (q * 256 + uwave * 64) * 16), 16, v, kbase * 2, 0, 0);
            }
        }
    };
    // MLX affine 4-bit weights: a row's 64 weights of a K tile are 8 u32 codes (32 bytes) under one scale and bias.
    // A thread takes half a row: one 16-byte code load plus the row's scale and bias, fetched into registers a whole
    // iteration before they are dequantised (w = q * scale + bias, rounded to bf16) and written as four 16-byte
    // chunks to their swizzled LDS slots, so the latency hides behind the MFMAs of the tile in between.
    constexpr bool Q4 = WFMT == W_Q4;
    const bool q4_active = Q4 && tid < BN * 2;   // BN rows x 2 halves (all 256 threads for the 128-row tiles)
    const int q4_row = tid >> 1, q4_half = tid & 1;
    uint4 q4_code = make_uint4(0, 0, 0, 0);
    uint32_t q4_sb = 0;
    auto q4_fetch = [&](int kt) {
        if constexpr (Q4) {
            if (q4_active) {
                const int k0 = kt * 64 + q4_half * 32;
                const int n = min(n0 + q4_row, p.N - 1);
                q4_code = make_uint4(0, 0, 0, 0);
                q4_sb = 0;
                if (k0 < p.K) {   // K is a multiple of 64, so a half row is inside or outside as a whole
                    const int groups = p.K / p.qgroup;
                    q4_code = *(const uint4*)((const uint32_t*)p.W + (int64_t)n * (p.K >> 3) + (k0 >> 3));
                    q4_sb = (uint32_t)p.qscales[(int64_t)n * groups + k0 / p.qgroup] |
                            ((uint32_t)p.qbiases[(int64_t)n * groups + k0 / p.qgroup] << 16);
                }
            }
        }
    };
    auto q4_store = [&](int buf, int kt) {
        if constexpr (Q4) {
            if (q4_active) {
                char* sw = smem + buf * STAGE + A_BYTES;
                const int f = ((q4_row >> 1) & 1) | (((q4_row / (4 * NT)) & 3) << 1);
                const float s = __uint_as_float(q4_sb << 16), b = __uint_as_float(q4_sb & 0xffff0000u);
                const uint32_t codes[4] = {q4_code.x, q4_code.y, q4_code.z, q4_code.w};
                const bool inside = kt * 64 + q4_half * 32 < p.K;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (inside) {
                        float w[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] = (float)((codes[j] >> (4 * e)) & 15u) * s + b;
                        v = make_uint4(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]), pack_bf16x2(w[4], w[5]),
                                       pack_bf16x2(w[6], w[7]));
                    }
                    const int logical = q4_half * 4 + j;   // 8-weight chunk of the row within this K tile
                    *(uint4*)(sw + (q4_row * 8 + (logical ^ f)) * 16) = v;
                }
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes within a stage)
    int a_rd[2], w_rd[2];
    {
        const int fa = (li >> 1) & 7;
        const int fw = ((li >> 1) & 1) | ((li >> 2) << 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a_rd[ks] = (wr * 16 * MT + li) * 128 + (((lg + 4 * ks) ^ fa) << 4);
            w_rd[ks] = A_BYTES + (wc * 16 * NT + (li >> 2) * (4 * NT) + (li & 3)) * 128 + (((lg + 4 * ks) ^ fw) << 4);
        }
    }

    // STAGES-deep ring of LDS tiles.  Iteration t: wait until this wave's loads of tile t have
    // landed (counted vmcnt: the STAGES-2 younger tiles stay in flight), barrier (now every wave's
    // part of tile t is in LDS and every wave has finished reading tile t-1), refill the buffer
    // tile t-1 occupied with tile t+STAGES-1, compute tile t.
    constexpr int INFLIGHT = (STAGES - 2) * (AP + (WFMT == W_BF16 ? WP : 0));
    static_assert(INFLIGHT <= 63, "vmcnt field overflow");
    const int nt = kt1 - kt0;
    if (nt > 0) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nt) {
                q4_fetch(kt0 + s);
                stage(s, kt0 + s);
                q4_store(s, kt0 + s);
            }
        for (int t = 0; t < nt; ++t) {
            if (t + STAGES - 2 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool refill = t + STAGES - 1 < nt;
            if (refill) {
                q4_fetch(kt0 + t + STAGES - 1);   // ahead of the LDS-direct loads, so waiting for it leaves those in flight
                stage((t + STAGES - 1) % STAGES, kt0 + t + STAGES - 1);
            }
            const char* sb = smem + (t % STAGES) * STAGE;
            // All fragment reads of the K tile are issued before its first MFMA (pinned: left alone, the
            // scheduler alternates small read groups with MFMAs behind full lgkmcnt waits, and a workgroup
            // with one wave per SIMD then sits through the LDS latency several times per tile).
            bf16x8 wf[2][NT], af[2][MT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int b = 0; b < NT; ++b) wf[ks][b] = *(const bf16x8*)(sb + w_rd[ks] + b * 4 * 128);
#pragma unroll
                for (int a = 0; a < MT; ++a) af[ks][a] = *(const bf16x8*)(sb + a_rd[ks] + a * 16 * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][b], af[ks][a], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (refill) q4_store((t + STAGES - 1) % STAGES, kt0 + t + STAGES - 1);
        }
    }

    // ---- epilogue: lane (li, lg) owns row m, columns c0 .. c0 + 4*NT - 1 ----
    const int c0 = n0 + wc * 16 * NT + lg * 4 * NT;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int m = m0 + wr * 16 * MT + a * 16 + li;
        const bool row_ok = m < p.M;
        int64_t orow = m;
        if (row_ok && p.rowmap) orow = p.rowmap[m];
        if constexpr (OUT == OUT_ARGMAX) {
            float best = -INFINITY;
            int bidx = 0x7fffffff;
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + b * 4 + r;
                    const float v = acc[a][b][r];
                    if (c < p.N && v > best) { best = v; bidx = c + p.col_base; }
                }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bidx, o, 64);
                if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
            }
            if (row_ok && lg == 0) {
                const int part = p.part_base + tn * 2 + wc;
                p.part_val[(int64_t)m * p.nparts + part] = best;
                p.part_idx[(int64_t)m * p.nparts + part] = bidx;
            }
            continue;
        }
        if (!row_ok || orow < 0) continue;
        float v[4 * NT];
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[b * 4 + r] = acc[a][b][r];
        if constexpr (OUT == OUT_SLAB) {
            float* o = (float*)p.out + blockIdx.z * p.slab_stride + orow * p.ldo + c0;
#pragma unroll
            for (int b = 0; b < NT; ++b)
                if (c0 + b * 4 < p.N) *(float4*)(o + b * 4) = make_float4(v[b * 4], v[b * 4 + 1], v[b * 4 + 2], v[b * 4 + 3]);
            continue;
        }
        if constexpr (OUT == OUT_SWIGLU) {
            // weight rows come in blocks of 16: 8 gate rows then the 8 matching up rows
            if (c0 < p.N) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = v[j], u = v[8 + j];
                    y[j] = g / (1.0f + __expf(-g)) * u;
                }
                bf16_t* o = (bf16_t*)p.out + orow * p.ldo + (c0 >> 1);
                *(uint4*)o = make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]),
                                        pack_bf16x2(y[6], y[7]));
            }
            continue;
        }
        const float* rt = p.rowtab ? p.rowtab + (int64_t)(m % p.rowtab_period) * p.N : nullptr;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
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

template <int MT, int NT, int AMODE, int OUT, int WFMT>
void launch_t(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const size_t lds = (size_t)gemm_stages(MT, NT) * (BM + BN) * 128;
    static PerDeviceOnce once;
    once.run([&] {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<MT, NT, AMODE, OUT, WFMT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
    static const std::string name = "gemm_kernel<" + std::to_string(MT) + "," + std::to_string(NT) + "," + std::to_string(AMODE) + "," +
                                    std::to_string(OUT) + "," + std::to_string(WFMT) + ">";
    {
        ProfScope ps(name.c_str(), 2.0 * p.M * p.N * p.K, s);
        hipLaunchKernelGGL((gemm_kernel<MT, NT, AMODE, OUT, WFMT>), dim3(tiles, 1, p.ksplit), dim3(256), lds, s, p);
    }
    HIP_CHECK(hipGetLastError());
}

template <int AMODE, int OUT, int WFMT>
void launch_tile(const GemmParams& p, int tile, hipStream_t s) {
    if (tile == 0) launch_t<4, 4, AMODE, OUT, WFMT>(p, s);
    else if (tile == 1) {
        if constexpr (OUT == OUT_SWIGLU) throw Q3Error("gemm: SwiGLU epilogue needs a 128-column tile");
        else launch_t<2, 2, AMODE, OUT, WFMT>(p, s);
    } else {
        if constexpr (OUT == OUT_SWIGLU) launch_t<2, 4, AMODE, OUT, WFMT>(p, s);
        else launch_t<4, 2, AMODE, OUT, WFMT>(p, s);
    }
}

template <int AMODE, int WFMT>
void launch_out(const GemmParams& p, GemmOut out, int tile, hipStream_t s) {
    switch (out) {
        case OUT_BF16: launch_tile<AMODE, OUT_BF16, WFMT>(p, tile, s); break;
        case OUT_F32: launch_tile<AMODE, OUT_F32, WFMT>(p, tile, s); break;
        case OUT_F32_ACC: launch_tile<AMODE, OUT_F32_ACC, WFMT>(p, tile, s); break;
        case OUT_SWIGLU: launch_tile<AMODE, OUT_SWIGLU, WFMT>(p, tile, s); break;
        case OUT_ARGMAX: launch_tile<AMODE, OUT_ARGMAX, WFMT>(p, tile, s); break;
        case OUT_SLAB: launch_tile<AMODE, OUT_SLAB, WFMT>(p, tile, s); break;
    }
}

}  // namespace

int gemm_pick_tile(int M, int N) {
    // Q3ASR_FORCE_TILE pins the automatic choice (parity tests run small models through every tile)
    const int pin = q3_knobs().force_tile;
    if (pin >= 0 && pin <= 3) return pin;
    const int64_t t256 = (int64_t)cdiv(M, 256) * cdiv(N, 256);
    if (t256 >= 224) return 3;  // about one workgroup per CU or more
    const int64_t t128 = (int64_t)cdiv(M, 128) * cdiv(N, 128);
    if (t128 >= 384) return 0;
    return 1;
}

int gemm_argmax_parts(int N, int tile) {
    if (tile == 3) return cdiv(N, 256) * 4;
    const int bn = (tile == 0) ? 128 : 64;
    return cdiv(N, bn) * 2;
}

void launch_gemm(const GemmParams& p, GemmAMode amode, GemmOut out, GemmWFmt wfmt, int tile, hipStream_t s) {
    Q3_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    Q3_REQUIRE(p.K % 32 == 0 && p.N % 4 == 0, "gemm: K must be a multiple of 32 and N of 4");
    Q3_REQUIRE(p.zero != nullptr, "gemm: zero page missing");
    // per-lane byte offsets from a tile's first row must stay inside the buffer descriptors' 1 GB range
    Q3_REQUIRE((129 * p.lda + p.K) * 2 < (1ll << 30) && 129ll * p.K * 2 < (1ll << 30), "gemm: row stride too large for the buffer range");
    if (amode == A_PLAIN) Q3_REQUIRE(p.lda % 8 == 0 && p.lda >= p.K, "gemm: lda must be >= K and a multiple of 8");
    else Q3_REQUIRE(p.K == 9 * p.Cin && p.Cin % 8 == 0, "gemm: conv K must be 9*Cin, Cin multiple of 8");
    if (wfmt == W_Q4) Q3_REQUIRE(p.qgroup == 64 && p.K % 64 == 0, "gemm: q4 needs group 64 and K multiple of 64");
    if (out == OUT_SWIGLU) Q3_REQUIRE(p.N % 16 == 0, "gemm: SwiGLU needs N multiple of 16");
    if (out != OUT_SLAB) Q3_REQUIRE(p.ksplit == 1, "gemm: split-K only with slab output");
    const bool conv256 = amode == A_CONV3X3S2 && wfmt == W_BF16 && out == OUT_BF16;
    const bool can256 = (amode == A_PLAIN && wfmt == W_BF16) || conv256;
    if (tile < 0) {
        tile = gemm_pick_tile(p.M, p.N);
        if (tile == 3 && !can256) tile = 0;
    }
    if (tile == 3 && conv256) {
        Q3_REQUIRE(p.K == 9 * p.Cin && p.Cin % 8 == 0, "gemm: conv K must be 9*Cin, Cin multiple of 8");
        launch_gemm256_conv(p, s);
        return;
    }
    if (tile == 3) {
        Q3_REQUIRE(can256, "gemm: the 256x256 tile needs plain rows and bf16 weights");
        if (out == OUT_ARGMAX) Q3_REQUIRE(p.part_base + gemm_argmax_parts(p.N, 3) <= p.nparts, "gemm: nparts mismatch");
        launch_gemm256(p, out, s);
        return;
    }
    if (out == OUT_SWIGLU && tile == 1) tile = 2;
    if (out == OUT_ARGMAX) Q3_REQUIRE(p.part_base + gemm_argmax_parts(p.N, tile == 2 ? 1 : tile) <= p.nparts, "gemm: nparts mismatch");
    if (amode == A_PLAIN) {
        if (wfmt == W_BF16) launch_out<A_PLAIN, W_BF16>(p, out, tile, s);
        else launch_out<A_PLAIN, W_Q4>(p, out, tile, s);
    } else {
        Q3_REQUIRE(wfmt == W_BF16 && out == OUT_BF16, "gemm: conv path is bf16 weights / bf16 output");
        launch_tile<A_CONV3X3S2, OUT_BF16, W_BF16>(p, tile, s);
    }
}
