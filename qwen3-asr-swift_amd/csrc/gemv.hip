// Decoder projections for a handful of rows (one realtime utterance, or a few): weights stream straight from HBM
// into registers, 16 bytes per lane, and meet the activation rows in LDS; no MFMA, no split-K partial sums.
//
// The reference runs exactly this regime -- batch 1 -- through MLX's quantizedMM / matmul
// (Sources/Qwen3ASR/QuantizedTextDecoder.swift:54-67, 130-268).  Three fused forms cover a decoder layer's projections:
//   GEMV_NORM_BF16   RMSNorm of the fp32 residual rows (:294-299 via MLXFast.rmsNorm) -> projection -> bf16 rows   (q/k/v)
//   GEMV_ACC_F32     bf16 rows -> projection -> += fp32 residual                                                   (o, down)
//   GEMV_NORM_SWIGLU RMSNorm -> gate/up projection (rows interleaved 8 gate / 8 up) -> silu(g) * u -> bf16 rows      (MLP in)
// Every output element has one owner lane, so the residual update is deterministic.
#include "gemv.h"
#include "prof.h"

namespace {

constexpr int GEMV_THREADS = 256;

// One wave per weight row (SwiGLU: per gate row and the up row 8 below it), four rows or pairs per workgroup.  A lane's
// share of the row -- NCH chunks of 8 weights, 512 apart -- is requested in one burst before anything else happens, so the
// whole projection is in flight at once (a layer's matrices are smaller than what HBM needs outstanding to reach its rate);
// the activation rows are normalised / copied into LDS underneath those loads.
// WQ: MLX affine 4-bit weights (u32 codes, low nibble first, bf16 scale and bias per 64 weights of a row;
// QuantizedTextDecoder.swift:54-67).  A chunk is then 16 weights per lane, 1024 apart, and every weight is rebuilt as
// bf16(q * scale + bias), the value the MFMA path multiplies with.
template <int MODE, int M, int NCH, bool WQ>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t xs[];  // [M][K]
    __shared__ float part[M][4];
    constexpr int R = MODE == GEMV_NORM_SWIGLU ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K;

    // ---- activation loads first (their consumers run while the weights are still in flight) ----
    constexpr int NA = (M * NCH * 64 * (WQ ? 2 : 1) + GEMV_THREADS - 1) / GEMV_THREADS;  // 16-byte pieces of the bf16 rows per thread
    uint4 av[MODE == GEMV_ACC_F32 ? NA : 1];
    float4 v[MODE == GEMV_ACC_F32 ? 1 : M][2];
    const int d4 = K >> 2, k8 = K >> 3;
    if (MODE == GEMV_ACC_F32) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + GEMV_THREADS * i;
            av[i] = c < M * k8 ? *(const uint4*)(p.a + (int64_t)(c / k8) * p.lda + 8 * (c % k8)) : make_uint4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = tid + GEMV_THREADS * j;
                v[m][j] = c < d4 ? ((const float4*)(p.hid + (int64_t)m * K))[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }

    // ---- this wave's weight rows, all chunks requested now ----
    int row[R];
    if (MODE == GEMV_NORM_SWIGLU) {
        const int pair = blockIdx.x * 4 + wave;             // gate row index over the whole matrix
        row[0] = (pair >> 3) * 16 + (pair & 7);              // 16-row blocks: 8 gate rows, then their 8 up rows
        row[R - 1] = row[0] + 8;
    } else {
        row[0] = blockIdx.x * 4 + wave;
    }
    const bf16_t* w = (const bf16_t*)p.w;
    constexpr int PER_LANE = WQ ? 16 : 8;   // weights per lane and chunk
    uint4 wv[R][NCH];                       // WQ: .x/.y = codes, .z = scale and bias bits
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k0 = (c * 64 + lane) * PER_LANE;
            wv[r][c] = make_uint4(0, 0, 0, 0);
            if (row[r] < p.N && k0 < K) {
                if (WQ) {
                    const uint2 codes = *(const uint2*)((const uint32_t*)p.w + (int64_t)row[r] * (K >> 3) + (k0 >> 3));
                    const int64_t g = (int64_t)row[r] * (K >> 6) + (k0 >> 6);
                    wv[r][c] = make_uint4(codes.x, codes.y, (uint32_t)p.scales[g] | ((uint32_t)p.biases[g] << 16), 0);
                } else {
                    wv[r][c] = *(const uint4*)(w + (int64_t)row[r] * K + k0);
                }
            }
        }

    // ---- activation rows into LDS ----
    if (MODE == GEMV_ACC_F32) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + GEMV_THREADS * i;
            if (c < M * k8) *(uint4*)(xs + (c / k8) * K + 8 * (c % k8)) = av[i];
        }
    } else {
        // RMSNorm with the element-to-thread mapping and operation order of the stand-alone kernel, so both
        // give the same bf16 rows (K <= 2048: two float4 per thread)
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                sq += v[m][j].x * v[m][j].x + v[m][j].y * v[m][j].y + v[m][j].z * v[m][j].z + v[m][j].w * v[m][j].w;
            sq = wave_sum(sq);
            if (lane == 0) part[m][wave] = sq;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float rstd = 1.0f / sqrtf((part[m][0] + part[m][1] + part[m][2] + part[m][3]) / (float)K + p.eps);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = tid + GEMV_THREADS * j;
                if (c < d4) {
                    const float4 ww = ((const float4*)p.norm_w)[c];
                    *(uint2*)(xs + m * K + 4 * c) = make_uint2(pack_bf16x2(v[m][j].x * rstd * ww.x, v[m][j].y * rstd * ww.y),
                                                               pack_bf16x2(v[m][j].z * rstd * ww.z, v[m][j].w * rstd * ww.w));
                }
            }
        }
    }
    __syncthreads();

    // ---- dot products ----
    float acc[R][M];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = (c * 64 + lane) * PER_LANE;
        if (WQ) {
            if (k0 < K) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {   // 8 weights per code word
                    float wf[R][8];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t code = h ? wv[r][c].y : wv[r][c].x;
                        const float sc = __uint_as_float(wv[r][c].z << 16), bi = __uint_as_float(wv[r][c].z & 0xffff0000u);
#pragma unroll
                        for (int j = 0; j < 8; ++j) wf[r][j] = bf16_to_f32(f32_to_bf16((float)((code >> (4 * j)) & 15u) * sc + bi));
                    }
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const uint4 xv = *(const uint4*)(xs + m * K + k0 + 8 * h);
                        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[r][m] = fmaf(wf[r][2 * e], __uint_as_float(xw[e] << 16), acc[r][m]);
                                acc[r][m] = fmaf(wf[r][2 * e + 1], __uint_as_float(xw[e] & 0xffff0000u), acc[r][m]);
                            }
                    }
                }
            }
        } else if (k0 < K) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint4 xv = *(const uint4*)(xs + m * K + k0);
                const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t ww[4] = {wv[r][c].x, wv[r][c].y, wv[r][c].z, wv[r][c].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[r][m] = fmaf(__uint_as_float(ww[e] << 16), __uint_as_float(xw[e] << 16), acc[r][m]);
                        acc[r][m] = fmaf(__uint_as_float(ww[e] & 0xffff0000u), __uint_as_float(xw[e] & 0xffff0000u), acc[r][m]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);

    // ---- epilogue: lane 0 of the wave owns its rows' outputs ----
    if (lane == 0 && row[0] < p.N) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (MODE == GEMV_NORM_SWIGLU) {
                const float g = acc[0][m], u = acc[R - 1][m];
                const int col = (row[0] >> 4) * 8 + (row[0] & 7);
                ((bf16_t*)p.out)[(int64_t)m * p.ldo + col] = f32_to_bf16(g / (1.0f + __expf(-g)) * u);
            } else if (MODE == GEMV_ACC_F32) {
                ((float*)p.out)[(int64_t)m * p.ldo + row[0]] += acc[0][m];
            } else {
                ((bf16_t*)p.out)[(int64_t)m * p.ldo + row[0]] = f32_to_bf16(acc[0][m]);
            }
        }
    }
}

template <int MODE, int M, int NCH, bool WQ>
void launch_one(const GemvParams& p, hipStream_t s) {
    const int rows = MODE == GEMV_NORM_SWIGLU ? p.N / 2 : p.N;   // waves needed
    const size_t lds = (size_t)p.M * p.K * sizeof(bf16_t);
    hipLaunchKernelGGL((gemv_kernel<MODE, M, NCH, WQ>), dim3(cdiv(rows, 4)), dim3(GEMV_THREADS), lds, s, p);
}

template <int MODE, int M>
void launch_rows(const GemvParams& p, hipStream_t s) {
    if (p.scales) {   // 4-bit weights: 1024 per wave and chunk
        const int nch = cdiv(p.K, 1024);
        if (nch <= 1) launch_one<MODE, M, 1, true>(p, s);
        else if (nch <= 2) launch_one<MODE, M, 2, true>(p, s);
        else if (nch <= 3) launch_one<MODE, M, 3, true>(p, s);
        else launch_one<MODE, M, 6, true>(p, s);
        return;
    }
    const int nch = cdiv(p.K, 512);
    if (nch <= 1) launch_one<MODE, M, 1, false>(p, s);
    else if (nch <= 2) launch_one<MODE, M, 2, false>(p, s);
    else if (nch <= 4) launch_one<MODE, M, 4, false>(p, s);
    else if (nch <= 6) launch_one<MODE, M, 6, false>(p, s);
    else launch_one<MODE, M, 12, false>(p, s);
}

template <int MODE>
void launch_mode(const GemvParams& p, hipStream_t s) {
    switch (p.M) {
        case 1: launch_rows<MODE, 1>(p, s); break;
        case 2: launch_rows<MODE, 2>(p, s); break;
        case 3: launch_rows<MODE, 3>(p, s); break;
        default: launch_rows<MODE, 4>(p, s); break;
    }
}

}  // namespace

bool gemv_supports(int rows, int N, int K, int mode, int wfmt) {
    if (rows < 1 || rows > GEMV_MAX_ROWS || K % 8 != 0 || K > GEMV_MAX_K) return false;
    if (wfmt == W_Q4 && K % 64 != 0) return false;   // one scale / bias per 64 weights
    if (mode != GEMV_ACC_F32 && K > 2048) return false;   // the fused RMSNorm holds a row as two float4 per thread
    if (mode == GEMV_NORM_SWIGLU && N % 16 != 0) return false;
    return true;
}

void launch_gemv(const GemvParams& p, int mode, hipStream_t s) {
    Q3_REQUIRE(gemv_supports(p.M, p.N, p.K, mode, p.scales ? W_Q4 : W_BF16), "gemv: unsupported shape");
    Q3_REQUIRE(!p.scales || (p.biases && p.qgroup == 64), "gemv: 4-bit weights need scales, biases and groups of 64");
    Q3_REQUIRE(((uintptr_t)p.w & 15) == 0, "gemv: weights must be 16-byte aligned");
    static const char* names[3] = {"gemv_kernel<norm,bf16>", "gemv_kernel<acc_f32>", "gemv_kernel<norm,swiglu>"};
    ProfScope ps(names[mode], (double)p.N * p.K * (p.scales ? 0.5625 : 2.0), s);   // work = weight bytes streamed
    switch (mode) {
        case GEMV_NORM_BF16: launch_mode<GEMV_NORM_BF16>(p, s); break;
        case GEMV_ACC_F32: launch_mode<GEMV_ACC_F32>(p, s); break;
        default: launch_mode<GEMV_NORM_SWIGLU>(p, s); break;
    }
    HIP_CHECK(hipGetLastError());
}
