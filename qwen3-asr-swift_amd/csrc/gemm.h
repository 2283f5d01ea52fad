// bf16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues (gfx950).
#pragma once
#include "common.h"

enum GemmAMode { A_PLAIN = 0, A_CONV3X3S2 = 1 };
enum GemmOut {
    OUT_BF16 = 0,     // out bf16 [rows][ldo] = act(acc + bias + rowtab)
    OUT_F32 = 1,      // out f32  [rows][ldo] = act(acc + bias + rowtab)
    OUT_F32_ACC = 2,  // out f32  [rows][ldo] += acc + bias        (residual add in place)
    OUT_SWIGLU = 3,   // weight rows interleaved 8 gate / 8 up; out bf16 [rows][N/2] = silu(g) * u
    OUT_ARGMAX = 4,   // per (row, 16*NT-column block) running (max, argmin-index) partials
    OUT_SLAB = 5      // out f32 [ksplit][M][N] partial sums (split-K)
};
enum GemmWFmt { W_BF16 = 0, W_Q4 = 1 };

struct GemmParams {
    const bf16_t* A = nullptr;  // activations, row-major [M][lda] (A_PLAIN) or NHWC image (A_CONV3X3S2)
    int64_t lda = 0;
    const void* W = nullptr;    // W_BF16: bf16 [N][K];  W_Q4: u32 [N][K/8] (MLX affine 4-bit, low nibble first)
    const bf16_t* qscales = nullptr;  // W_Q4: bf16 [N][K/group]
    const bf16_t* qbiases = nullptr;
    int qgroup = 64;
    int M = 0, N = 0, K = 0;
    const bf16_t* zero = nullptr;  // >= 64 zero bytes in device memory (source of masked loads)
    // A_CONV3X3S2: input [chunks][Hin][Win][Cin], output row m = (chunk, oy, ox); K = 9*Cin ordered (kh, kw, cin)
    int Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0;
    const int* win_valid = nullptr;  // per chunk: input columns >= win_valid[chunk] read as zero
    // epilogue
    const float* bias = nullptr;    // [N]
    const float* rowtab = nullptr;  // [period][N], added to row m as rowtab[m % period]
    int rowtab_period = 1;
    const int* rowmap = nullptr;    // output row of m (negative: row dropped)
    int act = 0;                    // 1: exact (erf) GELU
    void* out = nullptr;
    int64_t ldo = 0;
    float* part_val = nullptr;      // OUT_ARGMAX: [M][nparts]
    int* part_idx = nullptr;
    int nparts = 0;
    int part_base = 0;              // OUT_ARGMAX over a column range of a wider matrix: first partial slot of this launch
    int col_base = 0;               //   and the column index of W's first row in that matrix (added to the indices)
    int cus = 0;                    // compute units the launch stream may use (0: the whole device); sizes persistent grids
    int ksplit = 1;                 // OUT_SLAB: gridDim.z
    int64_t slab_stride = 0;
};

// tile: 0 -> 128x128, 1 -> 64x64, 2 -> 128x64, 3 -> 256x256 (BM x BN; 3 needs plain rows and bf16 weights);
// -1 -> pick by shape
void launch_gemm(const GemmParams& p, GemmAMode amode, GemmOut out, GemmWFmt wfmt, int tile, hipStream_t s);
// the 256x256 8-wave kernel (gemm256.hip); reached through launch_gemm
void launch_gemm256(const GemmParams& p, GemmOut out, hipStream_t s);
// the same kernel as an implicit-GEMM 3x3 stride-2 convolution (bf16 output); reached through launch_gemm
void launch_gemm256_conv(const GemmParams& p, hipStream_t s);
// number of 16*NT-column partial blocks the OUT_ARGMAX epilogue writes per row for a given N / tile
int gemm_argmax_parts(int N, int tile);
int gemm_pick_tile(int M, int N);
