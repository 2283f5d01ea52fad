// Projections of the decoder for a handful of rows: weights streamed once, straight into registers.
#pragma once
#include "common.h"
#include "gemm.h"

constexpr int GEMV_MAX_ROWS = 4;     // activation rows held in LDS together
constexpr int GEMV_MAX_K = 6144;     // widest input (the 1.7B model's MLP)

enum GemvMode {
    GEMV_NORM_BF16 = 0,    // RMSNorm(hid) -> W -> bf16 rows
    GEMV_ACC_F32 = 1,      // bf16 rows -> W -> += fp32 rows
    GEMV_NORM_SWIGLU = 2,  // RMSNorm(hid) -> W (8 gate / 8 up row blocks) -> silu(g) * u -> bf16 rows [M][N/2]
};

struct GemvParams {
    const void* w = nullptr;  // [N][K] bf16, K contiguous; or u32 [N][K/8] 4-bit codes when scales is set
    const bf16_t* scales = nullptr;  // MLX affine 4-bit: [N][K/64] bf16 scale and bias
    const bf16_t* biases = nullptr;
    int qgroup = 64;
    int N = 0, K = 0, M = 0;
    const float* hid = nullptr;     // NORM modes: fp32 residual rows [M][K]
    const float* norm_w = nullptr;  //             RMSNorm weight [K]
    float eps = 0.f;
    const bf16_t* a = nullptr;      // ACC mode: bf16 rows, stride lda
    int64_t lda = 0;
    void* out = nullptr;
    int64_t ldo = 0;
};

bool gemv_supports(int rows, int N, int K, int mode, int wfmt);
void launch_gemv(const GemvParams& p, int mode, hipStream_t s);
