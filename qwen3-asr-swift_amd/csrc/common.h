// Shared declarations for libq3asr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <stdexcept>

typedef uint16_t bf16_t;  // raw bfloat16 bit pattern

struct Q3Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void q3_set_error(const std::string& msg);

#define HIP_CHECK(expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            throw Q3Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " +  \
                          __FILE__ + ":" + std::to_string(__LINE__));                          \
    } while (0)

#define Q3_REQUIRE(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) throw Q3Error(std::string("q3asr: ") + (msg)); \
    } while (0)

#ifdef __HIPCC__
__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// Round-to-nearest-even through the hardware conversion (v_cvt_pk_bf16_f32): one instruction per pair of values.
typedef __bf16 q3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float q3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const q3_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, q3_bf16x2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, f) & 0xffffu); }
// Buffer descriptor over `bytes` bytes at a workgroup-uniform address.  The address goes through
// readfirstlane so the compiler keeps the descriptor in SGPRs (otherwise every buffer instruction is
// wrapped in a loop that re-checks its uniformity).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_buffer(const void* base, int bytes) {
    const uint64_t b = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
}
// Exact (erf) GELU, 0.5 x (1 + erf(x / sqrt 2)) (AudioEncoder.swift:379-384, 127-129: MLX gelu).  erfc(|z|) = p e by the
// rational approximation of Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 in erf): two transcendental instructions
// and a dozen multiply-adds instead of the library erff's ~30 instructions.  1 + erf(z) is p e for z < 0 and 2 - p e
// otherwise, so the negative tail is a product, not a cancellation: the GELU keeps the approximation's own relative
// accuracy there (3e-5 at x = -3, 7e-4 at x = -4.2, below one bf16 step of 3.9e-3 down to x = -6, where the value is
// 1e-8) and 5e-7 absolute everywhere.  Every caller rounds the result to bf16.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752440f, a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    const float p = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
    const float pe = p * e;
    return 0.5f * x * (z < 0.0f ? pe : 2.0f - pe);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Maximum over the 64 lanes of a wave, returned in every lane.  All 64 lanes must be active.  Lane exchanges inside a row
// of 16 are register moves with a DPP modifier (a shuffle through the LDS crossbar costs a memory round trip per step);
// the four row results are then read out of lanes 0 / 16 / 32 / 48.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_move<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_move<0x124>(v));   // row_ror:4
    v = fmaxf(v, dpp_move<0x128>(v));   // row_ror:8
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
#endif

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Environment switches the kernel launchers look at (tests and tuning runs pin a kernel variant with them).  They are
// read once at the start of every API call into a snapshot that belongs to the calling thread: a decode step launches
// tens of thousands of kernels, and getenv is not safe against a host thread that calls setenv meanwhile.
struct Q3Knobs {
    int flash_pair = -1;   // Q3ASR_FLASH_PAIR: 0 = one query head per workgroup in the prefill attention
    int attn_waves = 0;    // Q3ASR_ATTN_WAVES: 4 or 8 pins the decode attention's wave count
    int force_tile = -1;   // Q3ASR_FORCE_TILE: 0..3 pins the GEMM tile where the choice is automatic
};
Q3Knobs& q3_knobs();
void q3_knobs_refresh();

// One-time set-up per HIP device (function attributes, device properties): several models on several devices and
// several host threads may reach a launcher at once.  A set-up that throws is retried by the next caller.
struct PerDeviceOnce {
    static constexpr int MAX_DEVICES = 64;
    std::once_flag flags[MAX_DEVICES];
    template <class F>
    void run(F&& fn) {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        Q3_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device ordinal above 63");
        std::call_once(flags[dev], fn);
    }
};
// Compute units of the current device.
int device_cus();
