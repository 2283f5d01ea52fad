// Per-head RMSNorm followed by rotary position embedding, one wave per 128-wide head
// (QuantizedTextDecoder.swift:148-157 and :211-268): lane i holds elements 2i and 2i+1.
#pragma once
#include "common.h"

struct RopeTables {
    const float* cos_tab;  // [max_pos][half]
    const float* sin_tab;
    int half, interleaved, max_pos;
};

// (x0, x1) = this lane's two elements of the head; w = the head norm's weight; returns the rotated pair.
__device__ __forceinline__ float2 rope_head(float x0, float x1, int lane, const float* w, float eps, const RopeTables& t, int pos) {
    const float rstd = 1.0f / sqrtf(wave_sum(x0 * x0 + x1 * x1) / 128.0f + eps);
    const float y0 = x0 * rstd * w[2 * lane], y1 = x1 * rstd * w[2 * lane + 1];
    float r0 = y0, r1 = y1;
    const int pp = min(pos, t.max_pos - 1);
    const float* ct = t.cos_tab + (int64_t)pp * t.half;
    const float* st = t.sin_tab + (int64_t)pp * t.half;
    if (t.interleaved) {  // pairs (2i, 2i+1): QuantizedTextDecoder.swift:245-252
        if (lane < t.half) {
            const float c = ct[lane], s = st[lane];
            r0 = y0 * c - y1 * s;
            r1 = y1 * c + y0 * s;
        }
    } else {              // rotate-half, pairs (i, i + half): :253-259
        const int e0 = 2 * lane, h2 = t.half >> 1, rot = 2 * t.half;
        const int partner = (e0 < t.half) ? lane + h2 : lane - h2;
        const float p0 = __shfl(y0, partner & 63, 64), p1 = __shfl(y1, partner & 63, 64);
        if (e0 < t.half) {
            r0 = y0 * ct[e0] - p0 * st[e0];
            r1 = y1 * ct[e0 + 1] - p1 * st[e0 + 1];
        } else if (e0 < rot) {
            const int j = e0 - t.half;
            r0 = y0 * ct[j] + p0 * st[j];
            r1 = y1 * ct[j + 1] + p1 * st[j + 1];
        }
    }
    return make_float2(r0, r1);
}
