// Log-mel frontend for gfx950.
//
// Restates WhisperFeatureExtractor (reference: Sources/Qwen3ASR/AudioPreprocessing.swift:131-324): reflect padding, periodic
// Hann window, 201-bin power spectrum of 400-sample frames every 160 samples, slaney filterbank, log10.  The reference
// evaluates the spectrum as a direct DFT (:184-209); here it is an FFT on the vector ALU, which is what makes the stage
// bandwidth-bound: the 400 real samples of a frame are packed into 200 complex values z[m] = x[2m] + i x[2m+1], a
// 200-point complex FFT is taken as 8 x 25 (one radix-8 butterfly per (frame, residue mod 25), twiddles, one 25-point
// transform per (frame, residue mod 8), itself 5 x 5 in registers), and the real spectrum is untangled from Z[k] and
// Z[200-k].  About 10 kFLOP per frame instead of 160 kFLOP for the folded direct DFT.
//
// Work is cut into tiles of 32 frames of one utterance; persistent workgroups of 256 threads (two per CU, 76 KB of LDS each)
// loop over the tiles.  Per tile: the 5 360 samples it covers are staged in LDS (fetched into registers during the previous
// tile's arithmetic); phase 1 windows the samples and runs the 800 radix-8 butterflies; phase 2 the 256 25-point transforms;
// phase 3 untangles the 101 bin pairs of every frame into power values; phase 4 applies the slaney filterbank with
// compile-time filter supports, log10, the per-utterance running maximum (wave reduction + one atomic per wave) and stores
// [mel][frame] rows, 32 consecutive frames per store.  The kernel is bound by the LDS pipe and the vector ALU together
// (each about 70 % busy), so what shaped it is what takes work off those two: a thread keeps its residue n2 in phase 1,
// which puts its window values and twiddles in registers for the whole kernel (a third of the LDS traffic of the phase);
// the LDS reads of phase 3 (four items) and of a filterbank column tile are all issued before the arithmetic that uses
// them -- the compiler cannot move a read above an earlier write to the same array on its own; tiles that lie inside
// their utterance skip the per-frame edge selects; wave reductions are DPP register moves, not shuffles through LDS.
// (Measured and dropped: every wave owning 8 frames end to end with no workgroup barrier at all, 12 % slower -- the
// filterbank product then wastes half of its 16 rows and the stores come in 32-byte pieces; keeping the 8-byte LDS reads
// of phases 1 and 2 single instead of paired, which doubles their LDS rate, and a hand-written one-instruction-per-step
// DPP reduction: no change either -- neither pipe alone is the limit.)
//
// The max-8 clamp needs the utterance-wide maximum.  Inside the transcription pipeline the rows are stored raw and the
// consumer (conv1) clamps and scales them.  For callers that want the reference's exact output tensor the rows are stored
// already scaled, (y + 4) / 4, and the kernel also writes the minimum of every cell of 16 frames x 16 mel rows (sixteen
// per tile): the clamp only changes values more than eight decades under the utterance's peak (digital silence, a few
// low filters of a clean signal), so launch_mel_clamp afterwards reads 64 bytes per tile and touches only the cells
// whose minimum is below the floor.  x -> (x + 4) / 4 is monotonic in fp32, so clamping the scaled value to the scaled
// floor gives the bits of scaling the clamped value.  Twiddles are rounded from double; the window multiplies the
// samples in fp32 as the reference does (:161-164, 195-197).
#include "mel.h"
#include "prof.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {

constexpr int TF = MEL_TILE_FRAMES;                               // 32 frames per tile
constexpr int MEL_THREADS = 256;
constexpr int SAMPLES = (TF - 1) * MEL_HOP + MEL_NFFT;            // 5360 samples under a tile's frames
constexpr int XS_N = 6432;                                        // sample image, later the power rows [32][201]
constexpr int POW_STRIDE = MEL_NBINS;                             // 201: odd, so frames fall on distinct banks
constexpr int ZS = 200;                                           // complex values per frame row (see phase 2 for the bank argument)
constexpr int Z_N = TF * ZS;
// tables (floats): Hann window [400] | W200^(n2 k1) as (cos, -sin) pairs [8][25] | exp(-i pi k / 200) as (cos, -sin) pairs [101]
constexpr int TAB_HANN = 0, TAB_TW = 400, TAB_UW = 800, TAB_N = 1002;
constexpr int TAB_LDS = TAB_N - TAB_UW;                          // only the untangling twiddles are staged in LDS
constexpr int MEL_LDS_BYTES = (XS_N + 2 * Z_N + TAB_LDS + 6) * 4;  // 77 760 bytes: two workgroups per CU
static_assert(XS_N >= SAMPLES && XS_N >= TF * POW_STRIDE, "sample image and power rows share one region");
static_assert(MEL_LDS_BYTES <= 80 * 1024, "two workgroups must fit a CU's 160 KB");
constexpr int NLOAD = (SAMPLES + MEL_THREADS - 1) / MEL_THREADS;  // 21 samples per thread per tile

// Support of the slaney filters at 16 kHz / 400-point FFT / 128 bins (AudioPreprocessing.swift:23-126): first FFT bin and
// number of bins of every filter.  Fixed by the reference's constants; mel_tables_create checks the computed filterbank
// against it.  Knowing it at compile time turns the filterbank phase into straight-line code with immediate LDS offsets.
constexpr unsigned char FB_LO[MEL_NMELS] = {
    1, 1, 2, 2, 3, 3, 4, 5, 5, 6, 6, 7, 8, 8, 9, 9, 10, 10, 11, 12, 12, 13, 13, 14, 15, 15, 16, 16, 17, 17, 18, 19,
    19, 20, 20, 21, 22, 22, 23, 23, 24, 24, 25, 26, 26, 27, 28, 28, 29, 30, 30, 31, 32, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 41,
    42, 43, 44, 45, 46, 48, 49, 50, 51, 52, 54, 55, 56, 58, 59, 60, 62, 63, 65, 66, 68, 70, 71, 73, 75, 77, 79, 80, 82, 84, 86, 89,
    91, 93, 95, 98, 100, 102, 105, 107, 110, 113, 115, 118, 121, 124, 127, 130, 133, 136, 140, 143, 147, 150, 154, 158, 161, 165, 169, 174, 178, 182, 187, 191};
constexpr unsigned char FB_W[MEL_NMELS] = {
    1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1,
    1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 2, 1, 1, 2, 1, 1, 2, 1, 1, 2, 2, 2, 2, 1, 1, 2, 2, 2, 2, 2,
    2, 2, 2, 3, 3, 2, 2, 2, 3, 3, 2, 3, 3, 2, 3, 3, 3, 3, 3, 4, 3, 3, 4, 4, 4, 3, 3, 4, 4, 5, 5, 4,
    4, 5, 5, 4, 5, 5, 5, 6, 5, 5, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 8, 7, 7, 8, 9, 9, 8, 9, 9, 9, 9};
__device__ __forceinline__ unsigned enc_max(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_max(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct TileInfo {
    const float* x;  // utterance samples
    int64_t n;       // utterance length
    int u, f0;
};

__device__ __forceinline__ TileInfo tile_info(const float* audio, const int64_t* d_off, const int64_t* d_len, const int* tile_utt,
                                              const int* tile_f0, int t) {
    TileInfo ti;
    ti.u = tile_utt[t];
    ti.f0 = tile_f0[t];
    ti.n = d_len[ti.u];
    ti.x = audio + d_off[ti.u];
    return ti;
}

// This thread's share (samples tid, tid + 256, ...) of a tile's sample window, in the reflect-padded signal
// (AudioPreprocessing.swift:134-154: 200 samples mirrored at either end, edge-safe for short audio; zero beyond).
__device__ __forceinline__ void fetch_samples(const TileInfo& ti, float (&sv)[NLOAD], int tid) {
    const int64_t P0 = (int64_t)ti.f0 * MEL_HOP, n = ti.n;
    if (P0 >= 200 && P0 + NLOAD * MEL_THREADS <= n + 200) {   // interior: no padding inside the window (workgroup-uniform)
        const float* xs = ti.x + (P0 - 200);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) sv[i] = xs[tid + i * MEL_THREADS];
        return;
    }
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int64_t P = P0 + tid + i * MEL_THREADS;
        int64_t idx = P < 200 ? 200 - P : (P < 200 + n ? P - 200 : 2 * n + 198 - P);
        idx = max(min(idx, n - 1), (int64_t)0);
        sv[i] = (n > 0 && P < n + MEL_NFFT) ? ti.x[idx] : 0.f;
    }
}

// The filterbank as a banded matrix product on the fp32 matrix pipe: mel[frame][filter] = sum_bin P[frame][bin] W[bin][filter].
// A 16-filter column tile only touches the bins between its first filter's lower edge and its last filter's upper edge, so
// it takes ct_steps() MFMAs of 16 x 16 x 4 instead of 51: 56 steps over the eight column tiles (a dense product needs 408).
constexpr int ct_lo(int ct) { return FB_LO[16 * ct]; }
constexpr int ct_hi(int ct) {
    int hi = 0;
    for (int m = 16 * ct; m < 16 * ct + 16; ++m) hi = FB_LO[m] + FB_W[m] > hi ? FB_LO[m] + FB_W[m] : hi;
    return hi;
}
constexpr int ct_steps(int ct) { return (ct_hi(ct) - ct_lo(ct) + 3) / 4; }
// Two sets of column tiles with 28 steps each; wave w works on set w / 2 for the 16 frames of row tile w % 2.
constexpr int SET_A[3] = {7, 5, 0}, SET_B[5] = {6, 4, 3, 2, 1};
constexpr int SET_STEPS = 28;
static_assert(ct_steps(7) + ct_steps(5) + ct_steps(0) == SET_STEPS && ct_steps(6) + ct_steps(4) + ct_steps(3) + ct_steps(2) + ct_steps(1) == SET_STEPS,
              "the two filter sets must be balanced");

struct MelTileOut {
    float* out;          // utterance's [128][t_stride] block + first frame of this lane's four
    int t_stride;
    int n_store;         // how many of the lane's four frames are returned (0..4)
    int n_max;           // how many of them count for the running maximum
    bool vec;            // 16-byte stores allowed
    float vmax;
    float* cell_min;     // SCALED: the eight per-column-tile minima of this wave's 16 frames
};

typedef __attribute__((ext_vector_type(4))) float f32x4;

// One column tile's products: CT = tile, OFF = position of its first weight fragment in the wave's register-resident list.
// All of the tile's LDS reads are issued before the first product, and a wave's column tiles are independent chains
// that the scheduler interleaves.
template <int CT, int OFF>
__device__ __forceinline__ f32x4 mel_col_product(const float* pw, const float (&bf)[SET_STEPS]) {
    float a[ct_steps(CT)];
#pragma unroll
    for (int s = 0; s < ct_steps(CT); ++s) a[s] = pw[ct_lo(CT) + 4 * s];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ct_steps(CT); ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bf[OFF + s], acc, 0, 0, 0);
    return acc;
}

// log10, running extrema and the store of one column tile.  SCALED: store (y + 4) / 4 instead of y and keep the
// minimum of what is stored.
// FULL: all four frames of every lane are returned and 16-byte stores are allowed (a tile inside its utterance).
template <int CT, bool SCALED, bool FULL>
__device__ __forceinline__ void mel_col_store(f32x4 acc, int n, int lane, MelTileOut& o) {
    // log10 of the clamped filter output (AudioPreprocessing.swift:232-255).  The hardware log2 is accurate to 1 ulp
    // and the argument is never denormal, so the plain product is within 2e-6 of the correctly rounded value.
    float y[4], vmin = INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma clang fp contract(off)   // log10 is rounded before the scale touches it, exactly as when it is stored raw and scaled by its consumer
        y[r] = __log2f(fmaxf(acc[r], 1e-10f)) * 0.30102999566398120f;
        o.vmax = fmaxf(o.vmax, FULL || r < o.n_max ? y[r] : -INFINITY);
        if (SCALED) {
            vmin = fminf(vmin, FULL || r < o.n_store ? y[r] : INFINITY);
            y[r] = (y[r] + 4.0f) * 0.25f;    // AudioPreprocessing.swift:286-288
        }
    }
    if (SCALED) {
        vmin = -wave_max(-vmin);
        if (lane == 0) o.cell_min[CT] = vmin;
    }
    float* dst = o.out + (int64_t)(16 * CT + n) * o.t_stride;
    if (FULL || (o.vec && o.n_store == 4)) {
        *(float4*)dst = make_float4(y[0], y[1], y[2], y[3]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < o.n_store) dst[r] = y[r];
    }
}

// Phase 4 of a tile for one wave: waves 0-1 apply column tiles {7, 5, 0} and waves 2-3 {6, 4, 3, 2, 1} to the 16 frames
// of row tile wave % 2.
template <bool SCALED, bool FULL>
__device__ __forceinline__ void mel_filterbank(const float* pw, const float (&bf)[SET_STEPS], int wave, int n, int lane, MelTileOut& o) {
    if (wave < 2) {
        const f32x4 a7 = mel_col_product<7, 0>(pw, bf);
        const f32x4 a5 = mel_col_product<5, ct_steps(7)>(pw, bf);
        const f32x4 a0 = mel_col_product<0, ct_steps(7) + ct_steps(5)>(pw, bf);
        mel_col_store<7, SCALED, FULL>(a7, n, lane, o);
        mel_col_store<5, SCALED, FULL>(a5, n, lane, o);
        mel_col_store<0, SCALED, FULL>(a0, n, lane, o);
    } else {
        const f32x4 a6 = mel_col_product<6, 0>(pw, bf);
        const f32x4 a4 = mel_col_product<4, ct_steps(6)>(pw, bf);
        const f32x4 a3 = mel_col_product<3, ct_steps(6) + ct_steps(4)>(pw, bf);
        const f32x4 a2 = mel_col_product<2, ct_steps(6) + ct_steps(4) + ct_steps(3)>(pw, bf);
        const f32x4 a1 = mel_col_product<1, ct_steps(6) + ct_steps(4) + ct_steps(3) + ct_steps(2)>(pw, bf);
        mel_col_store<6, SCALED, FULL>(a6, n, lane, o);
        mel_col_store<4, SCALED, FULL>(a4, n, lane, o);
        mel_col_store<3, SCALED, FULL>(a3, n, lane, o);
        mel_col_store<2, SCALED, FULL>(a2, n, lane, o);
        mel_col_store<1, SCALED, FULL>(a1, n, lane, o);
    }
}

// Complex values live in register pairs (re, im) and all butterfly arithmetic is packed fp32 (v_pk_add / v_pk_mul /
// v_pk_fma: two lanes of work per instruction, the only way the vector ALU reaches its fp32 rate); multiplications by
// +-i and complex products are single packed instructions whose operand halves are swapped / negated by op_sel / neg.
typedef float cpx __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cpx add_mi(cpx x, cpx y) {   // x + (-i) y = (x.r + y.i, x.i - y.r)
    cpx d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
    return d;
}
__device__ __forceinline__ cpx add_pi(cpx x, cpx y) {   // x + (+i) y = (x.r - y.i, x.i + y.r)
    cpx d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(x), "v"(y));
    return d;
}
__device__ __forceinline__ cpx cmul(cpx a, cpx w) {     // a * (w.x + i w.y) = (a.r w.x - a.i w.y, a.i w.x + a.r w.y)
    cpx t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
__device__ __forceinline__ cpx splat(float v) { return (cpx){v, v}; }

// Forward 5-point DFT in place (w = exp(-2 pi i / 5)).
__device__ __forceinline__ void dft5(cpx& x0, cpx& x1, cpx& x2, cpx& x3, cpx& x4) {
    const cpx C1 = splat(0.30901699437494745f), C2 = splat(-0.8090169943749473f), S1 = splat(0.9510565162951535f), S2 = splat(0.5877852522924732f);
    const cpx s1 = x1 + x4, s2 = x2 + x3, d1 = x1 - x4, d2 = x2 - x3;
    const cpx a = x0 + C1 * s1 + C2 * s2;
    const cpx b = x0 + C2 * s1 + C1 * s2;
    const cpx p = S1 * d1 + S2 * d2;   // X1 = a - i p, X4 = a + i p
    const cpx q = S2 * d1 - S1 * d2;   // X2 = b - i q, X3 = b + i q
    x0 = x0 + s1 + s2;
    x1 = add_mi(a, p);
    x4 = add_pi(a, p);
    x2 = add_mi(b, q);
    x3 = add_pi(b, q);
}

// W25^(b c) = exp(-2 pi i b c / 25) for b, c in 1..4 as (cos, -sin), rounded from double.
__device__ __forceinline__ cpx tw25(cpx a, int b, int c) {
    constexpr float C[4][4] = {{0.9685831611286311f, 0.8763066800438636f, 0.7289686274214116f, 0.5358267949789965f},
                               {0.8763066800438636f, 0.5358267949789965f, 0.06279051952931353f, -0.4257792915650727f},
                               {0.7289686274214116f, 0.06279051952931353f, -0.6374239897486897f, -0.9921147013144778f},
                               {0.5358267949789965f, -0.4257792915650727f, -0.9921147013144778f, -0.6374239897486895f}};
    constexpr float S[4][4] = {{-0.2486898871648548f, -0.4817536741017153f, -0.6845471059286886f, -0.8443279255020151f},
                               {-0.4817536741017153f, -0.8443279255020151f, -0.9980267284282716f, -0.9048270524660195f},
                               {-0.6845471059286886f, -0.9980267284282716f, -0.7705132427757893f, -0.12533323356430454f},
                               {-0.8443279255020151f, -0.9048270524660195f, -0.12533323356430454f, 0.7705132427757894f}};
    return cmul(a, (cpx){C[b - 1][c - 1], S[b - 1][c - 1]});
}

// Phase 1 of a tile, one item = (frame f, residue n2): eight windowed sample pairs, the radix-8 butterfly, the twiddles.
// A lane keeps its residue for the whole kernel, so its eight window pairs hw and seven twiddles tw live in registers.
__device__ __forceinline__ void bfly_load(const float* xs, int f, int n2, const cpx (&hw)[8], cpx (&z)[8]) {
    const cpx* xf = (const cpx*)(xs + f * MEL_HOP) + n2;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) z[n1] = xf[25 * n1] * hw[n1];
}
__device__ __forceinline__ void bfly_store(cpx* zc, int f, int n2, const cpx (&tw)[7], const cpx (&z)[8]) {
    const cpx a0 = z[0] + z[4], a1 = z[0] - z[4], a2 = z[2] + z[6], a3 = z[2] - z[6];
    const cpx b0 = z[1] + z[5], b1 = z[1] - z[5], b2 = z[3] + z[7], b3 = z[3] - z[7];
    const cpx c0 = a0 + a2, c2 = a0 - a2, c1 = add_mi(a1, a3), c3 = add_pi(a1, a3);
    const cpx e0 = b0 + b2, e2 = b0 - b2, e1 = add_mi(b1, b3), e3 = add_pi(b1, b3);
    const cpx R = splat(0.70710678118654752440f);
    const cpx w1 = R * add_mi(e1, e1);    // e1 * (1 - i) / sqrt 2
    const cpx w3 = R * add_pi(e3, e3);    // -e3 * (-1 - i) / sqrt 2
    cpx y[8];
    y[0] = c0 + e0; y[4] = c0 - e0;
    y[1] = c1 + w1; y[5] = c1 - w1;
    y[2] = add_mi(c2, e2); y[6] = add_pi(c2, e2);
    y[3] = c3 - w3; y[7] = c3 + w3;
    cpx* yo = zc + f * ZS + n2;
    yo[0] = y[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) yo[25 * k1] = cmul(y[k1], tw[k1 - 1]);
}

// Phase 3 of a tile, one item = (frame f, bin pair k / 200 - k).
__device__ __forceinline__ void untangle_load(const cpx* zc, const float* tab, int u, cpx& A, cpx& B, cpx& W) {
    const int f = u / 101, k = u - 101 * f;
    A = zc[f * ZS + k];
    B = zc[f * ZS + (k == 0 ? 0 : 200 - k)];
    W = ((const cpx*)tab)[k];
}
__device__ __forceinline__ void untangle_store(float* xs, int u, cpx A, cpx B, cpx W) {
    const int f = u / 101, k = u - 101 * f;
    cpx E2, O2;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(E2) : "v"(A), "v"(B));                               // (ar + br, ai - bi)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(O2) : "v"(A), "v"(B));  // (ai + bi, br - ar)
    const cpx T = cmul(O2, W);
    // X = E2 + T and Y = E2 - T, held as P = (X.re, Y.re) and Q = (X.im, Y.im): both squared magnitudes come out of
    // three packed instructions, each square rounded before the sum (no fused multiply-add: the rounding is part of
    // the result).  The factor 1/4 lives in the filterbank weights.
    cpx P, Q;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(P) : "v"(E2), "v"(T));   // (er + tr, er - tr)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(Q) : "v"(E2), "v"(T));   // (ei + ti, ei - ti)
    cpx PW;
    {
#pragma clang fp contract(off)
        const cpx P2 = P * P, Q2 = Q * Q;
        PW = P2 + Q2;
    }
    float* pw = xs + f * POW_STRIDE;
    pw[k] = PW.x;
    if (k > 0 && k < 100) pw[200 - k] = PW.y;
}

// Persistent workgroups, each looping over tiles of 32 frames.
template <bool SCALED>
__global__ __launch_bounds__(MEL_THREADS, 2) void mel_kernel(const float* __restrict__ audio, const int64_t* __restrict__ d_off,
                                                             const int64_t* __restrict__ d_len, const int* __restrict__ tile_utt,
                                                             const int* __restrict__ tile_f0, int n_tiles,
                                                             const float* __restrict__ tables,
                                                             const float* __restrict__ fb_frags, float* __restrict__ out,
                                                             int t_stride, unsigned* __restrict__ d_max, int vec_ok,
                                                             float* __restrict__ tile_min) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                           // samples of the tile; from phase 3 on the power rows [32][201]
    cpx* zc = (cpx*)(sm + XS_N);              // transform values [32][200], (re, im) pairs
    float* tab = sm + XS_N + 2 * Z_N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < TAB_LDS; i += MEL_THREADS) tab[i] = tables[TAB_UW + i];
    // phase-1 role of this thread: threads 0-249 take residue n2 = tid % 25 of frame 10 r + tid / 25 in rounds r = 0..2;
    // frames 30 and 31 go to lanes 0-49 of one wave (fifty consecutive threads hold every residue twice), a different
    // wave from tile to tile so that no SIMD always carries the extra round
    const int n2 = tid % 25, p1_fr = tid / 25;
    cpx hw[8], tw[7];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) hw[n1] = ((const cpx*)(tables + TAB_HANN))[25 * n1 + n2];
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) tw[k1 - 1] = ((const cpx*)(tables + TAB_TW))[25 * k1 + n2];   // [k1][n2] (cos, -sin)
    // this wave's filterbank fragments (B operand of the phase-4 products) stay in registers for the whole kernel
    float bf[SET_STEPS];
#pragma unroll
    for (int i = 0; i < SET_STEPS; ++i) bf[i] = fb_frags[((wave >> 1) * SET_STEPS + i) * 64 + lane];

    // Tile order.  Workgroups b, b + 8, b + 16, ... run on one XCD (round-robin dispatch) and share its L2.  Each XCD takes
    // one contiguous eighth of the tiles and deals it tile by tile to its workgroups, so that at any time an XCD works
    // on neighbouring tiles: consecutive tiles of an utterance share 240 samples of input, and their 128-byte pieces of
    // the 128 output rows meet in one L2 and leave it as long runs instead of isolated pieces 12 KB apart.
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, stride = ((int)gridDim.x - xcd + 7) >> 3;   // workgroups of this XCD
    const int per = (n_tiles + 7) >> 3;
    const int t_end = gridDim.x >= 8 ? min(n_tiles, (xcd + 1) * per) : n_tiles;
    float sv[NLOAD];
    int t = gridDim.x >= 8 ? xcd * per + local : (int)blockIdx.x;
    const int t_step = gridDim.x >= 8 ? stride : (int)gridDim.x;
    if (t < t_end) {
        fetch_samples(tile_info(audio, d_off, d_len, tile_utt, tile_f0, t), sv, tid);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) xs[tid + i * MEL_THREADS] = sv[i];
    }
    int tile_count = 0;
    for (; t < t_end; t += t_step) {
        const TileInfo cur = tile_info(audio, d_off, d_len, tile_utt, tile_f0, t);
        const int tn = t + t_step;
        __syncthreads();   // this tile's samples (and, the first time, the tables) are in LDS

        // the next tile's samples travel through registers while this tile is transformed
        if (tn < t_end) fetch_samples(tile_info(audio, d_off, d_len, tile_utt, tile_f0, tn), sv, tid);

        // ---- phase 1: window, radix-8 butterflies over n1 (z index m = 25 n1 + n2), twiddles W200^(n2 k1) ----
        {
            cpx z[8];
            if (tid < 250) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    bfly_load(xs, 10 * r + p1_fr, n2, hw, z);
                    bfly_store(zc, 10 * r + p1_fr, n2, tw, z);
                }
            }
            if (wave == (tile_count & 3) && lane < 50) {
                bfly_load(xs, 30 + (lane >= 25), n2, hw, z);
                bfly_store(zc, 30 + (lane >= 25), n2, tw, z);
            }
            ++tile_count;
        }
        __syncthreads();

        // ---- phase 2: one 25-point transform per (frame, k1): n2 = 5 a + b, k2 = c + 5 d.  Thread tid owns row
        //      f = tid / 8, k1 = tid % 8: rows lie 25 pairs apart and frames 200, so the 16 lanes an 8-byte LDS access
        //      serves together touch 16 distinct bank pairs. ----
        {
            const bool mine = tid < TF * 8;
            const int f = (tid >> 3) & (TF - 1), k1 = tid & 7;
            const cpx* yi = zc + f * ZS + 25 * k1;
            cpx v[5][5];   // [b][a], then [b][c]
            if (mine) {
#pragma unroll
                for (int b = 0; b < 5; ++b)
#pragma unroll
                    for (int a = 0; a < 5; ++a) v[b][a] = yi[5 * a + b];
            }
            __syncthreads();   // every row has been read: the values can be overwritten in place
            if (mine) {
#pragma unroll
            for (int b = 0; b < 5; ++b) dft5(v[b][0], v[b][1], v[b][2], v[b][3], v[b][4]);
#pragma unroll
            for (int b = 1; b < 5; ++b)
#pragma unroll
                for (int c = 1; c < 5; ++c) v[b][c] = tw25(v[b][c], b, c);
            cpx* zo = zc + f * ZS + k1;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                dft5(v[0][c], v[1][c], v[2][c], v[3][c], v[4][c]);   // over b: entry d holds Z[k1 + 8 (c + 5 d)]
#pragma unroll
                for (int d = 0; d < 5; ++d) zo[8 * (c + 5 * d)] = v[d][c];
            }
            }
        }
        __syncthreads();

        // ---- phase 3: real spectrum from A = Z[k] and B = Z[200 - k]: with E2 = A + conj B and O2 = -i (A - conj B),
        //      X[k] = (E2 + w O2) / 2 and conj X[200 - k] = (E2 - w O2) / 2, w = exp(-i pi k / 200).  Bin 200 feeds no filter.
        //      3232 items: twelve rounds of 256 in groups of four with all reads first, then the 160 left over. ----
        for (int it = 0; it < 12; it += 4) {
            cpx A[4], B[4], W[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) untangle_load(zc, tab, tid + (it + j) * MEL_THREADS, A[j], B[j], W[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) untangle_store(xs, tid + (it + j) * MEL_THREADS, A[j], B[j], W[j]);
        }
        if (tid + 12 * MEL_THREADS < TF * 101) {
            cpx A, B, W;
            untangle_load(zc, tab, tid + 12 * MEL_THREADS, A, B, W);
            untangle_store(xs, tid + 12 * MEL_THREADS, A, B, W);
        }
        __syncthreads();

        // ---- phase 4: slaney filterbank (banded fp32 MFMA), log10, running maximum, stores.  An accumulator holds, per
        //      lane, one filter's values for four consecutive frames: one 16-byte store per lane. ----
        {
            const int frames_total = (int)(cur.n / MEL_HOP) + 1;
            const int frames_out = min(frames_total - 1, MEL_MAX_FRAMES);
            const int n = lane & 15, g = lane >> 4, rt = wave & 1;
            const int frame0 = cur.f0 + 16 * rt + 4 * g;
            MelTileOut o;
            o.out = out + (int64_t)cur.u * MEL_NMELS * t_stride + frame0;
            o.t_stride = t_stride;
            o.n_store = min(max(frames_out - frame0, 0), 4);
            // running maximum over every frame the reference computes, the dropped last one included (:269-270)
            o.n_max = min(max(frames_total - frame0, 0), 4);
            o.vec = vec_ok;
            o.vmax = -INFINITY;
            o.cell_min = SCALED ? tile_min + ((int64_t)t * 2 + rt) * 8 : nullptr;
            const float* pw = xs + (16 * rt + n) * POW_STRIDE + g;   // A operand: row = frame, k = bin
            if (vec_ok && cur.f0 + TF <= frames_out) mel_filterbank<SCALED, true>(pw, bf, wave, n, lane, o);   // workgroup-uniform
            else mel_filterbank<SCALED, false>(pw, bf, wave, n, lane, o);
            const float vmax = wave_max(o.vmax);
            if (lane == 0 && vmax > -INFINITY) atomicMax(d_max + cur.u, enc_max(vmax));
        }
        __syncthreads();   // the power rows have been consumed: the region takes the next tile's samples
        if (tn < t_end) {
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) xs[tid + i * MEL_THREADS] = sv[i];
        }
    }
}

__global__ void mel_normalize_kernel(float* __restrict__ out, int t_stride, const int* __restrict__ frames_out,
                                     const unsigned* __restrict__ d_max) {
    const int u = blockIdx.y, m = blockIdx.x;
    const int t = frames_out[u];
    const float floor_v = dec_max(d_max[u]) - 8.0f;
    float* row = out + ((int64_t)u * MEL_NMELS + m) * t_stride;
    for (int i = threadIdx.x; i < t; i += blockDim.x) row[i] = (fmaxf(row[i], floor_v) + 4.0f) / 4.0f;
}

// The clamp of rows that were stored scaled: values below (max - 8 + 4) / 4 are raised to it (AudioPreprocessing.swift:269-276).
// tile_min holds, per tile, the minimum of each cell of 16 frames x 16 mel rows ([2][8] per tile).  One wave per tile:
// it leaves at once when no cell's minimum is below the utterance's floor, and otherwise visits the cells that are,
// one 16-byte piece (four frames of one row) per lane.
__global__ __launch_bounds__(256) void mel_clamp_kernel(float* __restrict__ out, int t_stride, int n_tiles,
                                                        const int* __restrict__ tile_utt, const int* __restrict__ tile_f0,
                                                        const int* __restrict__ frames_out, const unsigned* __restrict__ d_max,
                                                        const float* __restrict__ tile_min, int vec_ok) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const int u = tile_utt[t], f0 = tile_f0[t];
    const float floor_v = dec_max(d_max[u]) - 8.0f;
    const float mn = lane < 16 ? tile_min[(int64_t)t * 16 + lane] : INFINITY;
    unsigned long long flagged = __ballot(mn < floor_v);
    if (!flagged) return;
    const float floor_s = (floor_v + 4.0f) * 0.25f;
    const int frames = frames_out[u], row = lane >> 2, q = lane & 3;
    while (flagged) {
        const int cell = __builtin_ctzll(flagged);   // cell = 8 rt + ct
        flagged &= flagged - 1;
        const int fr = f0 + 16 * (cell >> 3) + 4 * q;
        float* p = out + ((int64_t)u * MEL_NMELS + 16 * (cell & 7) + row) * t_stride + fr;
        const int nv = min(max(frames - fr, 0), 4);
        if (vec_ok && nv == 4) {
            const float4 v = *(const float4*)p;
            if (fminf(fminf(v.x, v.y), fminf(v.z, v.w)) < floor_s)
                *(float4*)p = make_float4(fmaxf(v.x, floor_s), fmaxf(v.y, floor_s), fmaxf(v.z, floor_s), fmaxf(v.w, floor_s));
        } else {
            for (int r = 0; r < nv; ++r)
                if (p[r] < floor_s) p[r] = floor_s;
        }
    }
}

// Tile list of a batch: utterance u owns tiles [tile0[u], tile0[u + 1]).
__global__ void mel_tiles_kernel(const int* __restrict__ tile0, int* __restrict__ tile_utt, int* __restrict__ tile_f0) {
    const int u = blockIdx.x, first = tile0[u], last = tile0[u + 1];
    for (int i = first + threadIdx.x; i < last; i += blockDim.x) {
        tile_utt[i] = u;
        tile_f0[i] = (i - first) * MEL_TILE_FRAMES;
    }
}

// AudioPreprocessing.swift:347-365: index in double, fraction and blend in float, no contraction.
__global__ void resample_kernel(const float* __restrict__ in, int64_t n_in, float* __restrict__ out, int64_t n_out,
                                double ratio) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const double src = (double)i / ratio;
    const int64_t fl = (int64_t)src;
    const int64_t ce = min(fl + 1, n_in - 1);
    const float frac = (float)(src - (double)fl);
    out[i] = __fadd_rn(__fmul_rn(in[fl], __fsub_rn(1.0f, frac)), __fmul_rn(in[ce], frac));
}

template <class T>
T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    HIP_CHECK(hipMalloc(&d, v.size() * sizeof(T)));
    HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

// Slaney filterbank in fp32, following AudioPreprocessing.swift:23-126 operation by operation.
std::vector<float> build_filterbank() {
    const float f_max = 16000.0f / 2.0f, min_log_hz = 1000.0f, min_log_mel = 15.0f;
    const float hz2mel = 27.0f / logf(6.4f), mel2hz = logf(6.4f) / 27.0f;
    auto hz_to_mel = [&](float hz) { return hz < min_log_hz ? 3.0f * hz / 200.0f : min_log_mel + logf(hz / min_log_hz) * hz2mel; };
    auto mel_to_hz = [&](float mel) { return mel < min_log_mel ? 200.0f * mel / 3.0f : min_log_hz * expf((mel - min_log_mel) * mel2hz); };
    std::vector<float> fft(MEL_NBINS), filt(MEL_NMELS + 2), diff(MEL_NMELS + 1);
    for (int i = 0; i < MEL_NBINS; ++i) fft[i] = (float)i * f_max / (float)(MEL_NBINS - 1);
    const float mel_min = hz_to_mel(0.0f), mel_max = hz_to_mel(f_max);
    for (int i = 0; i < MEL_NMELS + 2; ++i)
        filt[i] = mel_to_hz(mel_min + (float)i * (mel_max - mel_min) / (float)(MEL_NMELS + 1));
    for (int i = 0; i < MEL_NMELS + 1; ++i) diff[i] = filt[i + 1] - filt[i];
    std::vector<float> fb((size_t)MEL_NMELS * MEL_NBINS);
    for (int m = 0; m < MEL_NMELS; ++m) {
        const float enorm = 2.0f / (filt[m + 2] - filt[m]);
        for (int b = 0; b < MEL_NBINS; ++b) {
            const float down = (fft[b] - filt[m]) / diff[m], up = (filt[m + 2] - fft[b]) / diff[m + 1];
            fb[(size_t)m * MEL_NBINS + b] = fmaxf(0.0f, fminf(down, up)) * enorm;
        }
    }
    return fb;
}

}  // namespace

void mel_tables_create(MelTables& t) {
    const double PI = 3.14159265358979323846;
    std::vector<float> tab(TAB_N, 0.f);
    // periodic Hann window, all 400 taps evaluated in fp32 like the reference (AudioPreprocessing.swift:161-164)
    for (int i = 0; i < MEL_NFFT; ++i) tab[TAB_HANN + i] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)i / (float)MEL_NFFT));
    // twiddles between the radix-8 and the 25-point stage, W200^(n2 k1), and of the real-spectrum untangling,
    // exp(-i pi k / 200): angles reduced and evaluated in double, i.e. the mathematically exact values rounded once
    for (int n2 = 0; n2 < 25; ++n2)
        for (int k1 = 0; k1 < 8; ++k1) {
            const double ang = 2.0 * PI * (double)((n2 * k1) % 200) / 200.0;
            tab[TAB_TW + 2 * (k1 * 25 + n2)] = (float)cos(ang);
            tab[TAB_TW + 2 * (k1 * 25 + n2) + 1] = (float)-sin(ang);
        }
    for (int k = 0; k <= 100; ++k) {
        tab[TAB_UW + 2 * k] = (float)cos(PI * (double)k / 200.0);
        tab[TAB_UW + 2 * k + 1] = (float)-sin(PI * (double)k / 200.0);
    }
    t.twid = upload(tab);

    t.host_fb = build_filterbank();
    // The kernel's compile-time filter supports must cover every weight that matters; anything outside them has to be
    // rounding dust of the triangle edges (none with the libm this was built against).
    for (int m = 0; m < MEL_NMELS; ++m)
        for (int b = 0; b < MEL_NBINS; ++b)
            if (!(b >= FB_LO[m] && b < FB_LO[m] + FB_W[m]))
                Q3_REQUIRE(fabsf(t.host_fb[(size_t)m * MEL_NBINS + b]) < 1e-9f, "mel: the filterbank has a weight outside the kernel's filter supports");
    // Weight fragments of the banded products: lane (n = lane & 15, kg = lane >> 4) of step s of column tile ct holds
    // W[bin = ct_lo + 4 s + kg][filter 16 ct + n] / 4 (an exact scaling: the power rows hold four times the power), exactly zero outside the filter's support (so whatever the kernel
    // reads beside a row's bins contributes nothing).
    std::vector<float> frags((size_t)2 * SET_STEPS * 64, 0.f);
    for (int set = 0; set < 2; ++set) {
        int pos = 0;
        const int n_ct = set == 0 ? 3 : 5;
        for (int j = 0; j < n_ct; ++j) {
            const int ct = set == 0 ? SET_A[j] : SET_B[j];
            for (int st = 0; st < ct_steps(ct); ++st, ++pos)
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = 16 * ct + (lane & 15), bin = ct_lo(ct) + 4 * st + (lane >> 4);
                    if (bin >= FB_LO[m] && bin < FB_LO[m] + FB_W[m])
                        frags[((size_t)set * SET_STEPS + pos) * 64 + lane] = 0.25f * t.host_fb[(size_t)m * MEL_NBINS + bin];   // phase 3 leaves 4 |X|^2
                }
        }
        Q3_REQUIRE(pos == SET_STEPS, "mel: filter set size mismatch");
    }
    t.fb_packed = upload(frags);
}

void mel_tables_destroy(MelTables& t) {
    (void)hipFree(t.twid); (void)hipFree(t.fb_packed);
    t = MelTables();
}

void launch_resample(const float* in, int64_t n_in, float* out, int64_t n_out, double ratio, hipStream_t s) {
    if (n_out <= 0) return;
    hipLaunchKernelGGL(resample_kernel, dim3(cdiv(n_out, 256)), dim3(256), 0, s, in, n_in, out, n_out, ratio);
    HIP_CHECK(hipGetLastError());
}

static void mel_kernel_attributes() {
    HIP_CHECK(hipFuncSetAttribute((const void*)mel_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MEL_LDS_BYTES));
    HIP_CHECK(hipFuncSetAttribute((const void*)mel_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MEL_LDS_BYTES));
}

void launch_mel(const MelTables& t, const float* audio, const int64_t* d_off, const int64_t* d_len, const int* d_tile_utt,
                const int* d_tile_f0, int n_tiles, float* out, int t_stride, unsigned* d_max, hipStream_t s, double work,
                float* d_tile_min) {
    if (n_tiles <= 0) return;
    static PerDeviceOnce once;
    once.run([&] { mel_kernel_attributes(); });
    const int cus = device_cus();
    ProfScope ps("mel_kernel", work, s);
    // two workgroups fit per CU (79 KB of LDS each); each loops over its share of the tiles
    hipLaunchKernelGGL(d_tile_min ? mel_kernel<true> : mel_kernel<false>, dim3(std::min(n_tiles, 2 * cus)), dim3(MEL_THREADS), MEL_LDS_BYTES, s, audio, d_off, d_len,
                       d_tile_utt, d_tile_f0, n_tiles, t.twid, t.fb_packed, out, t_stride, d_max,
                       (t_stride % 4 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0,   // 16-byte stores need aligned rows
                       d_tile_min);
    HIP_CHECK(hipGetLastError());
}

void launch_mel_tiles(const int* d_tile0, int B, int* d_tile_utt, int* d_tile_f0, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(mel_tiles_kernel, dim3(B), dim3(64), 0, s, d_tile0, d_tile_utt, d_tile_f0);
    HIP_CHECK(hipGetLastError());
}

void launch_mel_clamp(float* out, int t_stride, int n_tiles, const int* d_tile_utt, const int* d_tile_f0, const int* d_frames_out,
                      const unsigned* d_max, const float* d_tile_min, hipStream_t s) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(mel_clamp_kernel, dim3(cdiv(n_tiles, 4)), dim3(256), 0, s, out, t_stride, n_tiles, d_tile_utt, d_tile_f0,
                       d_frames_out, d_max, d_tile_min, (t_stride % 4 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0);
    HIP_CHECK(hipGetLastError());
}

void launch_mel_normalize(float* out, int B, int t_stride, const int* d_frames_out, const unsigned* d_max, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(mel_normalize_kernel, dim3(MEL_NMELS, B), dim3(256), 0, s, out, t_stride, d_frames_out, d_max);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// Log-mel without any allocation, for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses
// that forbid it).  Same kernels as the allocating path; what that path takes from the model's workspace (utterance
// offsets and lengths on the device, the tile list, the per-utterance running maximum, the frame counts) lives in a
// fixed file-scope device array instead.  Every GPU holds its own copy of the array at its own address until the
// process ends, and all calls on one GPU share it, whichever model or stream they come through.  The library orders
// them: a per-device host mutex is held while a call's kernels are enqueued, and an event recorded behind a call's
// last kernel is what the next call's stream waits for before its set-up kernel rewrites the array -- so asynchronous
// calls on different streams, and calls through different models, run one after the other on the device.  A stream
// under graph capture can neither wait for an event outside its graph nor record one: calls issued while capturing
// are ordered only among the nodes of their own graph, and replays of such graphs must not overlap each other or
// any other user of this entry point on the same GPU.
namespace {

constexpr int NOALLOC_MAX_UTT = MEL_NOALLOC_MAX_UTT;
constexpr int NOALLOC_MAX_TILES = NOALLOC_MAX_UTT * ((MEL_MAX_FRAMES + 1 + MEL_TILE_FRAMES - 1) / MEL_TILE_FRAMES);   // 94 per 30 s clip

struct NoAllocScratch {
    int64_t off[NOALLOC_MAX_UTT];
    int64_t len[NOALLOC_MAX_UTT];
    unsigned dmax[NOALLOC_MAX_UTT];
    int frames[NOALLOC_MAX_UTT];
    int tile_utt[NOALLOC_MAX_TILES];
    int tile_f0[NOALLOC_MAX_TILES];
    float tile_min[NOALLOC_MAX_TILES * 16];   // per tile: minima of its [2][8] cells of 16 frames x 16 mel rows
};
static_assert(sizeof(NoAllocScratch) <= 1800 * 1024, "scratch is meant to stay modest");

// 1.7 MB per GPU.  In host code this name is only a handle for hipGetSymbolAddress: the device address is looked up
// once per device and every use below goes through that pointer.
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

PerDeviceOnce g_scratch_once;
NoAllocScratch* g_scratch_ptr[PerDeviceOnce::MAX_DEVICES];
// per device: the last call's "scratch free again" event, whether it has ever been recorded, and the enqueue mutex
// (created once at set-up, never destroyed: they live as long as the array they guard)
hipEvent_t g_scratch_free[PerDeviceOnce::MAX_DEVICES];
bool g_scratch_used[PerDeviceOnce::MAX_DEVICES];
std::mutex* g_scratch_mu = new std::mutex[PerDeviceOnce::MAX_DEVICES];

struct OffsetsArg {   // utterance boundaries by value in the kernel arguments: no host-to-device copy is needed
    int64_t v[NOALLOC_MAX_UTT + 1];
};

// Fills the scratch for one call: offsets, lengths, frame counts, a zeroed maximum, and the tile list in utterance
// order (the same list the allocating path builds on the host).
__global__ __launch_bounds__(256) void mel_noalloc_setup_kernel(OffsetsArg offs, int n_utt, NoAllocScratch* sc) {
    __shared__ int tile0[NOALLOC_MAX_UTT + 1];
    const int tid = threadIdx.x;
    for (int u = tid; u < n_utt; u += 256) {
        const int64_t n = offs.v[u + 1] - offs.v[u];
        sc->off[u] = offs.v[u];
        sc->len[u] = n;
        sc->dmax[u] = 0u;
        int t = (int)(n / MEL_HOP);                      // frames returned: min(n / 160, 3000)
        sc->frames[u] = t > MEL_MAX_FRAMES ? MEL_MAX_FRAMES : t;
    }
    if (tid == 0) {
        int acc = 0;
        for (int u = 0; u < n_utt; ++u) {
            tile0[u] = acc;
            const int total = (int)((offs.v[u + 1] - offs.v[u]) / MEL_HOP) + 1;
            acc += (total + MEL_TILE_FRAMES - 1) / MEL_TILE_FRAMES;
        }
        tile0[n_utt] = acc;
    }
    __syncthreads();
    for (int u = 0; u < n_utt; ++u)
        for (int i = tile0[u] + tid; i < tile0[u + 1]; i += 256) {
            sc->tile_utt[i] = u;
            sc->tile_f0[i] = (i - tile0[u]) * MEL_TILE_FRAMES;
        }
}

}  // namespace

void mel_noalloc_init() {
    g_scratch_once.run([&] {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        void* p = nullptr;
        HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_scratch_ptr[dev] = (NoAllocScratch*)p;
        HIP_CHECK(hipEventCreateWithFlags(&g_scratch_free[dev], hipEventDisableTiming));
        mel_kernel_attributes();
    });
}

void mel_noalloc_run(const MelTables& t, const float* audio_dev, const int64_t* offsets_host, int n_utt, float* mel_out_dev,
                     int t_stride, int32_t* n_frames_host, hipStream_t s) {
    Q3_REQUIRE(n_utt >= 1 && n_utt <= NOALLOC_MAX_UTT,
               "mel_noalloc: between 1 and " + std::to_string(NOALLOC_MAX_UTT) + " utterances per call (fixed scratch)");
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    NoAllocScratch* sc = g_scratch_ptr[dev];
    Q3_REQUIRE(sc != nullptr, "mel_noalloc: the scratch address of this device was not set up");
    OffsetsArg offs;
    int64_t tiles = 0;
    int tmax = 1;
    double bytes = 0;
    offs.v[0] = offsets_host[0];
    for (int u = 0; u < n_utt; ++u) {
        offs.v[u + 1] = offsets_host[u + 1];
        const int64_t n = offsets_host[u + 1] - offsets_host[u];
        Q3_REQUIRE(n >= 0, "mel_noalloc: offsets must be non-decreasing");
        Q3_REQUIRE(n <= (int64_t)MEL_NOALLOC_MAX_SAMPLES, "mel_noalloc: an utterance is longer than " +
                                                              std::to_string(MEL_NOALLOC_MAX_SAMPLES / 16000) + " s (fixed scratch)");
        tiles += (mel_frames_total(n) + MEL_TILE_FRAMES - 1) / MEL_TILE_FRAMES;
        n_frames_host[u] = mel_frames_out(n);
        tmax = std::max(tmax, mel_frames_out(n));
        bytes += 4.0 * (double)n + 4.0 * MEL_NMELS * (double)mel_frames_out(n);
    }
    Q3_REQUIRE(tiles <= NOALLOC_MAX_TILES, "mel_noalloc: too much audio for the fixed scratch (at most " +
                                               std::to_string(NOALLOC_MAX_TILES) + " tiles of 32 frames per call)");
    Q3_REQUIRE(t_stride >= tmax, "mel_noalloc: t_stride is smaller than the longest utterance's frame count");
    std::lock_guard<std::mutex> enqueue(g_scratch_mu[dev]);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    HIP_CHECK(hipStreamIsCapturing(s, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && g_scratch_used[dev]) HIP_CHECK(hipStreamWaitEvent(s, g_scratch_free[dev], 0));   // the previous user has finished
    hipLaunchKernelGGL(mel_noalloc_setup_kernel, dim3(1), dim3(256), 0, s, offs, n_utt, sc);
    HIP_CHECK(hipGetLastError());
    launch_mel(t, audio_dev, sc->off, sc->len, sc->tile_utt, sc->tile_f0, (int)tiles, mel_out_dev, t_stride, sc->dmax, s, bytes, sc->tile_min);
    launch_mel_clamp(mel_out_dev, t_stride, (int)tiles, sc->tile_utt, sc->tile_f0, sc->frames, sc->dmax, sc->tile_min, s);
    if (!capturing) {
        HIP_CHECK(hipEventRecord(g_scratch_free[dev], s));
        g_scratch_used[dev] = true;
    }
}
