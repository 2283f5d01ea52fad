// Log-mel frontend for gfx950.
//
// Restates WhisperFeatureExtractor (reference: Sources/Qwen3ASR/AudioPreprocessing.swift:131-324).
// Work is cut into tiles of 48 frames of one utterance; persistent workgroups (one per CU) loop over the tiles.
// The 7 920 samples a tile covers are staged once in LDS (coalesced, reflect padding resolved on load).  Because the
// frame is real, the 400-point DFT folds to Re X[k] = sum_{n<=200} h[n](x[n] + x[400-n]) cos(2 pi k n / 400) and
// Im X[k] = -sum h[n](x[n] - x[400-n]) sin(...); splitting those sums into even and odd n gives four
// [frames x 101] x [101 x 101] real products (Ce, Co, Se, So over bins 0..100) from which bin k is
// (Ce + Co, Se + So) and bin 200 - k is (Ce - Co, Se - So): a quarter of the direct DFT's multiplies,
// evaluated on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, an fp32 fma chain) with the Hann window folded into the
// twiddle tables.  The difference form loses accuracy only where bin 200 - k is far weaker than bin k: relative error
// ~6e-8 * |X[k]| / |X[200-k]|, i.e. < 1e-5 of log-mel for a 60 dB spread between the halves of the spectrum.
// Seven waves transform (one 16-bin tile each, twiddles resident in registers); an eighth wave applies the slaney
// filterbank, log10 and the per-utterance running maximum (wave reduction + one atomic per tile) to the previous
// tile's power spectrum and stages the next tile's samples, so neither phase waits for the other.  Output is the raw
// log10 mel; the max-8 clamp and (x+4)/4 scale need the utterance-wide maximum and are applied by the consumer
// (conv1) or by launch_mel_normalize for callers that want the reference's exact output tensor.
#include "mel.h"
#include "prof.h"
#include <algorithm>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

namespace {

constexpr int KSTEPS = 26;     // 104 >= 101 folded taps of each parity, 4 per MFMA
constexpr int NTILES = 7;      // 112 >= 101 bins
constexpr int MTILES = MEL_TILE_FRAMES / 16;
constexpr int MEL_THREADS = 64 * (NTILES + 1);  // one transform wave per 16-bin tile + one filterbank / staging wave
constexpr int SAMPLES = MEL_TILE_FRAMES * MEL_HOP + (MEL_NFFT - MEL_HOP) + 1;  // 7921
constexpr int SREG = 8032;     // skewed sample image: skew(7920) = 8017
constexpr int POW_STRIDE = MEL_NBINS;
constexpr int POW_N = MEL_TILE_FRAMES * POW_STRIDE + 16;
constexpr int HANN_N = 208;
constexpr int MEL_LDS_BYTES = (2 * SREG + 2 * POW_N) * 4;  // two sample images, two power spectra: 142 KB

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
constexpr int fb_offset(int m) {  // position of filter m's first weight in the packed weight list
    int o = 0;
    for (int i = 0; i < m; ++i) o += FB_W[i];
    return o;
}
constexpr int FB_TOTAL = fb_offset(MEL_NMELS);       // 394
constexpr int FB_REGS = (FB_TOTAL + 63) / 64;        // packed weights, one per lane per register

__device__ __forceinline__ int frame_skew(int q) { return 4 * (q >> 1) + (q & 1); }
// Frame starts (multiples of 160 samples) land on distinct banks for the stride-2 fragment reads.
__device__ __forceinline__ int skew(int p) { return p + frame_skew(p / MEL_HOP); }

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

__device__ __forceinline__ bool tile_is_interior(const TileInfo& ti) {  // no padding inside the tile
    const int64_t P0 = (int64_t)ti.f0 * MEL_HOP;
    return P0 >= 200 && P0 + SAMPLES <= ti.n + 200;
}

// Interior tile: loads this thread's share (samples id, id + STRIDE, ...) into registers; all loads are issued before
// anything waits on them, and write_tile stores them into a skewed LDS image later.
template <int N, int STRIDE, int I0>
__device__ __forceinline__ void fetch_tile(const TileInfo& ti, float (&sv)[N], int id) {
    const float* xs = ti.x + ((int64_t)ti.f0 * MEL_HOP - 200);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int p = id + (I0 + i) * STRIDE;
        sv[i] = ((I0 + i + 1) * STRIDE <= SAMPLES || p < SAMPLES) ? xs[p] : 0.f;
    }
}

template <int N, int STRIDE, int I0>
__device__ __forceinline__ void write_tile(const float (&sv)[N], float* dst, int id) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int p = id + (I0 + i) * STRIDE;
        if ((I0 + i + 1) * STRIDE <= SAMPLES || p < SAMPLES) dst[skew(p)] = sv[i];
    }
}

// Tile that touches the reflect padding of 200 samples or the end of the utterance (AudioPreprocessing.swift:134-154):
// staged directly, BATCH loads at a time.
template <int BATCH>
__device__ __forceinline__ void stage_edge_tile(const TileInfo& ti, float* dst, int id, int stride) {
    const int64_t P0 = (int64_t)ti.f0 * MEL_HOP, n = ti.n;
    for (int p0 = id; p0 < SAMPLES; p0 += BATCH * stride) {
        float v[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int64_t P = P0 + p0 + i * stride;
            int64_t idx = P < 200 ? 200 - P : (P < 200 + n ? P - 200 : 2 * n + 198 - P);
            idx = max(min(idx, n - 1), (int64_t)0);
            v[i] = (n > 0 && P < n + MEL_NFFT) ? ti.x[idx] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int p = p0 + i * stride;
            if (p < SAMPLES) dst[skew(p)] = v[i];
        }
    }
}

// Filter M applied to the power spectrum row of this lane's frame; weights come from the lane-packed registers.
template <int M>
__device__ __forceinline__ float mel_filter(const float* pw, const float (&fbv)[FB_REGS]) {
    constexpr int lo = FB_LO[M], w = FB_W[M], off = fb_offset(M);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < w; ++c) {
        const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fbv[(off + c) >> 6]), (off + c) & 63));
        sum = fmaf(pw[lo + c], wt, sum);
    }
    return sum;
}

constexpr int MEL_GROUP = 16;  // filters evaluated together: straight-line code, one predicated burst of stores

template <int M0, int J>
__device__ __forceinline__ void mel_values(const float* pw, const float (&fbv)[FB_REGS], float (&v)[MEL_GROUP]) {
    if constexpr (J < MEL_GROUP) {
        // log10 of the clamped filter output (AudioPreprocessing.swift:232-255).  The hardware log2 is accurate to 1 ulp
        // and the argument is never denormal, so the plain product is within 2e-6 of the correctly rounded value.
        v[J] = __log2f(fmaxf(mel_filter<M0 + J>(pw, fbv), 1e-10f)) * 0.30102999566398120f;
        mel_values<M0, J + 1>(pw, fbv, v);
    }
}

struct MelOut {
    float* out;     // this lane's column of the utterance's [128][t_stride] block
    int t_stride;
    bool store;
    float vmax;
};

template <int M0, int M1>
__device__ __forceinline__ void mel_rows(const float* pw, const float (&fbv)[FB_REGS], MelOut& o) {
    if constexpr (M0 < M1) {
        float v[MEL_GROUP];
        mel_values<M0, 0>(pw, fbv, v);
#pragma unroll
        for (int j = 0; j < MEL_GROUP; ++j) o.vmax = fmaxf(o.vmax, v[j]);
        if (o.store) {
#pragma unroll
            for (int j = 0; j < MEL_GROUP; ++j) o.out[(int64_t)(M0 + j) * o.t_stride] = v[j];
        }
        mel_rows<M0 + MEL_GROUP, M1>(pw, fbv, o);
    }
}

// Persistent workgroups: each takes tiles blockIdx.x, blockIdx.x + gridDim.x, ... of 48 frames.  Waves 0-6 transform: wave w
// owns bins 16w .. 16w+15 of all four products and keeps their twiddles (4 x 26 fragments) in registers for the whole kernel,
// so the MFMA loop reads only LDS.  Wave 7 meanwhile turns the previous tile's power spectrum into log-mel rows and stages the
// next tile's samples; sample image and power spectrum are double buffered and one barrier per tile hands them over.
__global__ __launch_bounds__(MEL_THREADS) void mel_kernel(const float* __restrict__ audio, const int64_t* __restrict__ d_off,
                                                          const int64_t* __restrict__ d_len, const int* __restrict__ tile_utt,
                                                          const int* __restrict__ tile_f0, int n_tiles,
                                                          const float* __restrict__ twid,
                                                          const float* __restrict__ fb_packed, float* __restrict__ out,
                                                          int t_stride, unsigned* __restrict__ d_max) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pwr = sm + 2 * SREG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int t = blockIdx.x;
    if (t < n_tiles) {
        stage_edge_tile<4>(tile_info(audio, d_off, d_len, tile_utt, tile_f0, t), sm, tid, MEL_THREADS);
    }

    if (wave < NTILES) {
        // ================= transform waves =================
        const int li = lane & 15, lg = lane >> 4;
        float tw[4][KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) tw[pr][s] = twid[((s * 4 + pr) * NTILES + wave) * 64 + lane];
        const int k = 16 * wave + li;
        int buf = 0;
        for (; t < n_tiles; t += gridDim.x) {
            __syncthreads();  // this tile's samples are staged; the power buffer written below has been consumed
            float* pw = pwr + buf * POW_N;
            // Tap e of frame fr sits at fr * 160 + e + frame_skew(fr + e / 160); e / 160 is fixed per step except at step 10.
            // Forward taps (e, e + 1) are read upwards from pa*, mirrored taps (399 - e, 400 - e) from pb*, whose origin
            // is the lowest mirrored tap.  The next 16 frames lie 16 * 160 + frame_skew(16) floats further on.
            constexpr int MIRROR0 = MEL_NFFT - 1 - 8 * (KSTEPS - 1);
            constexpr int ROWTILE = 16 * MEL_HOP + 32;
            const int row0 = buf * SREG + li * MEL_HOP;
            int qa0 = row0 + frame_skew(li) + 2 * lg;                 // taps e < 160
            int qa1 = row0 + frame_skew(li + 1) + 2 * lg;             // taps e >= 160
            int qb2 = row0 + frame_skew(li + 2) + MIRROR0 - 2 * lg;   // mirrored taps 400 - e >= 320
            int qb1 = row0 + frame_skew(li + 1) + MIRROR0 - 2 * lg;   // mirrored taps 400 - e < 320
            int qbe = lg == 0 ? qb2 : qb1;                            // step 10: e = 80 mirrors to 320, e > 80 below it
            // opaque to the compiler: keeps every read an immediate offset from these five registers
            asm volatile("" : "+v"(qa0), "+v"(qa1), "+v"(qb2), "+v"(qb1), "+v"(qbe));
            // operands of step g + 2 are read from LDS before the MFMAs of step g are issued (g runs over all row tiles)
            f32x2 fw[3], mr[3];  // (x[e], x[e+1]) and (x[399-e], x[400-e])
            auto fetch = [&](int g) {
                const int st = g % KSTEPS;
                if (st == 0 && g > 0) {
                    qa0 += ROWTILE, qa1 += ROWTILE, qb2 += ROWTILE, qb1 += ROWTILE, qbe += ROWTILE;
                    asm volatile("" : "+v"(qa0), "+v"(qa1), "+v"(qb2), "+v"(qb1), "+v"(qbe));
                }
                const int pa = st < 20 ? qa0 : qa1;
                const int pb = st < 10 ? qb2 : (st == 10 ? qbe : qb1);
                const int pd = st < 10 ? qb2 : qb1;
                fw[g % 3] = (f32x2){sm[pa + 8 * st], sm[pa + 8 * st + 1]};
                mr[g % 3] = (f32x2){sm[pd + 8 * (KSTEPS - 1 - st)], sm[pb + 8 * (KSTEPS - 1 - st) + 1]};
            };
            // power spectrum |X[k]|^2 (AudioPreprocessing.swift:207): bins k and 200-k from the same four sums
            auto power = [&](const f32x4 (&acc)[4], int mt) {
                if (k <= 100) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * lg + r;
                        const float ce = acc[0][r], co = acc[1][r], se = acc[2][r], so = acc[3][r];
                        const float re = ce + co, im = se + so;
                        pw[row * POW_STRIDE + k] = re * re + im * im;
                        if (k < 100) {
                            const float re2 = ce - co, im2 = se - so;
                            pw[row * POW_STRIDE + 200 - k] = re2 * re2 + im2 * im2;
                        }
                    }
                }
            };
            {
                f32x4 acc[2][4];  // two sets: a row tile's power spectrum is written while the next tile's MFMAs run
                fetch(0);
                fetch(1);
#pragma unroll
                for (int mt = 0; mt < MTILES; ++mt)
#pragma unroll
                for (int st = 0; st < KSTEPS; ++st) {
                    const int g = mt * KSTEPS + st;
                    // (x[e] + x[400-e], x[e+1] + x[399-e]) and the differences: one packed instruction each,
                    // the mirrored pair taken with its halves swapped
                    f32x2 sum, dif;
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(sum) : "v"(fw[g % 3]), "v"(mr[g % 3]));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                        : "=v"(dif) : "v"(fw[g % 3]), "v"(mr[g % 3]));
                    if (g + 2 < MTILES * KSTEPS) fetch(g + 2);
                    __builtin_amdgcn_sched_barrier(0);  // keeps those reads ahead of this step's arithmetic
                    // the window is part of the twiddles: even / odd samples, cosine / sine part
                    const float av[4] = {sum[0], sum[1], dif[0], dif[1]};
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {
                        const f32x4 c = st == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mt & 1][pr];
                        acc[mt & 1][pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[pr], tw[pr][st], c, 0, 0, 0);
                    }
                    if (st == 2 && mt > 0) power(acc[(mt - 1) & 1], mt - 1);
                }
                power(acc[(MTILES - 1) & 1], MTILES - 1);
            }
            buf ^= 1;
        }
        __syncthreads();  // releases the last power spectrum to the filterbank wave
    } else {
        // ================= filterbank / staging wave =================
        float fbv[FB_REGS];
#pragma unroll
        for (int i = 0; i < FB_REGS; ++i) fbv[i] = fb_packed[i * 64 + lane];
        // Per tile: filters 0-63 of the previous tile, then filters 64-127; the next tile's samples travel through
        // registers in two halves, each fetched before one filterbank half and written to the other image after it.
        constexpr int N1 = ((SAMPLES + 63) / 64 + 1) / 2;
        float sv[N1];
        MelOut o;
        const float* pw_row = nullptr;
        TileInfo cur, next;
        bool live = false;
        auto begin_tile = [&](int tile, const float* pw_tile) {
            cur = tile_info(audio, d_off, d_len, tile_utt, tile_f0, tile);
            const int frames_total = (int)(cur.n / MEL_HOP) + 1;
            const int frames_out = min(frames_total - 1, MEL_MAX_FRAMES);
            const int frame = cur.f0 + lane;
            const bool in_tile = lane < MEL_TILE_FRAMES;   // lanes beyond the tile do nothing
            o.out = out + (int64_t)cur.u * MEL_NMELS * t_stride + frame;
            o.t_stride = t_stride;
            o.store = in_tile && frame < frames_out;
            o.vmax = -INFINITY;
            // running maximum over every frame the reference computes, the dropped last one included (:269-270)
            live = in_tile && frame < frames_total;
            pw_row = pw_tile + min(lane, MEL_TILE_FRAMES - 1) * POW_STRIDE;
        };
        auto end_tile = [&]() {
            const float vmax = wave_max(live ? o.vmax : -INFINITY);
            if (lane == 0 && vmax > -INFINITY) atomicMax(d_max + cur.u, enc_max(vmax));
        };
        int buf = 0, prev = -1;
        for (; t < n_tiles; t += gridDim.x) {
            __syncthreads();
            const int tn = t + gridDim.x;
            float* dst = sm + (buf ^ 1) * SREG;
            bool interior = false;
            if (tn < n_tiles) {
                next = tile_info(audio, d_off, d_len, tile_utt, tile_f0, tn);
                interior = tile_is_interior(next);
            }
            if (interior) fetch_tile<N1, 64, 0>(next, sv, lane);
            if (prev >= 0) {
                begin_tile(prev, pwr + (buf ^ 1) * POW_N);
                mel_rows<0, MEL_NMELS / 2>(pw_row, fbv, o);
            }
            if (interior) {
                write_tile<N1, 64, 0>(sv, dst, lane);
                fetch_tile<N1, 64, N1>(next, sv, lane);
            }
            if (prev >= 0) {
                mel_rows<MEL_NMELS / 2, MEL_NMELS>(pw_row, fbv, o);
                end_tile();
            }
            if (interior) write_tile<N1, 64, N1>(sv, dst, lane);
            else if (tn < n_tiles) stage_edge_tile<16>(next, dst, lane, 64);
            prev = t;
            buf ^= 1;
        }
        __syncthreads();
        if (prev >= 0) {
            begin_tile(prev, pwr + (buf ^ 1) * POW_N);
            mel_rows<0, MEL_NMELS>(pw_row, fbv, o);
            end_tile();
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
    std::vector<float> hann(HANN_N, 0.f);
    // AudioPreprocessing.swift:161-164, evaluated in fp32 like the reference
    for (int i = 0; i <= 200; ++i) hann[i] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)i / (float)MEL_NFFT));

    // Folded DFT tables as MFMA B-fragments: lane (g = lane>>4, j = lane&15) of step s, product pr,
    // bin tile nt holds T_pr[n][k] with k = 16 nt + j and n = 2 (4 s + g) (+1 for the odd products).
    // Angles are reduced (k n mod 400) and evaluated in double: the mathematically exact twiddles.
    std::vector<float> tw((size_t)KSTEPS * 4 * NTILES * 64, 0.f);
    for (int s = 0; s < KSTEPS; ++s)
        for (int pr = 0; pr < 4; ++pr)
            for (int nt = 0; nt < NTILES; ++nt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, j = lane & 15, k = 16 * nt + j;
                    const int n = 2 * (4 * s + g) + (pr & 1);
                    float v = 0.f;
                    const bool cosine = pr < 2;
                    const bool n_ok = cosine ? n <= 200 : (n >= 1 && n <= 199);
                    if (k <= 100 && n_ok) {
                        const double ang = 2.0 * PI * (double)((k * n) % MEL_NFFT) / (double)MEL_NFFT;
                        double val = cosine ? cos(ang) : sin(ang);
                        if (cosine && n == 200) val *= 0.5;  // the fold counts the centre tap twice
                        v = (float)(val * (double)hann[n]);  // the window rides on the twiddle
                    }
                    tw[(((size_t)s * 4 + pr) * NTILES + nt) * 64 + lane] = v;
                }
    t.twid = upload(tw);

    t.host_fb = build_filterbank();
    // The kernel's compile-time filter supports must cover every weight that matters; anything outside them has to be
    // rounding dust of the triangle edges (none with the libm this was built against).
    std::vector<float> packed((size_t)FB_REGS * 64, 0.f);
    for (int m = 0; m < MEL_NMELS; ++m)
        for (int b = 0; b < MEL_NBINS; ++b) {
            const float w = t.host_fb[(size_t)m * MEL_NBINS + b];
            if (b >= FB_LO[m] && b < FB_LO[m] + FB_W[m]) packed[fb_offset(m) + (b - FB_LO[m])] = w;
            else Q3_REQUIRE(fabsf(w) < 1e-9f, "mel: the filterbank has a weight outside the kernel's filter supports");
        }
    t.fb_packed = upload(packed);
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

void launch_mel(const MelTables& t, const float* audio, const int64_t* d_off, const int64_t* d_len, const int* d_tile_utt,
                const int* d_tile_f0, int n_tiles, float* out, int t_stride, unsigned* d_max, hipStream_t s, double work) {
    if (n_tiles <= 0) return;
    static PerDeviceOnce once;
    once.run([&] { HIP_CHECK(hipFuncSetAttribute((const void*)mel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MEL_LDS_BYTES)); });
    const int cus = device_cus();
    ProfScope ps("mel_kernel", work, s);
    // one workgroup fits per CU (142 KB of LDS); each loops over its share of the tiles
    hipLaunchKernelGGL(mel_kernel, dim3(std::min(n_tiles, cus)), dim3(MEL_THREADS), MEL_LDS_BYTES, s, audio, d_off, d_len,
                       d_tile_utt, d_tile_f0, n_tiles, t.twid, t.fb_packed, out, t_stride, d_max);
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
// process ends; all calls on one GPU share it, so calls that overlap on one GPU race (calls through one model are
// serialised by the library; two models on one GPU must not run this entry point at the same time).
namespace {

constexpr int NOALLOC_MAX_UTT = MEL_NOALLOC_MAX_UTT;
constexpr int NOALLOC_MAX_TILES = NOALLOC_MAX_UTT * ((MEL_MAX_FRAMES + 1 + MEL_TILE_FRAMES - 1) / MEL_TILE_FRAMES);   // 63 per 30 s clip

struct NoAllocScratch {
    int64_t off[NOALLOC_MAX_UTT];
    int64_t len[NOALLOC_MAX_UTT];
    unsigned dmax[NOALLOC_MAX_UTT];
    int frames[NOALLOC_MAX_UTT];
    int tile_utt[NOALLOC_MAX_TILES];
    int tile_f0[NOALLOC_MAX_TILES];
};
static_assert(sizeof(NoAllocScratch) <= 160 * 1024, "scratch is meant to stay modest");

// 135 KB per GPU.  In host code this name is only a handle for hipGetSymbolAddress: the device address is looked up
// once per device and every use below goes through that pointer.
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

PerDeviceOnce g_scratch_once;
NoAllocScratch* g_scratch_ptr[PerDeviceOnce::MAX_DEVICES];

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
        HIP_CHECK(hipFuncSetAttribute((const void*)mel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MEL_LDS_BYTES));
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
                                               std::to_string(NOALLOC_MAX_TILES) + " tiles of 48 frames per call)");
    Q3_REQUIRE(t_stride >= tmax, "mel_noalloc: t_stride is smaller than the longest utterance's frame count");
    hipLaunchKernelGGL(mel_noalloc_setup_kernel, dim3(1), dim3(256), 0, s, offs, n_utt, sc);
    HIP_CHECK(hipGetLastError());
    launch_mel(t, audio_dev, sc->off, sc->len, sc->tile_utt, sc->tile_f0, (int)tiles, mel_out_dev, t_stride, sc->dmax, s, bytes);
    launch_mel_normalize(mel_out_dev, n_utt, t_stride, sc->frames, sc->dmax, s);
}
