// Log-mel frontend for gfx950.
//
// Restates WhisperFeatureExtractor (reference: Sources/Qwen3ASR/AudioPreprocessing.swift:131-324).
// One workgroup = 64 frames of one utterance.  The 10 480 samples those frames cover are staged
// once in LDS (coalesced, reflect padding resolved on load).  Because the frame is real, the
// 400-point DFT folds to Re X[k] = sum_{n<=200} h[n](x[n] + x[400-n]) cos(2 pi k n / 400) and
// Im X[k] = -sum h[n](x[n] - x[400-n]) sin(...); splitting those sums into even and odd n gives four
// [frames x 101] x [101 x 101] real products (Ce, Co, Se, So over bins 0..100) from which bin k is
// (Ce + Co, Se + So) and bin 200 - k is (Ce - Co, Se - So): a quarter of the direct DFT's multiplies,
// evaluated on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, an fp32 fma chain).  The difference form
// loses accuracy only where bin 200 - k is far weaker than bin k: relative error ~6e-8 * |X[k]| / |X[200-k]|,
// i.e. < 1e-5 of log-mel for a 60 dB spread between the halves of the spectrum.  Power, the sparse
// slaney filterbank, log10 and a per-utterance running maximum (wave reduction + one atomic per
// workgroup) follow without leaving the CU.  Output is the raw log10 mel; the max-8 clamp and
// (x+4)/4 scale need the utterance-wide maximum and are applied by the consumer (conv1) or by
// launch_mel_normalize for callers that want the reference's exact output tensor.
#include "mel.h"
#include "prof.h"
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int KSTEPS = 26;     // 104 >= 101 folded taps of each parity, 4 per MFMA
constexpr int NTILES = 7;      // 112 >= 101 bins
constexpr int SAMPLES = MEL_TILE_FRAMES * MEL_HOP + (MEL_NFFT - MEL_HOP) + 1;  // 10481
constexpr int POW_STRIDE = MEL_NBINS;
constexpr int REGION = MEL_TILE_FRAMES * POW_STRIDE;  // 12864 floats >= skewed sample image (10481 + 133)
constexpr int HANN_N = 208;

__device__ __forceinline__ int skew(int p) {
    const int q = p / MEL_HOP;
    return p + 4 * (q >> 1) + (q & 1);  // frame starts land on distinct banks for the stride-2 fragment reads
}

__device__ __forceinline__ unsigned enc_max(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_max(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ audio, const int64_t* __restrict__ d_off,
                                                  const int64_t* __restrict__ d_len, const int* __restrict__ tile_utt,
                                                  const int* __restrict__ tile_f0, const float* __restrict__ hann_g,
                                                  const float* __restrict__ twid, const int* __restrict__ fb_lo,
                                                  const float* __restrict__ fb_dense, float* __restrict__ out, int t_stride,
                                                  unsigned* __restrict__ d_max) {
    __shared__ __attribute__((aligned(16))) float sm[REGION + HANN_N + 4];
    float* hann = sm + REGION;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = tile_utt[blockIdx.x], f0 = tile_f0[blockIdx.x];
    const int64_t n = d_len[u];
    const float* x = audio + d_off[u];
    const int frames_total = (int)(n / MEL_HOP) + 1;
    const int frames_out = min(frames_total - 1, MEL_MAX_FRAMES);

    // ---- stage samples (reflect pad of 200: AudioPreprocessing.swift:134-154) ----
    const int64_t P0 = (int64_t)f0 * MEL_HOP;
    for (int p = tid; p < SAMPLES; p += 256) {
        const int64_t P = P0 + p;
        float v = 0.f;
        if (n > 0 && P < n + MEL_NFFT) {
            int64_t idx;
            if (P < 200) idx = min((int64_t)200 - P, n - 1);
            else if (P < 200 + n) idx = P - 200;
            else idx = n - 2 - (P - 200 - n);
            v = x[max(idx, (int64_t)0)];
        }
        sm[skew(p)] = v;
    }
    if (tid < HANN_N) hann[tid] = hann_g[tid];
    __syncthreads();

    // ---- folded DFT on the fp32 MFMA: wave w owns frames 16w .. 16w+15 ----
    f32x4 acc[4][NTILES];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NTILES; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lg = lane >> 4;
    const int s0 = (16 * wave + li) * MEL_HOP;
    const float* tw = twid + lane;
    for (int s = 0; s < KSTEPS; ++s) {
        const int e = 2 * (4 * s + lg);
        const float xa = sm[skew(s0 + e)], xb = sm[skew(s0 + MEL_NFFT - e)];
        const float xc = sm[skew(s0 + e + 1)], xd = sm[skew(s0 + MEL_NFFT - 1 - e)];
        const float he = hann[e], ho = hann[e + 1];
        float av[4];
        av[0] = he * (xa + xb);  // even samples, cosine part
        av[1] = ho * (xc + xd);  // odd samples, cosine part
        av[2] = he * (xa - xb);  // even samples, sine part
        av[3] = ho * (xc - xd);  // odd samples, sine part
#pragma unroll
        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt) {
                const float b = tw[((s * 4 + pr) * NTILES + nt) * 64];
                acc[pr][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[pr], b, acc[pr][nt], 0, 0, 0);
            }
    }
    __syncthreads();  // every wave is done with the sample image; reuse it for the power spectrum

    // ---- power spectrum |X[k]|^2 (AudioPreprocessing.swift:207): bins k and 200-k from the same four sums ----
#pragma unroll
    for (int nt = 0; nt < NTILES; ++nt) {
        const int k = 16 * nt + li;
        if (k <= 100) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = 16 * wave + 4 * lg + r;
                const float ce = acc[0][nt][r], co = acc[1][nt][r], se = acc[2][nt][r], so = acc[3][nt][r];
                const float re = ce + co, im = se + so;
                sm[fr * POW_STRIDE + k] = re * re + im * im;
                if (k < 100) {
                    const float re2 = ce - co, im2 = se - so;
                    sm[fr * POW_STRIDE + 200 - k] = re2 * re2 + im2 * im2;
                }
            }
        }
    }
    __syncthreads();

    // ---- sparse slaney filterbank + log10 (AudioPreprocessing.swift:232-255), running max (:269-270) ----
    const int f = tid & 63, mg = tid >> 6;
    const int frame = f0 + f;
    const float* pw = sm + f * POW_STRIDE;
    float vmax = -INFINITY;
    // Every filter is stored MEL_FB_WIDTH wide (zero weights after its last bin), so the table reads are
    // wave-uniform loads of fixed size that do not depend on one another; the zero terms leave the sum unchanged.
#pragma unroll 4
    for (int i = 0; i < MEL_NMELS / 4; ++i) {
        const int m = mg * (MEL_NMELS / 4) + i;
        const int lo = fb_lo[m];
        const float* w = fb_dense + m * MEL_FB_WIDTH;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < MEL_FB_WIDTH; ++c) sum = fmaf(pw[lo + c], w[c], sum);
        const float v = log10f(fmaxf(sum, 1e-10f));
        if (frame < frames_total) vmax = fmaxf(vmax, v);
        if (frame < frames_out) out[((int64_t)u * MEL_NMELS + m) * t_stride + frame] = v;
    }
    vmax = wave_max(vmax);
    __shared__ float wmax[4];
    if (lane == 0) wmax[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
        const float b = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (b > -INFINITY) atomicMax(d_max + u, enc_max(b));
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
    t.hann = upload(hann);

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
                        v = (float)val;
                    }
                    tw[(((size_t)s * 4 + pr) * NTILES + nt) * 64 + lane] = v;
                }
    t.twid = upload(tw);

    t.host_fb = build_filterbank();
    std::vector<int> lo(MEL_NMELS);
    std::vector<float> dense((size_t)MEL_NMELS * MEL_FB_WIDTH, 0.f);
    for (int m = 0; m < MEL_NMELS; ++m) {
        int first = -1, last = -1;
        for (int b = 0; b < MEL_NBINS; ++b)
            if (t.host_fb[(size_t)m * MEL_NBINS + b] != 0.f) {
                if (first < 0) first = b;
                last = b;
            }
        if (first < 0) first = 0, last = -1;
        Q3_REQUIRE(last - first + 1 <= MEL_FB_WIDTH, "mel: a filter spans more FFT bins than the kernel's table width");
        lo[m] = first;
        for (int b = first; b <= last; ++b) dense[(size_t)m * MEL_FB_WIDTH + (b - first)] = t.host_fb[(size_t)m * MEL_NBINS + b];
    }
    t.fb_lo = upload(lo);
    t.fb_dense = upload(dense);
}

void mel_tables_destroy(MelTables& t) {
    (void)hipFree(t.hann); (void)hipFree(t.twid); (void)hipFree(t.fb_lo); (void)hipFree(t.fb_dense);
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
    ProfScope ps("mel_kernel", work, s);
    hipLaunchKernelGGL(mel_kernel, dim3(n_tiles), dim3(256), 0, s, audio, d_off, d_len, d_tile_utt, d_tile_f0, t.hann,
                       t.twid, t.fb_lo, t.fb_dense, out, t_stride, d_max);
    HIP_CHECK(hipGetLastError());
}

void launch_mel_normalize(float* out, int B, int t_stride, const int* d_frames_out, const unsigned* d_max, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(mel_normalize_kernel, dim3(MEL_NMELS, B), dim3(256), 0, s, out, t_stride, d_frames_out, d_max);
    HIP_CHECK(hipGetLastError());
}
