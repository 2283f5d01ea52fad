/* Oracle (test infrastructure, never linked into the product): the reference's log-mel frontend restated in plain C,
 * loop for loop, so that its own cost -- a direct DFT of every frame on one thread -- can be timed on the GPU box's
 * host cores beside the GPU path, and so that the numpy restatement's "literal" mode has a second witness.
 * Follows Sources/Qwen3ASR/AudioPreprocessing.swift:131-324 of the reference (line numbers in the comments).
 * vDSP_dotpr is restated as a scalar fp32 dot product (the compiler may vectorise it, as Accelerate does). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N_FFT 400
#define HOP 160
#define N_MELS 128
#define N_BINS (N_FFT / 2 + 1)
#define MAX_FRAMES 3000

static float dotpr(const float* a, const float* b, int n) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* audio: n samples at 16 kHz.  fb: [128][201] filterbank (AudioPreprocessing.swift:23-126, built by the caller).
 * out: [128][t_stride] receives the [mel_bins, time_frames] result; *t_out the frame count.  Returns 0, or 1 when out
 * of memory / t_stride too small. */
int q3oracle_mel_direct(const float* audio, int64_t n, const float* fb, float* out, int t_stride, int* t_out) {
    const int pad = N_FFT / 2;                                              /* :135 */
    const int64_t np = pad + n + pad;
    float* padded = (float*)calloc((size_t)np, sizeof(float));              /* :136 */
    if (!padded) return 1;
    for (int i = 0; i < pad; ++i) {                                          /* :140-143 */
        int64_t src = pad - i < n - 1 ? pad - i : n - 1;
        padded[i] = audio[src > 0 ? src : 0];
    }
    memcpy(padded + pad, audio, (size_t)n * sizeof(float));                 /* :146-148 */
    for (int i = 0; i < pad; ++i) {                                          /* :151-154 */
        int64_t src = n - 2 - i;
        padded[pad + n + i] = audio[src > 0 ? src : 0];
    }
    const int64_t n_frames = (np - N_FFT) / HOP + 1;                        /* :157 */
    float window[N_FFT];
    for (int i = 0; i < N_FFT; ++i) window[i] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)i / (float)N_FFT));   /* :161-164 */
    float* tw_re = (float*)malloc(sizeof(float) * N_BINS * N_FFT);          /* :167-176: rebuilt on every call */
    float* tw_im = (float*)malloc(sizeof(float) * N_BINS * N_FFT);
    float* mag = (float*)malloc(sizeof(float) * (size_t)n_frames * N_BINS);
    float* mel = (float*)malloc(sizeof(float) * (size_t)n_frames * N_MELS);
    if (!tw_re || !tw_im || !mag || !mel) return 1;
    for (int k = 0; k < N_BINS; ++k)
        for (int t = 0; t < N_FFT; ++t) {
            const float angle = -2.0f * (float)M_PI * (float)k * (float)t / (float)N_FFT;   /* :172, fp32 throughout */
            tw_re[k * N_FFT + t] = cosf(angle);
            tw_im[k * N_FFT + t] = sinf(angle);
        }
    float frame[N_FFT];
    for (int64_t f = 0; f < n_frames; ++f) {                                 /* :184-209 */
        const int64_t start = f * HOP;
        for (int i = 0; i < N_FFT; ++i) frame[i] = start + i < np ? padded[start + i] * window[i] : 0.f;
        for (int k = 0; k < N_BINS; ++k) {
            /* :203-204: Array(twiddle[...]) makes a heap copy of the row for each of the two dot products */
            float* row_re = (float*)malloc(sizeof(float) * N_FFT);
            float* row_im = (float*)malloc(sizeof(float) * N_FFT);
            if (!row_re || !row_im) return 1;
            memcpy(row_re, tw_re + k * N_FFT, sizeof(float) * N_FFT);
            memcpy(row_im, tw_im + k * N_FFT, sizeof(float) * N_FFT);
            const float sr = dotpr(frame, row_re, N_FFT), si = dotpr(frame, row_im, N_FFT);
            free(row_re);
            free(row_im);
            mag[f * N_BINS + k] = sr * sr + si * si;                        /* :207 */
        }
    }
    for (int64_t f = 0; f < n_frames; ++f)                                   /* :232-240 */
        for (int m = 0; m < N_MELS; ++m) {
            float s = 0.f;
            for (int b = 0; b < N_BINS; ++b) s += mag[f * N_BINS + b] * fb[m * N_BINS + b];
            mel[f * N_MELS + m] = s;
        }
    float mx = -INFINITY;
    for (int64_t i = 0; i < n_frames * N_MELS; ++i) {                        /* :252-255, :269-270 */
        mel[i] = log10f(fmaxf(mel[i], 1e-10f));
        mx = fmaxf(mx, mel[i]);
    }
    const float floor_v = mx - 8.0f;                                         /* :273 */
    for (int64_t i = 0; i < n_frames * N_MELS; ++i) mel[i] = (fmaxf(mel[i], floor_v) + 4.0f) / 4.0f;   /* :274-288 */
    int64_t t = n_frames - 1;                                                /* :301 */
    if (t > MAX_FRAMES) t = MAX_FRAMES;                                      /* :307-313 */
    if (t < 0) t = 0;
    *t_out = (int)t;
    int rc = 0;
    if (t > t_stride) rc = 1;
    else
        for (int64_t f = 0; f < t; ++f)                                      /* :322-323: [frames, mels] -> [mels, frames] */
            for (int m = 0; m < N_MELS; ++m) out[(int64_t)m * t_stride + f] = mel[f * N_MELS + m];
    free(padded); free(tw_re); free(tw_im); free(mag); free(mel);
    return rc;
}
