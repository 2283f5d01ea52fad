// Log-mel frontend (resample -> STFT power -> slaney mel -> log10 -> max-8 clamp -> (x+4)/4).
#pragma once
#include "common.h"
#include <vector>

constexpr int MEL_NFFT = 400, MEL_HOP = 160, MEL_NMELS = 128, MEL_NBINS = 201, MEL_MAX_FRAMES = 3000;
constexpr int MEL_TILE_FRAMES = 32;  // frames per kernel tile

struct MelTables {           // device-resident constants, built once per model
    float* twid = nullptr;   // Hann window [400], inter-stage twiddles [25][8] (cos, -sin), untangling twiddles [101] (cos, sin)
    float* fb_packed = nullptr;  // filterbank as fp32 MFMA B-fragments of the banded column tiles: [2 sets][28 steps][64 lanes]
    std::vector<float> host_fb;  // dense [128][201] filterbank (host copy, for inspection / tests)
};

void mel_tables_create(MelTables& t);
void mel_tables_destroy(MelTables& t);

// Frames the reference produces before trimming: n/160 + 1; returned frames: min(n/160, 3000).
static inline int mel_frames_total(int64_t n) { return (int)(n / MEL_HOP) + 1; }
static inline int mel_frames_out(int64_t n) {
    int t = mel_frames_total(n) - 1;
    return t > MEL_MAX_FRAMES ? MEL_MAX_FRAMES : (t < 0 ? 0 : t);
}

// Linear-interpolation resample of one utterance (device pointers).
void launch_resample(const float* in, int64_t n_in, float* out, int64_t n_out, double ratio, hipStream_t s);

// audio: concatenated 16 kHz samples; d_off[u] / d_len[u]: start and length of utterance u.
// d_tile_utt / d_tile_f0: one entry per MEL_TILE_FRAMES-frame tile.  out: [B][128][t_stride] raw log10 mel.
// d_max: [B] order-preserving encodings of the per-utterance maximum (zeroed by the caller).
// d_tile_min: null (raw rows), or [n_tiles][16] floats: the rows are then stored scaled, (y + 4) / 4, and every tile's
// sixteen cells of 16 frames x 16 mel rows ([2][8]) receive the minimum of their returned values before scaling;
// launch_mel_clamp finishes the job.
void launch_mel(const MelTables& t, const float* audio, const int64_t* d_off, const int64_t* d_len, const int* d_tile_utt,
                const int* d_tile_f0, int n_tiles, float* out, int t_stride, unsigned* d_max, hipStream_t s, double work = 0,
                float* d_tile_min = nullptr);
// Expands d_tile0 ([B + 1] running tile counts: utterance u owns tiles [tile0[u], tile0[u + 1])) into the kernel's tile list.
void launch_mel_tiles(const int* d_tile0, int B, int* d_tile_utt, int* d_tile_f0, hipStream_t s);
// The max-8 clamp on rows stored scaled: only cells whose minimum lies under their utterance's floor are read again.
void launch_mel_clamp(float* out, int t_stride, int n_tiles, const int* d_tile_utt, const int* d_tile_f0, const int* d_frames_out,
                      const unsigned* d_max, const float* d_tile_min, hipStream_t s);
// In-place clamp / scale of rows [0, frames_out[u]) with the per-utterance maximum.
void launch_mel_normalize(float* out, int B, int t_stride, const int* d_frames_out, const unsigned* d_max, hipStream_t s);

// Allocation-free log-mel of 16 kHz audio resident in device memory (limits of its fixed device scratch below).
// mel_noalloc_init: once per device, with that device current (looks up the scratch array's device address).
// mel_noalloc_run: enqueues the kernels on stream s; does not wait for them.
constexpr int MEL_NOALLOC_MAX_UTT = 256;
constexpr int64_t MEL_NOALLOC_MAX_SAMPLES = 16000 * 600;   // per utterance (frames past 30 s only feed the maximum)
void mel_noalloc_init();
void mel_noalloc_run(const MelTables& t, const float* audio_dev, const int64_t* offsets_host, int n_utt, float* mel_out_dev,
                     int t_stride, int32_t* n_frames_host, hipStream_t s);
