// Bandwidth-bound glue kernels of the transcription path (gfx950).
#pragma once
#include "common.h"

// ---- encoder front: conv2d1 (1 -> C channels, 3x3, stride 2, pad 1) + exact GELU, fused with the
//      log-mel max-8 clamp and (x+4)/4 scale.  Output NHWC bf16 [chunk][64][50][C].
struct Conv1Params {
    const float* mel;       // raw log10 mel [B][128][t_stride]
    int t_stride;
    const unsigned* d_max;  // per-utterance maximum (order-preserving encoding)
    const int* chunk_utt;   // per chunk: utterance, first frame, valid frames
    const int* chunk_t0;
    const int* chunk_len;
    int n_chunks;
    const float* w;         // [C][9]
    const float* bias;      // [C]
    int C;
    bf16_t* out;
    int normalized;         // 1: mel is already clamped / scaled (stand-alone encode entry point)
};
void launch_conv1(const Conv1Params& p, hipStream_t s);

// LayerNorm over fp32 rows -> bf16 (two-pass mean / variance, affine).
void launch_layernorm(const float* x, const float* w, const float* b, bf16_t* out, int rows, int d, float eps, hipStream_t s);

// RMSNorm over fp32 rows -> bf16.  rowidx (optional) gathers input rows.  When nslab > 0 the row
// is first updated in place with the split-K partial sums: x[row] += sum_s slab[s][row].
// dst[i] = row rowidx[i] of src (bf16 rows of `cols` values, a multiple of 8; source row stride ld)
void launch_gather_rows_bf16(const bf16_t* src, int64_t ld, const int* rowidx, bf16_t* dst, int rows, int cols, hipStream_t s);
void launch_rmsnorm(float* x, const float* w, bf16_t* out, int rows, int d, float eps, const int* rowidx, const float* slab,
                    int nslab, int64_t slab_stride, hipStream_t s);

// Per-head q/k RMSNorm + RoPE + KV-cache append (QuantizedTextDecoder.swift:148-169, 211-268).
struct RopeParams {
    const bf16_t* qkv;      // [rows][(heads + 2 kv_heads) * 128] projections (ignored when slab != null)
    int64_t ld;
    const float* slab;      // split-K partial sums [nslab][rows][ld] of the same projections, or null
    int nslab;
    int64_t slab_stride;
    bf16_t* q_out;          // [rows][q_ld]: normalised + rotated queries (may alias qkv)
    int64_t q_ld;
    int rows, heads, kv_heads;
    const float* q_norm_w;  // [128]
    const float* k_norm_w;
    float eps;
    const float* cos_tab;   // [max_pos][half]
    const float* sin_tab;
    int half, interleaved, max_pos;
    const int* row_utt;     // row -> utterance slot (null: row itself)
    const int* row_pos;     // row -> position (null: cur_len[utterance])
    const int* cur_len;
    bf16_t* kcache;         // [utt][kv_head][pos][128]
    bf16_t* vcache;
    int64_t utt_stride, head_stride;
};
void launch_rope(const RopeParams& p, hipStream_t s);

// Prompt embedding gather, std(text)/std(audio) rescale and audio splice (Qwen3ASR.swift:324-363).
struct SpliceParams {
    const int* ids;          // [rows] prompt token ids (packed over utterances)
    const int* row_utt;      // [rows]
    const int* utt_row0;     // [B] first packed row of each utterance
    const int* utt_rows;     // [B] S_u
    const int* utt_a0;       // [B] index of the first audio_pad row inside the prompt
    const int* utt_na;       // [B] audio tokens N_u
    const int* utt_arow0;    // [B] first row in the audio embedding buffer
    int rows, B, hidden;
    const bf16_t* table;     // dense bf16 [vocab][hidden], or null when q4
    const uint32_t* qw; const bf16_t* qs; const bf16_t* qb; int qgroup;  // MLX q4 embedding
    const float* audio;      // [sum N][hidden]
    double* stats;           // [B][4] scratch: text sum, text sumsq, audio sum, audio sumsq (zeroed by caller)
    int embed_scale;
    float* hidden_out;       // [rows][hidden]
};
void launch_splice(const SpliceParams& p, hipStream_t s);

// Token embedding for the decode step: hidden[r] = table[next_tok[slot ? slot[r] : r]].
void launch_embed_tokens(const int* tok, const int* slot, int B, int hidden, const bf16_t* table, const uint32_t* qw, const bf16_t* qs,
                         const bf16_t* qb, int qgroup, float* out, hipStream_t s);

// Final reduction of the LM-head argmax partials: lowest index among equal maxima.
void launch_argmax_reduce(const float* part_val, const int* part_idx, int nparts, int rows, int* out, hipStream_t s);
// Plain argmax over fp32 logits rows (debug tap path).
void launch_argmax_rows(const float* logits, int rows, int n, int* out, hipStream_t s);

// Greedy-loop bookkeeping and stop rules (Qwen3ASR.swift:567-627), one thread per row of the step.
// The step's B rows belong to utterances slot[0..B) (null: row r is utterance r); every per-utterance array
// is indexed by utterance.  The reference stops computing for a clip the moment it stops (:571-574, batch 1);
// here the kernel also writes the utterances still running, in row order, to live_out, so that the host can
// drop the finished rows from the next steps by switching to that list.
struct StepParams {
    const int* argmax;   // [B] token chosen by this step, per row
    int* tokens;         // [utt][max_tokens]
    int* ntok;           // [utt]
    int* done;           // [utt]
    int* seen_asr;       // [utt]
    int* next_tok;       // [utt] token fed to the next step
    int* cur_len;        // [utt] keys in the cache; incremented when advance != 0
    int* n_active;       // [1] utterances still running after this step
    const int* forced;   // [utt][max_tokens] teacher-forced inputs or null
    int B, max_tokens, advance;
    int stop_rules;      // 0: none; 1: the reference's rules on the generated tokens; 2: on the teacher-forced inputs
    const int* slot = nullptr;    // [B] row -> utterance
    int* live_out = nullptr;      // [<= B] utterances still running after this step, in row order (optional)
    long long* keys_out = nullptr;  // [1] sum over those utterances of the keys their next step attends (optional)
};
void launch_step(const StepParams& p, hipStream_t s);

// SwiGLU over split-K partial sums of the interleaved gate/up projection (blocks of 8 gate, 8 up
// columns): out[row][j] = silu(sum_s gate) * (sum_s up), bf16 [rows][n2 / 2].
void launch_swiglu_reduce(const float* slab, int nslab, int64_t slab_stride, int rows, int n2, bf16_t* out, hipStream_t s);

// Average milliseconds of one streaming read of `bytes` bytes (16-byte loads, `blocks` workgroups): the HBM read
// rate this GPU reaches, measured beside the kernels that are bound by it.
float read_probe_ms(const void* p, size_t bytes, int blocks, int reps, hipStream_t s);
// TFLOP/s of a register-only stream of fp32 16x16x4 MFMAs (4 independent chains per wave) with the given residency.
float mfma_f32_probe_tflops(float* sink, int waves_per_simd, int iters, hipStream_t s);

// Pseudo-random bf16 values in (-1, 1) for the kernel timing entry points.
void launch_fill_bf16(bf16_t* p, size_t n, uint32_t seed, hipStream_t s);
// One empty workgroup: a stream-ordered point that has passed every wait queued before it.
void launch_noop(hipStream_t s);
