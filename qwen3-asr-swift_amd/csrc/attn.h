// Attention kernels (gfx950): MFMA flash attention over packed variable-length sequences
// (encoder windows, decoder prefill) and the single-query KV-cache decode step.
#pragma once
#include "common.h"

struct AttnSeq {       // one attention problem (an encoder window, or one utterance's prompt)
    int q_row0;        // first row of the sequence in the packed Q / O buffers
    int len;           // number of queries == number of keys
    int64_t kv_off;    // element offset of (key 0, kv-head 0) from the K / V base pointers
};
struct AttnTile {      // one workgroup: 64 consecutive queries of one sequence
    int seq;
    int q0;
};

struct FlashParams {
    const bf16_t* Q; int64_t q_ld;          // Q[row][head*D + d]
    const bf16_t* K; const bf16_t* V;       // K/V[kv_off + key*kv_ld + kvhead*kv_hs + d]
    int64_t kv_ld, kv_hs;
    bf16_t* O; int64_t o_ld;                // O[row][head*D + d]
    const AttnSeq* seqs; const AttnTile* tiles; int n_tiles;
    int heads, group;                       // group = q heads per kv head
    float scale;
    int causal;
};
// work: algorithmic FLOPs of the launch (profiling only)
void launch_flash_attn(const FlashParams& p, int head_dim, hipStream_t s, double work = 0);

// Optional input of the decode-attention kernel: this step's raw q/k/v projections.  When given, the
// kernel applies the per-head RMSNorm and RoPE itself (QuantizedTextDecoder.swift:148-169), appends the
// new K and V rows to the cache at position lens[u], and attends over lens[u] cached keys plus the new one.
struct DecodeRopeIn {
    const bf16_t* qkv = nullptr;   // [B][ld] bf16 projections ((heads + 2 kv_heads) * 128 wide), or
    const float* slab = nullptr;   // split-K partial sums [nslab][B][ld] of the same
    int nslab = 0;
    int64_t slab_stride = 0, ld = 0;
    const float* q_norm_w = nullptr;
    const float* k_norm_w = nullptr;
    float eps = 0.f;
    const float* cos_tab = nullptr;  // [max_pos][half]
    const float* sin_tab = nullptr;
    int half = 0, interleaved = 0, max_pos = 0;
};

// The step's B rows belong to utterances slot[0..B) (null: row r is utterance r).  Q, O and the raw projections
// are indexed by row; the cache, lens and done by utterance.
struct DecodeAttnParams {
    const bf16_t* Q; int64_t q_ld;          // Q[row][head*D + d] (already normalised + rotated)
    const bf16_t* K; const bf16_t* V;       // cache[u*utt_stride + kvhead*head_stride + pos*D + d]
    int64_t utt_stride, head_stride;
    const int* lens; int len_add;           // keys visible to utterance u = lens[u] + len_add
    bf16_t* O; int64_t o_ld;
    int B, kv_heads, group;
    float scale;
    int max_blocks = 0;                     // > 0: at most this many workgroups, each looping over (row, head) items
    DecodeRopeIn rope;                      // set: Q is ignored, K/V are appended in place (len_add must be 0)
    const int* slot = nullptr;              // [B] row -> utterance
    const int* done = nullptr;              // [utt] non-zero: the utterance has stopped (Qwen3ASR.swift:571-574): its cache is
                                            // neither read nor appended to, its output rows are zero
};
// work: algorithmic K/V bytes of the launch (profiling only)
void launch_decode_attn(const DecodeAttnParams& p, int head_dim, hipStream_t s, double work = 0);
