// Host runtime of libq3asr: weights, workspace, and the batched transcription pipeline.
#pragma once
#include "../../include/q3asr.h"
#include "common.h"
#include "gemm.h"
#include "attn.h"
#include "ops.h"
#include "mel.h"
#include <map>
#include <set>
#include <string>
#include <vector>

struct HostTensor {
    std::vector<uint8_t> data;
    std::vector<int64_t> shape;
    int dtype = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct LinearW {  // device-resident weight of one (possibly fused) linear, rows = output features
    const void* w = nullptr;
    const bf16_t* scales = nullptr;
    const bf16_t* biases = nullptr;
    int fmt = W_BF16;
    int N = 0, K = 0;
};

struct EncLayerW {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    LinearW qkv, out, fc1, fc2;
    float *qkv_b, *out_b, *fc1_b, *fc2_b;
};

struct DecLayerW {
    float *in_norm, *post_norm, *q_norm, *k_norm;
    LinearW qkv, o, gate_up, down;
};

// Bump allocator over a list of device blocks.  Blocks persist across calls, so a steady stream of equally
// shaped batches allocates nothing after the first one.  Requests above the default block size get a block of
// their own (the KV cache is one such request of up to ~100 GB); when a later call asks for a slightly larger one,
// the idle block that has become too small is released first instead of being kept beside its replacement, and an
// out-of-memory answer releases every idle block before the one retry.  Within a call blocks are never erased
// from the list (marks hold indices): a released block stays as an empty entry until the next call's reset.
struct BlockArena {
    size_t DEFAULT_BLOCK = (size_t)1 << 30;   // Q3ASR_ARENA_BLOCK_MB overrides it (tests reach the large-request path with small models)
    struct Block {
        char* p;
        size_t cap, off;
    };
    std::vector<Block> blocks;
    size_t cur = 0;
    struct Mark {
        size_t cur;
        std::vector<size_t> offs;
    };
    void reset() {   // start of a call: every block is idle, marks of earlier calls are void
        size_t n = 0;
        for (auto& b : blocks)
            if (b.p) blocks[n++] = Block{b.p, b.cap, 0};
        blocks.resize(n);
        cur = 0;
    }
    Mark mark() const {
        Mark k{cur, {}};
        for (auto& b : blocks) k.offs.push_back(b.off);
        return k;
    }
    void restore(const Mark& k) {
        cur = k.cur;
        for (size_t i = 0; i < blocks.size(); ++i) blocks[i].off = i < k.offs.size() ? k.offs[i] : 0;
    }
    size_t footprint() const {
        size_t n = 0;
        for (auto& b : blocks) n += b.cap;
        return n;
    }
    // hipFree waits for the device, so memory of an idle block that earlier kernels of this call still read
    // (handed out before a restore) is not pulled from under them.
    void drop(Block& b) {
        if (b.p) (void)hipFree(b.p);
        b = Block{nullptr, 0, 0};
    }
    template <class T>
    T* get(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        while (cur < blocks.size() && blocks[cur].off + bytes > blocks[cur].cap) ++cur;
        if (cur == blocks.size()) {
            const size_t cap = bytes > DEFAULT_BLOCK ? bytes : DEFAULT_BLOCK;
            if (bytes > DEFAULT_BLOCK)   // idle blocks of the large kind that this request has outgrown
                for (auto& b : blocks)
                    if (b.p && b.off == 0 && b.cap > DEFAULT_BLOCK && b.cap < bytes) drop(b);
            void* p = nullptr;
            hipError_t e = hipMalloc(&p, cap);
            if (e == hipErrorOutOfMemory) {
                (void)hipGetLastError();
                for (auto& b : blocks)
                    if (b.p && b.off == 0) drop(b);
                e = hipMalloc(&p, cap);
            }
            if (e != hipSuccess)
                throw Q3Error(std::string("q3asr: workspace allocation of ") + std::to_string(cap >> 20) + " MiB failed: " + hipGetErrorString(e));
            blocks.push_back(Block{(char*)p, cap, 0});   // cur == the new block: the scan above ran off the end
        }
        T* r = (T*)(blocks[cur].p + blocks[cur].off);
        blocks[cur].off += bytes;
        return r;
    }
    void release() {
        for (auto& b : blocks)
            if (b.p) (void)hipFree(b.p);
        blocks.clear();
        cur = 0;
    }
};

struct q3asr_model {
    q3asr_config cfg;
    int device = 0;
    // Calls on one model are serialised: they share its streams and workspace.  Several host threads may call
    // (the reference's model is driven from detached tasks, Qwen3ASR.swift:685-687, RealtimeTranslation.swift:459-477);
    // each call runs alone and to completion, like the reference's synchronous transcribe.
    std::mutex call_mutex;
    hipStream_t stream = nullptr;
    bool finalized = false;
    std::map<std::string, HostTensor> host;  // checkpoint tensors until finalize
    std::vector<void*> owned;                // device allocations freed at destroy

    MelTables mel;
    bf16_t* zero_page = nullptr;
    // encoder
    float *conv1_w = nullptr, *conv1_b = nullptr, *conv2_b = nullptr, *conv3_b = nullptr;
    LinearW conv2, conv3, conv_out, proj1, proj2;
    float *proj1_b = nullptr, *proj2_b = nullptr, *ln_post_w = nullptr, *ln_post_b = nullptr, *pe_tab = nullptr;
    std::vector<EncLayerW> enc;
    int conv_f3 = 16;  // mel bins left after the three stride-2 convs
    // decoder
    LinearW embed;     // tied embedding / LM head
    float* final_norm = nullptr;
    std::vector<DecLayerW> dec;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    int rope_max_pos = 0;
    BlockArena arena;  // per-call workspace
    // decode-step GEMM variants (tile, split-K) per projection, measured once per batch size
    struct DecodePlan {
        int tile[4], ks[4];  // qkv, o, gate_up, down
        int fused[4];        // 1: unsplit, with the consumer's work in the GEMM epilogue instead of partial-sum slabs
    };
    std::map<int, DecodePlan> decode_plans;   // measured plans (tuning runs), per row bucket
    std::set<int> plans_reported;             // buckets whose plan the verbose log has shown
    // Decode of a large group: the attention kernels run on a stream confined to `split_att_cus` compute units and
    // every other kernel of the step on a stream confined to the remaining ones.  The streams belong to the process
    // (one set per device and share, created when first needed and kept: see ensure_split_streams); a model only
    // borrows them.
    hipStream_t stream_att = nullptr, stream_mm = nullptr;
    hipStream_t stream_att2 = nullptr;       // second stream on the attention's CUs (the second half's attention launches)
    int split_att_cus = 0;
    struct SplitStreams { hipStream_t att = nullptr, mm = nullptr, att2 = nullptr; };
    std::vector<hipEvent_t> chain_events;    // ordering events between the two decode streams

    template <class T>
    T* dalloc(size_t count) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, count * sizeof(T) > 0 ? count * sizeof(T) : 256));
        owned.push_back(p);
        return (T*)p;
    }
};

void model_finalize(q3asr_model* m);
void model_mel(q3asr_model* m, const float* audio, const int64_t* offsets, int n_utt, int sample_rate, int audio_space,
               float* mel_out, int t_stride, int32_t* n_frames, int out_space);
void model_encode(q3asr_model* m, const float* mel, int t_stride, const int32_t* n_frames, int n_utt, int mel_space,
                  float* embeds_out, int64_t out_rows_capacity, int32_t* cu_tokens, int out_space);
void model_transcribe(q3asr_model* m, const q3asr_request* req, q3asr_result* res);
int model_output_length(const q3asr_config& cfg, int frames);
int model_decode_att_cus(const q3asr_model* m, int n_utt);
// Destroys the process's CU-confined decode streams (the library's exit handler, after the last model is gone).
void model_release_process_streams();
