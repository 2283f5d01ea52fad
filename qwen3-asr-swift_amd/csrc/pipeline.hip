// Batched transcription pipeline: mel -> encoder -> prompt splice -> prefill -> greedy decode.
// Host control flow of Qwen3ASRModel.transcribe / _generateText (reference: Qwen3ASR.swift:60-145,
// 170-632) re-designed for many utterances at once on one GPU: utterances are sorted by length,
// processed in groups of cfg.max_batch, packed without padding (rows = sum of lengths) through
// the GEMMs, and decoded together with per-utterance KV-cache lengths; the stop rules run on the
// device and the host only polls a counter every few steps.
#include "model.h"
#include "gemv.h"
#include "prof.h"
#include "decode_plan_table.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <numeric>

namespace {

constexpr int TOK_IM_START = 151644, TOK_IM_END = 151645, TOK_AUDIO_START = 151669, TOK_AUDIO_END = 151670;
constexpr int TOK_AUDIO_PAD = 151676, TOK_ASR_TEXT = 151704, TOK_NEWLINE = 198, TOK_SYSTEM = 8948, TOK_USER = 872;
constexpr int TOK_ASSISTANT = 77091;
constexpr int ENC_CHUNK_BUDGET = 5120;  // most chunks per encoder pass (170 clips of 30 s; conv1's output is 3 MB per chunk;
                                        // measured: 960 -> 673 ms, 3840 -> 590 ms, 7680 -> 581 ms per 1024 clips)
constexpr int POLL_STEPS = 8;
constexpr int SPLIT_MIN_ROWS = 640;     // groups from this size on decode as two halves on disjoint sets of compute units
constexpr int SPLIT_MIN_ROWS_HEAVY_Q4 = 1024;   // ... for 4-bit weights at the 1.7B widths (not measured below 1024)
constexpr int SPLIT_MERGE_ROWS = 512;   // ... until this few utterances are still running

inline int conv_len(int n) { return (n - 1) / 2 + 1; }
inline int feat_len(int clen) { return conv_len(conv_len(conv_len(clen))); }

struct Ctx {
    q3asr_model* m;
    hipStream_t s;
    std::deque<std::vector<uint8_t>> keep;  // host staging kept alive until the call's last sync
    template <class T>
    T* get(size_t n) { return m->arena.get<T>(n); }
    template <class T>
    T* up(const std::vector<T>& v) {
        T* d = get<T>(v.size());
        if (!v.empty()) {
            keep.emplace_back((const uint8_t*)v.data(), (const uint8_t*)v.data() + v.size() * sizeof(T));
            HIP_CHECK(hipMemcpyAsync(d, keep.back().data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
        }
        return d;
    }
};

GemmParams gp(const Ctx& c, const bf16_t* A, int64_t lda, const LinearW& W, int M) {
    GemmParams p;
    p.A = A; p.lda = lda; p.W = W.w; p.qscales = W.scales; p.qbiases = W.biases; p.qgroup = c.m->cfg.group_size;
    p.M = M; p.N = W.N; p.K = W.K; p.zero = c.m->zero_page;
    return p;
}

// ------------------------------------------------------------------------------------------ mel
struct MelRun {
    float* raw = nullptr;      // [b][128][t_stride] raw log10 mel
    unsigned* dmax = nullptr;  // [b]
    int* d_frames = nullptr;   // [b]
    int t_stride = 0;
    std::vector<int> frames;   // frames returned per utterance
    std::vector<int64_t> n16;  // samples at 16 kHz
};

// src_off / src_len: position of every utterance in `audio` (input sample rate).
// dst: optional device buffer [b][128][t_stride] that receives the raw log-mel instead of workspace memory.
// normalize: the rows come back clamped and scaled (the stand-alone stage); otherwise they stay raw for conv1.
MelRun run_mel(Ctx& c, const float* audio, int space, const std::vector<int64_t>& src_off, const std::vector<int64_t>& src_len,
               int sample_rate, int t_stride, float* dst = nullptr, bool normalize = false) {
    const int b = (int)src_off.size();
    MelRun r;
    r.n16.resize(b);
    r.frames.resize(b);
    const double ratio = 16000.0 / (double)sample_rate;
    int64_t tot_in = 0, tot16 = 0;
    for (int u = 0; u < b; ++u) {
        // resample output length: AudioPreprocessing.swift:348-349
        r.n16[u] = sample_rate == 16000 ? src_len[u] : (int64_t)((double)src_len[u] * ratio);
        r.frames[u] = mel_frames_out(r.n16[u]);
        tot_in += src_len[u];
        tot16 += r.n16[u];
    }
    // stage the input on the device
    const float* d_in = audio;
    std::vector<int64_t> in_off = src_off;
    if (space == Q3ASR_HOST) {
        float* buf = c.get<float>(tot_in);
        int64_t o = 0;
        for (int u = 0; u < b; ++u) {
            if (src_len[u]) HIP_CHECK(hipMemcpyAsync(buf + o, audio + src_off[u], src_len[u] * 4, hipMemcpyHostToDevice, c.s));
            in_off[u] = o;
            o += src_len[u];
        }
        d_in = buf;
    }
    const float* d16 = d_in;
    std::vector<int64_t> off16 = in_off;
    if (sample_rate != 16000) {
        float* buf = c.get<float>(tot16);
        int64_t o = 0;
        for (int u = 0; u < b; ++u) {
            launch_resample(d_in + in_off[u], src_len[u], buf + o, r.n16[u], ratio, c.s);
            off16[u] = o;
            o += r.n16[u];
        }
        d16 = buf;
    }
    std::vector<int> tile0(b + 1, 0);   // the device expands the running counts into the tile list
    int tmax = 1;
    for (int u = 0; u < b; ++u) {
        tile0[u + 1] = tile0[u] + (mel_frames_total(r.n16[u]) + MEL_TILE_FRAMES - 1) / MEL_TILE_FRAMES;
        tmax = std::max(tmax, r.frames[u]);
    }
    const int n_tiles = tile0[b];
    int* d_tile_utt = c.get<int>(std::max(n_tiles, 1));
    int* d_tile_f0 = c.get<int>(std::max(n_tiles, 1));
    launch_mel_tiles(c.up(tile0), b, d_tile_utt, d_tile_f0, c.s);
    r.t_stride = t_stride > 0 ? t_stride : ((tmax + 3) & ~3);
    Q3_REQUIRE(r.t_stride >= tmax, "mel: t_stride is smaller than the longest utterance's frame count");
    double mel_bytes = 0;  // algorithmic traffic: samples in + returned frames out
    for (int u = 0; u < b; ++u) mel_bytes += 4.0 * (double)r.n16[u] + 4.0 * MEL_NMELS * (double)r.frames[u];
    r.raw = dst ? dst : c.get<float>((size_t)b * MEL_NMELS * r.t_stride);
    r.dmax = c.get<unsigned>(b);
    HIP_CHECK(hipMemsetAsync(r.dmax, 0, (size_t)b * 4, c.s));
    r.d_frames = c.up(r.frames);
    const char* env_two = getenv("Q3ASR_MEL_TWO_PASS");   // 1: raw rows, then a full clamp-and-scale pass (what the scaled store is measured against)
    float* d_min = nullptr;   // per tile: minima of its sixteen cells
    if (normalize && !(env_two && atoi(env_two) != 0)) d_min = c.get<float>((size_t)std::max(n_tiles, 1) * 16);
    launch_mel(c.m->mel, d16, c.up(off16), c.up(r.n16), d_tile_utt, d_tile_f0, n_tiles, r.raw, r.t_stride, r.dmax, c.s, mel_bytes, d_min);
    if (normalize && d_min) launch_mel_clamp(r.raw, r.t_stride, n_tiles, d_tile_utt, d_tile_f0, r.d_frames, r.dmax, d_min, c.s);
    else if (normalize) launch_mel_normalize(r.raw, b, r.t_stride, r.d_frames, r.dmax, c.s);
    return r;
}

// ------------------------------------------------------------------------------------------ encoder
// Encodes utterances whose (raw or normalised) mel rows live in `mel`; writes packed fp32
// embeddings [sum N][out_dim] to `out` in the given order and returns the token counts.
std::vector<int> run_encoder(Ctx& c, const float* mel, int t_stride, const unsigned* dmax, int normalized,
                             const std::vector<int>& frames, float* out) {
    q3asr_model* m = c.m;
    const q3asr_config& cfg = m->cfg;
    const int b = (int)frames.size(), d = cfg.enc_d_model, C = cfg.conv_channels, F3 = m->conv_f3;
    const int chunk = cfg.n_window * 2;
    Q3_REQUIRE(chunk == 100 && cfg.n_mels == 128, "encoder: kernels are built for 128 mel bins and 100-frame chunks");
    std::vector<int> ntok(b);
    for (int u = 0; u < b; ++u) ntok[u] = model_output_length(cfg, frames[u]);
    int64_t out_row = 0;
    int u0 = 0;
    while (u0 < b) {
        // ---- group of utterances within the chunk budget ----
        int u1 = u0, nch = 0;
        while (u1 < b) {
            const int k = (frames[u1] + chunk - 1) / chunk;
            if (u1 > u0 && nch + k > ENC_CHUNK_BUDGET) break;
            nch += k;
            ++u1;
        }
        // The layers' narrow linears (N = d_model) run one 256 x 256 tile per compute unit per round, and with only
        // d_model / 256 column tiles a pass of 170 clips leaves most of its last round empty (1036 tiles: a fifth round
        // for 12).  When more utterances follow, the pass gives up to an eighth of its utterances to the next one if
        // the rounds it then runs are fuller (168 clips of 30 s: 65 520 rows = 256 row tiles, four full rounds).
        if (u1 < b) {
            const int cus = device_cus(), tiles_n = cdiv(d, 256);
            std::vector<int64_t> rows_upto(u1 - u0 + 1, 0);
            for (int u = u0; u < u1; ++u) rows_upto[u - u0 + 1] = rows_upto[u - u0] + ntok[u];
            auto fill = [&](int64_t rows) {
                const int64_t rounds = cdiv((int64_t)cdiv(rows, 256) * tiles_n, cus);
                return (double)rows * d / ((double)rounds * cus * 65536.0);
            };
            int best = u1;
            double best_fill = fill(rows_upto[u1 - u0]);
            for (int e = u1 - 1; e > u0 && e >= u1 - (u1 - u0) / 8; --e) {
                const double f = fill(rows_upto[e - u0]);
                if (f > best_fill + 0.03) { best_fill = f; best = e; }
            }
            u1 = best;
        }
        const auto mark = m->arena.mark();
        std::vector<int> ch_utt, ch_t0, ch_len, win1, win2, rowmap3, rowmap_tok;
        std::vector<AttnSeq> seqs;
        std::vector<AttnTile> tiles;
        int ntok_g = 0;
        for (int u = u0; u < u1; ++u) {
            const int T = frames[u], k = (T + chunk - 1) / chunk;
            // chunk lengths and the common padded length: AudioEncoder.swift:335-374
            const int last = (T % chunk == 0) ? chunk : T % chunk;
            const int wp = k > 1 ? chunk : last;
            const int w1 = conv_len(wp), w2 = conv_len(w1);
            const int tok0 = ntok_g;
            for (int i = 0; i < k; ++i) {
                const int clen = (i == k - 1) ? last : chunk;
                ch_utt.push_back(u); ch_t0.push_back(i * chunk); ch_len.push_back(clen);
                win1.push_back(w1); win2.push_back(w2);
                const int fl = feat_len(clen);  // valid tokens of the chunk: :416-423
                for (int t = 0; t < 13; ++t) rowmap_tok.push_back(t < fl ? ntok_g + t : -1);
                ntok_g += fl;
            }
            Q3_REQUIRE(ntok_g - tok0 == ntok[u], "encoder: token count mismatch");
            // attention windows: :441-466
            if (k > 0) {
                int maxfl = 0;
                for (int i = 0; i < k; ++i) maxfl = std::max(maxfl, feat_len(i == k - 1 ? last : chunk));
                const int window = maxfl * (cfg.n_window_infer / chunk);
                for (int st = 0; st < ntok[u]; st += window) {
                    const int len = std::min(window, ntok[u] - st);
                    seqs.push_back({tok0 + st, len, (int64_t)(tok0 + st) * 3 * d});
                    for (int q0 = 0; q0 < len; q0 += 64) tiles.push_back({(int)seqs.size() - 1, q0});
                }
            }
        }
        const int nc = (int)ch_utt.size();
        double enc_attn_flops = 0;
        for (auto& sq : seqs) enc_attn_flops += 4.0 * sq.len * sq.len * d;
        if (nc > 0 && ntok_g > 0) {
            for (int ch = 0; ch < nc; ++ch)
                for (int oy = 0; oy < F3; ++oy)
                    for (int ox = 0; ox < 13; ++ox) rowmap3.push_back((ch * 13 + ox) * F3 + oy);
            bf16_t* c1 = c.get<bf16_t>((size_t)nc * 64 * 50 * C);
            bf16_t* c2 = c.get<bf16_t>((size_t)nc * 32 * 25 * C);
            bf16_t* c3 = c.get<bf16_t>((size_t)nc * 13 * F3 * C);
            float* h = c.get<float>((size_t)ntok_g * d);
            bf16_t* xn = c.get<bf16_t>((size_t)ntok_g * d);
            bf16_t* qkv = c.get<bf16_t>((size_t)ntok_g * 3 * d);
            bf16_t* att = c.get<bf16_t>((size_t)ntok_g * d);
            bf16_t* ffn = c.get<bf16_t>((size_t)ntok_g * cfg.enc_ffn);
            const int* d_win1 = c.up(win1);
            const int* d_win2 = c.up(win2);
            const int* d_rowmap3 = c.up(rowmap3);
            const int* d_rowmap_tok = c.up(rowmap_tok);
            const AttnSeq* d_seqs = c.up(seqs);
            const AttnTile* d_tiles = c.up(tiles);

            Conv1Params p1{mel, t_stride, dmax, c.up(ch_utt), c.up(ch_t0), c.up(ch_len), nc, m->conv1_w, m->conv1_b, C, c1, normalized};
            launch_conv1(p1, c.s);
            {   // conv2d2 + GELU: implicit GEMM, rows (chunk, 32, 25)
                GemmParams p = gp(c, c1, 0, m->conv2, nc * 32 * 25);
                p.Hin = 64; p.Win = 50; p.Cin = C; p.Hout = 32; p.Wout = 25; p.win_valid = d_win1;
                p.bias = m->conv2_b; p.act = 1; p.out = c2; p.ldo = C;
                launch_gemm(p, A_CONV3X3S2, OUT_BF16, W_BF16, -1, c.s);
            }
            {   // conv2d3 + GELU, rows (chunk, 16, 13) stored as [chunk][t][f][c]
                GemmParams p = gp(c, c2, 0, m->conv3, nc * F3 * 13);
                p.Hin = 32; p.Win = 25; p.Cin = C; p.Hout = F3; p.Wout = 13; p.win_valid = d_win2;
                p.bias = m->conv3_b; p.act = 1; p.out = c3; p.ldo = C; p.rowmap = d_rowmap3;
                launch_gemm(p, A_CONV3X3S2, OUT_BF16, W_BF16, -1, c.s);
            }
            {   // conv_out (no bias) + per-chunk sinusoid + valid-token packing: :401-436
                GemmParams p = gp(c, c3, (int64_t)F3 * C, m->conv_out, nc * 13);
                p.rowtab = m->pe_tab; p.rowtab_period = 13; p.rowmap = d_rowmap_tok; p.out = h; p.ldo = d;
                launch_gemm(p, A_PLAIN, OUT_F32, W_BF16, -1, c.s);
            }
            for (int l = 0; l < cfg.enc_layers; ++l) {
                const EncLayerW& L = m->enc[l];
                launch_layernorm(h, L.ln1_w, L.ln1_b, xn, ntok_g, d, cfg.ln_eps, c.s);
                {
                    GemmParams p = gp(c, xn, d, L.qkv, ntok_g);
                    p.bias = L.qkv_b; p.out = qkv; p.ldo = 3 * d;
                    launch_gemm(p, A_PLAIN, OUT_BF16, W_BF16, -1, c.s);
                }
                FlashParams fp{qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, d / cfg.enc_heads, att, d, d_seqs, d_tiles, (int)tiles.size(),
                               cfg.enc_heads, 1, 1.0f / sqrtf((float)(d / cfg.enc_heads)), 0};
                launch_flash_attn(fp, d / cfg.enc_heads, c.s, enc_attn_flops);
                {
                    GemmParams p = gp(c, att, d, L.out, ntok_g);
                    p.bias = L.out_b; p.out = h; p.ldo = d;
                    launch_gemm(p, A_PLAIN, OUT_F32_ACC, W_BF16, -1, c.s);
                }
                launch_layernorm(h, L.ln2_w, L.ln2_b, xn, ntok_g, d, cfg.ln_eps, c.s);
                {
                    GemmParams p = gp(c, xn, d, L.fc1, ntok_g);
                    p.bias = L.fc1_b; p.act = 1; p.out = ffn; p.ldo = cfg.enc_ffn;
                    launch_gemm(p, A_PLAIN, OUT_BF16, W_BF16, -1, c.s);
                }
                {
                    GemmParams p = gp(c, ffn, cfg.enc_ffn, L.fc2, ntok_g);
                    p.bias = L.fc2_b; p.out = h; p.ldo = d;
                    launch_gemm(p, A_PLAIN, OUT_F32_ACC, W_BF16, -1, c.s);
                }
            }
            launch_layernorm(h, m->ln_post_w, m->ln_post_b, xn, ntok_g, d, cfg.ln_eps, c.s);
            {
                GemmParams p = gp(c, xn, d, m->proj1, ntok_g);
                p.bias = m->proj1_b; p.act = 1; p.out = att; p.ldo = d;
                launch_gemm(p, A_PLAIN, OUT_BF16, W_BF16, -1, c.s);
            }
            {
                GemmParams p = gp(c, att, d, m->proj2, ntok_g);
                p.bias = m->proj2_b; p.out = out + out_row * cfg.enc_out_dim; p.ldo = cfg.enc_out_dim;
                launch_gemm(p, A_PLAIN, OUT_F32, W_BF16, -1, c.s);
            }
        }
        out_row += ntok_g;
        m->arena.restore(mark);
        u0 = u1;
    }
    return ntok;
}

}  // namespace

int model_output_length(const q3asr_config& cfg, int frames) {
    // getOutputLength: AudioEncoder.swift:254-270
    const int chunk = cfg.n_window * 2, rem = frames % chunk;
    return (frames / chunk) * 13 + (rem > 0 ? std::max(feat_len(rem), 1) : 0);
}

void model_mel(q3asr_model* m, const float* audio, const int64_t* offsets, int n_utt, int sample_rate, int audio_space,
               float* mel_out, int t_stride, int32_t* n_frames, int out_space) {
    Q3_REQUIRE(n_utt > 0 && audio && offsets && mel_out && n_frames, "mel: null argument or empty batch");
    Q3_REQUIRE(sample_rate > 0, "mel: sample rate must be positive");
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    Ctx c{m, m->stream, {}};
    std::vector<int64_t> off(n_utt), len(n_utt);
    for (int u = 0; u < n_utt; ++u) {
        off[u] = offsets[u];
        len[u] = offsets[u + 1] - offsets[u];
        Q3_REQUIRE(len[u] >= 0, "mel: offsets must be non-decreasing");
    }
    Q3_REQUIRE(t_stride > 0, "mel: t_stride must be positive");
    const bool direct = out_space == Q3ASR_DEVICE;   // a device caller's buffer is written by the kernels themselves
    MelRun r = run_mel(c, audio, audio_space, off, len, sample_rate, t_stride, direct ? mel_out : nullptr, true);
    if (!direct) HIP_CHECK(hipMemcpyAsync(mel_out, r.raw, (size_t)n_utt * MEL_NMELS * r.t_stride * 4, hipMemcpyDeviceToHost, c.s));
    HIP_CHECK(hipStreamSynchronize(c.s));
    for (int u = 0; u < n_utt; ++u) n_frames[u] = r.frames[u];
}

void model_encode(q3asr_model* m, const float* mel, int t_stride, const int32_t* n_frames, int n_utt, int mel_space,
                  float* embeds_out, int64_t out_rows_capacity, int32_t* cu_tokens, int out_space) {
    Q3_REQUIRE(m->finalized, "encode: weights not finalized");
    Q3_REQUIRE(n_utt > 0 && mel && n_frames && embeds_out && cu_tokens, "encode: null argument or empty batch");
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    Ctx c{m, m->stream, {}};
    std::vector<int> frames(n_frames, n_frames + n_utt);
    int64_t total = 0;
    cu_tokens[0] = 0;
    for (int u = 0; u < n_utt; ++u) {
        Q3_REQUIRE(frames[u] >= 0 && frames[u] <= t_stride && frames[u] <= MEL_MAX_FRAMES, "encode: frame count out of range");
        total += model_output_length(m->cfg, frames[u]);
        cu_tokens[u + 1] = (int32_t)total;
    }
    Q3_REQUIRE(total <= out_rows_capacity, "encode: output buffer too small");
    const float* d_mel = mel;
    if (mel_space == Q3ASR_HOST) {
        float* buf = c.get<float>((size_t)n_utt * MEL_NMELS * t_stride);
        HIP_CHECK(hipMemcpyAsync(buf, mel, (size_t)n_utt * MEL_NMELS * t_stride * 4, hipMemcpyHostToDevice, c.s));
        d_mel = buf;
    }
    float* d_out = c.get<float>((size_t)std::max<int64_t>(total, 1) * m->cfg.enc_out_dim);
    run_encoder(c, d_mel, t_stride, nullptr, 1, frames, d_out);
    if (total > 0)
        HIP_CHECK(hipMemcpyAsync(embeds_out, d_out, (size_t)total * m->cfg.enc_out_dim * 4,
                                 out_space == Q3ASR_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, c.s));
    HIP_CHECK(hipStreamSynchronize(c.s));
}

// ------------------------------------------------------------------------------------------ transcribe
namespace {

// The decode-step projections run `rows` rows against wide weights, where the best (tile, split-K) pair depends
// on the row count and on the projection's shape.  The plan is a fixed function of (rows, N, K, weight format),
// evaluated per power-of-two bucket of the row count, so that equal requests are computed by equal kernels in
// every process and on every rank (the reference is deterministic for a given input): a table of the fastest
// measured variant for the known model shapes (decode_plan_table.h), a workgroup-count rule for any other shape.  All unsplit variants
// accumulate K in the same order with the same MFMA, so the tile never changes a result; the split count and the
// fused-versus-slab form do (within the bf16 tolerance).  Q3ASR_DECODE_TILE / _SPLIT / _NO_FUSE pin one variant;
// Q3ASR_DECODE_AUTOTUNE=1 replaces the rule by a measurement of every candidate on layer 0's weights (tuning runs).
// `cus` > 0: the projections run on a stream confined to that many compute units (0: the whole device).
int plan_bucket(int rows) {
    int b = 8;
    while (b < rows) b <<= 1;
    return b;
}

void tile_dims(int tile, int& bm, int& bn) {
    bm = tile == 3 ? 256 : (tile == 1 ? 64 : 128);
    bn = tile == 3 ? 256 : (tile == 0 ? 128 : 64);
}

q3asr_model::DecodePlan decode_plan_rule(const q3asr_model* m, int rows, GemmWFmt wf, int cus) {
    const int ncu = cus > 0 ? cus : device_cus();
    const DecLayerW& L = m->dec[0];
    const LinearW* W[4] = {&L.qkv, &L.o, &L.gate_up, &L.down};
    q3asr_model::DecodePlan plan;
    const char* env_tile = getenv("Q3ASR_DECODE_TILE");
    const char* env_split = getenv("Q3ASR_DECODE_SPLIT");
    const bool allow_fused = !getenv("Q3ASR_DECODE_NO_FUSE");
    const int R = plan_bucket(rows);
    for (int g = 0; g < 4; ++g) {
        const int N = W[g]->N, K = W[g]->K, ktiles = K / 64;
        // the largest tile that still gives every CU a workgroup without splitting; otherwise the tile that wastes the
        // fewest rows, split along K until about 256 workgroups stream the weights (at least four K tiles per split)
        int tile = -1, ks = 1;
        for (int t : {3, 0, 2, 1}) {
            if (t == 3 && (wf != W_BF16 || R <= 128)) continue;
            if (t == 1 && (R > 512 || g == 2)) continue;   // no SwiGLU epilogue on the 64x64 tile
            int bm, bn;
            tile_dims(t, bm, bn);
            if ((int64_t)cdiv(R, bm) * cdiv(N, bn) >= (t == 3 ? ncu * 3 / 4 : ncu)) { tile = t; break; }
        }
        if (tile < 0) {
            tile = R <= 64 && g != 2 ? 1 : 2;
            int bm, bn;
            tile_dims(tile, bm, bn);
            const int64_t wgs = (int64_t)cdiv(R, bm) * cdiv(N, bn);
            while (ks < 8 && wgs * ks < ncu && ktiles / (ks * 2) >= 4) ks *= 2;
        }
        bool want_fused = ks == 1;
        for (const DecodePlanEntry& e : DECODE_PLAN_TABLE)
            if (e.N == N && e.K == K && e.wfmt == (int)wf && e.bucket == R && e.cus == cus) {
                tile = e.tile;
                ks = std::max(1, e.split);
                want_fused = e.split == 0;
                break;
            }
        if (env_tile) tile = atoi(env_tile);
        if (env_split) { ks = std::max(1, atoi(env_split)); want_fused = atoi(env_split) == 0; }   // "0": fused form, "1": one slab
        const bool fused = want_fused && ks == 1 && allow_fused;
        if (tile == 1 && g == 2 && fused) tile = 2;
        plan.tile[g] = tile;
        plan.ks[g] = ks;
        plan.fused[g] = fused;
    }
    return plan;
}

// Times every candidate of the four projections at `b` rows.  Launch i of a candidate uses layer i's weights, so the
// weights come from HBM as they do in a decode step (28 layers of the 1.7B model are 3.4 GB, far beyond the caches),
// while the activations stay hot as the previous kernel's output would be.
q3asr_model::DecodePlan decode_plan_measured(Ctx& c, int b, GemmWFmt wf, int cus, hipStream_t s) {
    q3asr_model* m = c.m;
    const int layers = (int)m->dec.size();
    auto weight = [&](int l, int g) -> const LinearW& {
        const DecLayerW& L = m->dec[l];
        return g == 0 ? L.qkv : g == 1 ? L.o : g == 2 ? L.gate_up : L.down;
    };
    q3asr_model::DecodePlan plan;
    const char* env_tile = getenv("Q3ASR_DECODE_TILE");
    const char* env_split = getenv("Q3ASR_DECODE_SPLIT");
    int max_n = 0, max_k = 0;
    for (int g = 0; g < 4; ++g) { max_n = std::max(max_n, weight(0, g).N); max_k = std::max(max_k, weight(0, g).K); }
    const auto mark = m->arena.mark();
    bf16_t* A = c.get<bf16_t>((size_t)b * max_k);
    float* slab = c.get<float>((size_t)8 * b * max_n);
    launch_fill_bf16(A, (size_t)b * max_k, 0x51u, s);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    // epilogue of the unsplit, fused form of each projection: bf16 q/k/v rows, residual add, SwiGLU, residual add
    const GemmOut fused_out[4] = {OUT_BF16, OUT_F32_ACC, OUT_SWIGLU, OUT_F32_ACC};
    const bool allow_fused = !getenv("Q3ASR_DECODE_NO_FUSE");
    for (int g = 0; g < 4; ++g) {
        const int N = weight(0, g).N, K = weight(0, g).K;
        plan.tile[g] = b > 64 ? 2 : 1;
        plan.ks[g] = 1;
        plan.fused[g] = 0;
        double best = 1e30;
        for (int tile : {1, 2, 0, 3}) {
            if ((tile == 3 && (wf != W_BF16 || b <= 128)) || (tile == 1 && b > 512)) continue;
            if (env_tile && atoi(env_tile) != tile) continue;
            for (int ks : {0, 1, 2, 4, 8}) {  // 0: the fused form
                if (ks > 1 && ks * 4 > K / 64) continue;  // at least four K tiles per split
                if (env_split && atoi(env_split) != ks) continue;
                if (ks == 0 && (!allow_fused || (g == 2 && tile == 1))) continue;  // no SwiGLU epilogue on the 64x64 tile
                const GemmOut out = ks == 0 ? fused_out[g] : OUT_SLAB;
                float ms = 1e30f;
                for (int pass = 0; pass < 3; ++pass) {   // pass 0 warms up; the faster of the other two counts
                    if (pass) HIP_CHECK(hipEventRecord(e0, s));
                    for (int l = 0; l < layers; ++l) {
                        GemmParams p = gp(c, A, K, weight(l, g), b);
                        p.out = slab; p.ldo = g == 2 && ks == 0 ? N / 2 : N;
                        p.ksplit = std::max(ks, 1); p.slab_stride = (int64_t)b * N; p.cus = cus;
                        launch_gemm(p, A_PLAIN, out, wf, tile, s);
                    }
                    if (pass) {
                        HIP_CHECK(hipEventRecord(e1, s));
                        HIP_CHECK(hipEventSynchronize(e1));
                        float t = 0;
                        HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
                        ms = std::min(ms, t);
                    }
                }
                // a split's consumer reads the slabs back (~3 TB/s); gate/up partials also need the reduce kernel
                const double extra_us = ks == 0 ? 0.0 : ks * b * (double)N * 4.0 / 3.0e6 + (g == 2 ? 4.0 : 0.0);
                const double us = ms * 1000.0 / layers + extra_us;
                if (getenv("Q3ASR_VERBOSE") && atoi(getenv("Q3ASR_VERBOSE")) > 1)
                    fprintf(stderr, "q3asr: cus %d rows %d proj %d tile %d split %d: %.1f us (+%.1f us for its partial sums)\n", cus, b, g, tile, ks,
                            ms * 1000.0 / layers, extra_us);
                if (us < best) { best = us; plan.tile[g] = tile; plan.ks[g] = std::max(ks, 1); plan.fused[g] = ks == 0; }
            }
        }
    }
    HIP_CHECK(hipStreamSynchronize(s));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    m->arena.restore(mark);
    return plan;
}

q3asr_model::DecodePlan decode_plan(Ctx& c, int rows, GemmWFmt wf, int cus, hipStream_t s) {
    q3asr_model* m = c.m;
    const int bucket = plan_bucket(rows), key = bucket + (cus << 20);
    const char* env_auto = getenv("Q3ASR_DECODE_AUTOTUNE");
    const bool measured = env_auto && atoi(env_auto) != 0;
    q3asr_model::DecodePlan plan;
    bool fresh = true;
    if (measured) {   // a measurement is kept per bucket; the rule is cheap and follows the pinning variables of each call
        const auto hit = m->decode_plans.find(key);
        fresh = hit == m->decode_plans.end();
        plan = fresh ? decode_plan_measured(c, rows, wf, cus, s) : hit->second;
        m->decode_plans[key] = plan;
    } else {
        plan = decode_plan_rule(m, rows, wf, cus);
        fresh = m->plans_reported.insert(key).second;
    }
    if (fresh && getenv("Q3ASR_VERBOSE"))
        fprintf(stderr, "q3asr: decode plan for up to %d rows on %d CUs%s (tile/split, split 0 = fused epilogue): qkv %d/%d, o %d/%d, gate_up %d/%d, down %d/%d\n",
                bucket, cus > 0 ? cus : device_cus(), measured ? " (measured)" : "", plan.tile[0], plan.fused[0] ? 0 : plan.ks[0], plan.tile[1],
                plan.fused[1] ? 0 : plan.ks[1], plan.tile[2], plan.fused[2] ? 0 : plan.ks[2], plan.tile[3], plan.fused[3] ? 0 : plan.ks[3]);
    return plan;
}

// Compute units the decode attention of a group of `b` utterances is confined to (0: one stream, the whole device for
// every kernel).  Q3ASR_DECODE_ATT_CUS pins the share (tests, tuning).  Measured at 2048 clips against one stream:
// 1.7B bf16 -9 % decode time with 160 of 256 CUs (128: -3 %, 144: slower, the 256-wide projection tiles fill 96 CUs in
// whole rounds), 0.6B bf16 -2.5 %, 0.6B 4-bit -4.6 %; the 1.7B 4-bit projections (dequantisation-bound) are too slow
// on 96 CUs (+19 %) and take half of the device (-4 %).  By group size, decode time of the split against one stream:
// 1.7B bf16 256 clips +39 %, 384 +11 %, 512 +13 %, 640 -11 %, 768 -14 %, 896 -11 % (two halves of 256 rows are one row
// tile each for the 256-wide projection kernel, which then leaves most of its 96 CUs idle); 0.6B bf16 256 +13 %,
// 512 -2 %; 0.6B 4-bit 256 +10 %, 512 -9 %.
int decode_att_cus(const q3asr_model* m, int b) {
    const int cus_all = device_cus();
    const char* env_att = getenv("Q3ASR_DECODE_ATT_CUS");
    const bool heavy_q4 = m->cfg.weight_format == 1 && m->cfg.hidden >= 2048;
    int att_cus = (env_att ? atoi(env_att) : (b >= (heavy_q4 ? SPLIT_MIN_ROWS_HEAVY_Q4 : SPLIT_MIN_ROWS) ? (heavy_q4 ? cus_all / 2 : cus_all * 5 / 8) : 0)) & ~7;
    if (b < 2 || cus_all % 8 != 0 || att_cus < 8 || cus_all - att_cus < 8) att_cus = 0;
    return att_cus;
}

// Streams confined to complementary sets of compute units: mask bit i selects CU i / 8 of XCD i % 8, so the first
// `att_cus` bits (a multiple of 8) take the same number of CUs from every XCD (a mask that leaves an XCD without a
// CU is not accepted by the runtime).
// Q3ASR_TRACE=1: the decode loop's host-side progress on stderr (where a stalled call stands)
bool trace_on() {
    static const bool on = [] { const char* e = getenv("Q3ASR_TRACE"); return e && atoi(e) != 0; }();
    return on;
}
#define Q3_TRACE(...) do { if (trace_on()) { fprintf(stderr, "[q3asr] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

// The CU-confined streams for a share of a device.  They belong to the process, not to a model, and live until the
// process ends: the runtime does not give a destroyed CU-masked queue's slot back, and the 17th
// hipExtStreamCreateWithCUMask of a process that has destroyed 16 such streams never returns (measured with the
// decode loop's event chain on two masked streams, destroyed and re-created under another mask between rounds:
// `scripts/probe_cu_mask.hip chain`; with the streams kept the same loop runs on).  A process that loads and unloads
// models all day (or a test run) therefore must not create streams per model.  Models on one device share them; calls
// of different models then queue behind one another on these streams, each ordered by its own events.
struct SplitStreamCache {
    std::mutex mu;
    std::map<std::pair<int, int>, q3asr_model::SplitStreams> by_share;   // (device, attention CUs)
};
SplitStreamCache& split_stream_cache() {
    static SplitStreamCache* c = new SplitStreamCache();   // never destructed: no HIP call from a static destructor
    return *c;
}

void ensure_split_streams(q3asr_model* m, int att_cus, bool second_att_stream) {
    SplitStreamCache& cache = split_stream_cache();
    std::lock_guard<std::mutex> lk(cache.mu);
    auto& st = cache.by_share[{m->device, att_cus}];
    const int cus = device_cus(), words = (cus + 31) / 32;
    std::vector<uint32_t> ma(words, 0), mb(words, 0);
    for (int i = 0; i < cus; ++i) (i < att_cus ? ma : mb)[i >> 5] |= 1u << (i & 31);
    if (!st.att) {
        Q3_TRACE("split streams: creating the streams for %d attention CUs", att_cus);
        Q3_REQUIRE(cus % 8 == 0 && att_cus % 8 == 0 && att_cus >= 8 && cus - att_cus >= 8, "decode: bad compute-unit split");
        Q3_REQUIRE(cache.by_share.size() <= 4, "decode: more than four different compute-unit shares asked for in one process");
        HIP_CHECK(hipExtStreamCreateWithCUMask(&st.att, (uint32_t)words, ma.data()));
        HIP_CHECK(hipExtStreamCreateWithCUMask(&st.mm, (uint32_t)words, mb.data()));
    }
    if (second_att_stream && !st.att2) HIP_CHECK(hipExtStreamCreateWithCUMask(&st.att2, (uint32_t)words, ma.data()));
    m->stream_att = st.att;
    m->stream_att2 = st.att2;
    m->stream_mm = st.mm;
    m->split_att_cus = att_cus;
}

struct Events {
    hipEvent_t e[5];
    Events() { for (auto& x : e) HIP_CHECK(hipEventCreate(&x)); }
    ~Events() { for (auto& x : e) (void)hipEventDestroy(x); }
};

void transcribe_group(Ctx& c, const q3asr_request* req, q3asr_result* res, const std::vector<int>& idx,
                      const std::vector<int64_t>& tap_a_off, const std::vector<int64_t>& tap_s_off) {
    q3asr_model* m = c.m;
    const q3asr_config& cfg = m->cfg;
    const int b = (int)idx.size(), H = cfg.hidden, hd = cfg.head_dim, nh = cfg.heads, nkv = cfg.kv_heads;
    const int QKV = (nh + 2 * nkv) * hd, max_tokens = req->max_tokens;
    const GemmWFmt wf = cfg.weight_format == 1 ? W_Q4 : W_BF16;
    Events ev;
    HIP_CHECK(hipEventRecord(ev.e[0], c.s));

    // ---- mel ----
    std::vector<int64_t> off(b), len(b);
    for (int i = 0; i < b; ++i) {
        off[i] = req->offsets[idx[i]];
        len[i] = req->offsets[idx[i] + 1] - off[i];
    }
    MelRun mel = run_mel(c, req->audio, req->audio_space, off, len, req->sample_rate, 0);
    HIP_CHECK(hipEventRecord(ev.e[1], c.s));

    // ---- encoder ----
    std::vector<int> ntok(b);
    int64_t tot_a = 0;
    for (int i = 0; i < b; ++i) {
        ntok[i] = model_output_length(cfg, mel.frames[i]);
        tot_a += ntok[i];
        res->audio_seconds += (double)std::min<int64_t>(mel.n16[i], (int64_t)MEL_MAX_FRAMES * MEL_HOP) / 16000.0;
    }
    float* audio_emb = c.get<float>((size_t)std::max<int64_t>(tot_a, 1) * cfg.enc_out_dim);
    run_encoder(c, mel.raw, mel.t_stride, mel.dmax, 0, mel.frames, audio_emb);
    HIP_CHECK(hipEventRecord(ev.e[2], c.s));

    // ---- prompt (Qwen3ASR.swift:282-317) ----
    std::vector<int> ids, row_utt, row_pos, utt_row0(b), utt_rows(b), utt_a0(b), utt_na(b), utt_arow0(b), last_rows(b), seen0(b, 0);
    int arow = 0, cap = 0;
    for (int i = 0; i < b; ++i) {
        const int u = idx[i];
        utt_row0[i] = (int)ids.size();
        const int head[] = {TOK_IM_START, TOK_SYSTEM, TOK_NEWLINE, TOK_IM_END, TOK_NEWLINE, TOK_IM_START, TOK_USER, TOK_NEWLINE, TOK_AUDIO_START};
        for (int t : head) ids.push_back(t);
        utt_a0[i] = 9;
        for (int k = 0; k < ntok[i]; ++k) ids.push_back(TOK_AUDIO_PAD);
        const int tail[] = {TOK_AUDIO_END, TOK_IM_END, TOK_NEWLINE, TOK_IM_START, TOK_ASSISTANT, TOK_NEWLINE};
        for (int t : tail) ids.push_back(t);
        if (req->lang_ids && req->lang_offsets && req->lang_offsets[u + 1] > req->lang_offsets[u]) {
            bool has = false;
            for (int k = req->lang_offsets[u]; k < req->lang_offsets[u + 1]; ++k) {
                Q3_REQUIRE(req->lang_ids[k] >= 0 && req->lang_ids[k] < cfg.vocab, "transcribe: language prefix id out of range");
                ids.push_back(req->lang_ids[k]);
                has |= req->lang_ids[k] == TOK_ASR_TEXT;
            }
            if (!has) ids.push_back(TOK_ASR_TEXT);
            seen0[i] = 1;  // :565-566
        }
        utt_rows[i] = (int)ids.size() - utt_row0[i];
        utt_na[i] = ntok[i];
        utt_arow0[i] = arow;
        arow += ntok[i];
        for (int k = 0; k < utt_rows[i]; ++k) { row_utt.push_back(i); row_pos.push_back(k); }
        last_rows[i] = (int)ids.size() - 1;
        cap = std::max(cap, utt_rows[i] + max_tokens);
        if (res->n_audio_tokens) res->n_audio_tokens[u] = ntok[i];
    }
    const int R = (int)ids.size();
    cap += 16;  // spare rows: the decode-attention kernel reads whole 16-key tiles
    Q3_REQUIRE(cap <= m->rope_max_pos, "transcribe: prompt + max_tokens exceeds the RoPE table");
    Q3_REQUIRE(cfg.enc_out_dim == H, "transcribe: encoder output dim must equal decoder hidden size");

    // ---- buffers ----
    const int64_t head_stride = (int64_t)cap * hd, utt_stride = head_stride * nkv, layer_stride = utt_stride * b;
    bf16_t* kc = c.get<bf16_t>((size_t)layer_stride * cfg.layers);
    bf16_t* vc = c.get<bf16_t>((size_t)layer_stride * cfg.layers);
    float* hid = c.get<float>((size_t)R * H);
    const int* d_ids = c.up(ids);
    const int* d_row_utt = c.up(row_utt);
    const int* d_row_pos = c.up(row_pos);
    const int* d_row0 = c.up(utt_row0);
    const int* d_rows = c.up(utt_rows);
    const int* d_last = c.up(last_rows);
    double* stats = c.get<double>((size_t)b * 4);
    HIP_CHECK(hipMemsetAsync(stats, 0, (size_t)b * 32, c.s));
    int* st_tokens = c.get<int>((size_t)b * max_tokens);
    int* st_ntok = c.get<int>(b);
    int* st_done = c.get<int>(b);
    int* st_seen = c.up(seen0);
    int* st_next = c.get<int>(b);
    int* st_len = c.up(utt_rows);
    int* st_active = c.get<int>(64);
    int* st_argmax = c.get<int>(b);
    // row -> utterance of the current decode rows, and the list the latest step wrote (ping-pong)
    std::vector<int> ident(b);
    std::iota(ident.begin(), ident.end(), 0);
    int* st_map[2] = {c.up(ident), c.get<int>(b)};
    long long* st_keys = c.get<long long>((size_t)2 * (max_tokens + 2));  // per half and step: keys the step's attention covers
    HIP_CHECK(hipMemsetAsync(st_keys, 0, (size_t)2 * (max_tokens + 2) * sizeof(long long), c.s));
    HIP_CHECK(hipMemsetAsync(st_ntok, 0, (size_t)b * 4, c.s));
    HIP_CHECK(hipMemsetAsync(st_done, 0, (size_t)b * 4, c.s));
    const int* d_forced = nullptr;
    if (req->forced_tokens) {
        std::vector<int> f((size_t)b * max_tokens);
        for (int i = 0; i < b; ++i) memcpy(&f[(size_t)i * max_tokens], req->forced_tokens + (size_t)idx[i] * max_tokens, (size_t)max_tokens * 4);
        for (int t : f) Q3_REQUIRE(t >= 0 && t < cfg.vocab, "transcribe: forced token id out of range");
        d_forced = c.up(f);
    }
    auto head_tile_for = [&](int rows) {  // 256x256 once the rows fill more than half of the tile
        int t = gemm_pick_tile(rows, cfg.vocab);
        if (t == 1 || (t == 3 && (wf != W_BF16 || (rows <= 128 && q3_knobs().force_tile < 0)))) t = 0;
        return t;
    };
    const int max_parts = gemm_argmax_parts(cfg.vocab, 1);  // the narrowest tile writes the most partials
    float* part_val = c.get<float>((size_t)b * max_parts);
    int* part_idx = c.get<int>((size_t)b * max_parts);
    bf16_t* x_last = c.get<bf16_t>((size_t)b * H);
    float* logits = req->tap_logits && req->tap_steps > 0 ? c.get<float>((size_t)b * cfg.vocab) : nullptr;

    const bool qemb = cfg.weight_format == 1;
    SpliceParams sp{d_ids, d_row_utt, d_row0, d_rows, c.up(utt_a0), c.up(utt_na), c.up(utt_arow0), R, b, H,
                    qemb ? nullptr : (const bf16_t*)m->embed.w, qemb ? (const uint32_t*)m->embed.w : nullptr, m->embed.scales,
                    m->embed.biases, cfg.group_size, audio_emb, stats, cfg.embed_scale, hid};
    if (req->inject_inputs_embeds) {
        for (int i = 0; i < b; ++i)
            HIP_CHECK(hipMemcpyAsync(hid + (int64_t)utt_row0[i] * H, req->inject_inputs_embeds + tap_s_off[idx[i]] * H,
                                     (size_t)utt_rows[i] * H * 4, hipMemcpyHostToDevice, c.s));
    } else {
        launch_splice(sp, c.s);
    }
    if (req->tap_audio_embeds || req->tap_inputs_embeds) {
        for (int i = 0; i < b; ++i) {
            if (req->tap_audio_embeds && ntok[i])
                HIP_CHECK(hipMemcpyAsync(req->tap_audio_embeds + tap_a_off[idx[i]] * H, audio_emb + (int64_t)utt_arow0[i] * H,
                                         (size_t)ntok[i] * H * 4, hipMemcpyDeviceToHost, c.s));
            if (req->tap_inputs_embeds)
                HIP_CHECK(hipMemcpyAsync(req->tap_inputs_embeds + tap_s_off[idx[i]] * H, hid + (int64_t)utt_row0[i] * H,
                                         (size_t)utt_rows[i] * H * 4, hipMemcpyDeviceToHost, c.s));
        }
    }

    const float scale = 1.0f / sqrtf((float)hd);
    const int half = std::min(cfg.rotary_dim, hd) / 2;

    // LM head on `rows` normalised rows -> st_argmax (and optional logits tap)
    // (utterances r0 .. r0 + rows - 1 of the group, on stream s)
    int cur_cus = 0;   // compute units of the stream the decode projections are issued on (0: the whole device)
    auto lm_head = [&](const bf16_t* x, int r0, int rows, int step, hipStream_t s) {
        const bool tap = logits && step < req->tap_steps;
        const int head_tile = head_tile_for(rows);
        GemmParams p = gp(c, x, H, m->embed, rows);
        p.cus = cur_cus;
        if (tap) {
            float* lg = logits + (size_t)r0 * cfg.vocab;
            p.out = lg; p.ldo = cfg.vocab;
            launch_gemm(p, A_PLAIN, OUT_F32, wf, head_tile, s);
            launch_argmax_rows(lg, rows, cfg.vocab, st_argmax + r0, s);
            for (int i = 0; i < rows; ++i)
                HIP_CHECK(hipMemcpyAsync(req->tap_logits + ((size_t)idx[r0 + i] * req->tap_steps + step) * cfg.vocab,
                                         lg + (size_t)i * cfg.vocab, (size_t)cfg.vocab * 4, hipMemcpyDeviceToHost, s));
        } else {
            const int nparts = gemm_argmax_parts(cfg.vocab, head_tile);
            p.part_val = part_val + (size_t)r0 * max_parts; p.part_idx = part_idx + (size_t)r0 * max_parts; p.nparts = nparts;
            launch_gemm(p, A_PLAIN, OUT_ARGMAX, wf, head_tile, s);
            launch_argmax_reduce(p.part_val, p.part_idx, nparts, rows, st_argmax + r0, s);
        }
    };
    const int rules = req->forced_tokens ? (req->stop_rules == 2 ? 2 : 0) : (req->stop_rules ? 1 : 0);
    // bookkeeping of one step over `rows` rows that start at row r0 and belong to utterances map[0..rows)
    auto step = [&](int advance, int r0, int rows, int* active, const int* map, int* map_out, long long* keys_out, hipStream_t s) {
        StepParams sp2{st_argmax + r0, st_tokens, st_ntok, st_done, st_seen, st_next, st_len, active, d_forced, rows, max_tokens,
                       advance, rules};
        sp2.slot = map; sp2.live_out = map_out; sp2.keys_out = keys_out;
        launch_step(sp2, s);
    };

    // ---- prefill (QuantizedTextDecoder.swift:384-446 over the packed prompts) ----
    {
        const auto mark = m->arena.mark();
        bf16_t* xn = c.get<bf16_t>((size_t)R * H);
        bf16_t* qkv = c.get<bf16_t>((size_t)R * QKV);
        bf16_t* att = c.get<bf16_t>((size_t)R * nh * hd);
        bf16_t* act = c.get<bf16_t>((size_t)R * cfg.inter);
        std::vector<AttnSeq> seqs;
        std::vector<AttnTile> tiles;
        for (int i = 0; i < b; ++i) {
            seqs.push_back({utt_row0[i], utt_rows[i], (int64_t)i * utt_stride});
            for (int q0 = 0; q0 < utt_rows[i]; q0 += 64) tiles.push_back({i, q0});
        }
        const AttnSeq* d_seqs = c.up(seqs);
        const AttnTile* d_tiles = c.up(tiles);
        double prefill_attn_flops = 0;  // causal: half of the dense 4 S^2 D per head
        for (auto& sq : seqs) prefill_attn_flops += 2.0 * sq.len * sq.len * hd * nh;
        // The prefill's only output besides the cache is the last position's hidden state (Qwen3ASR.swift:386-388 takes
        // logits[-1]).  Every layer needs all rows for its successor, except the last one: its K / V rows go to the cache
        // for the decode steps, but only the last row of every prompt goes on through attention, o, and the MLP -- one
        // 64-query tile and one row per utterance instead of S rows.  Each kept value is computed as before (the GEMM
        // tile never changes a sum), so the tokens are the same bits.  Q3ASR_PREFILL_FULL_LAST=1 computes every row.
        const char* env_full = getenv("Q3ASR_PREFILL_FULL_LAST");
        const bool short_last = !(env_full && atoi(env_full) != 0);
        std::vector<AttnTile> last_tiles;
        double last_attn_flops = 0;
        for (int i = 0; i < b; ++i) {
            const int q0 = (utt_rows[i] - 1) / 64 * 64;
            last_tiles.push_back({i, q0});
            last_attn_flops += 4.0 * (utt_rows[i] - q0) * (q0 + 0.5 * (utt_rows[i] - q0)) * hd * nh;
        }
        const AttnTile* d_last_tiles = short_last ? c.up(last_tiles) : nullptr;
        bf16_t* att_last = short_last ? c.get<bf16_t>((size_t)b * nh * hd) : nullptr;
        for (int l = 0; l < cfg.layers; ++l) {
            const DecLayerW& L = m->dec[l];
            launch_rmsnorm(hid, L.in_norm, xn, R, H, cfg.rms_eps, nullptr, nullptr, 0, 0, c.s);
            {
                GemmParams p = gp(c, xn, H, L.qkv, R);
                p.out = qkv; p.ldo = QKV;
                launch_gemm(p, A_PLAIN, OUT_BF16, wf, -1, c.s);
            }
            RopeParams rp{qkv, QKV, nullptr, 0, 0, qkv, QKV, R, nh, nkv, L.q_norm, L.k_norm, cfg.rms_eps, m->rope_cos, m->rope_sin, half,
                          cfg.rope_interleaved, m->rope_max_pos, d_row_utt, d_row_pos, st_len, kc + l * layer_stride, vc + l * layer_stride,
                          utt_stride, head_stride};
            launch_rope(rp, c.s);
            if (short_last && l + 1 == cfg.layers) {
                FlashParams fp{qkv, QKV, kc + l * layer_stride, vc + l * layer_stride, hd, head_stride, att, (int64_t)nh * hd, d_seqs,
                               d_last_tiles, b, nh, nh / nkv, scale, 1};
                launch_flash_attn(fp, hd, c.s, last_attn_flops);
                launch_gather_rows_bf16(att, (int64_t)nh * hd, d_last, att_last, b, nh * hd, c.s);
                {
                    GemmParams p = gp(c, att_last, (int64_t)nh * hd, L.o, b);
                    p.out = hid; p.ldo = H; p.rowmap = d_last;
                    launch_gemm(p, A_PLAIN, OUT_F32_ACC, wf, -1, c.s);
                }
                launch_rmsnorm(hid, L.post_norm, xn, b, H, cfg.rms_eps, d_last, nullptr, 0, 0, c.s);   // compact rows
                {
                    GemmParams p = gp(c, xn, H, L.gate_up, b);
                    p.out = act; p.ldo = cfg.inter;
                    launch_gemm(p, A_PLAIN, OUT_SWIGLU, wf, -1, c.s);
                }
                {
                    GemmParams p = gp(c, act, cfg.inter, L.down, b);
                    p.out = hid; p.ldo = H; p.rowmap = d_last;
                    launch_gemm(p, A_PLAIN, OUT_F32_ACC, wf, -1, c.s);
                }
                break;
            }
            FlashParams fp{qkv, QKV, kc + l * layer_stride, vc + l * layer_stride, hd, head_stride, att, (int64_t)nh * hd, d_seqs, d_tiles,
                           (int)tiles.size(), nh, nh / nkv, scale, 1};
            launch_flash_attn(fp, hd, c.s, prefill_attn_flops);
            {
                GemmParams p = gp(c, att, (int64_t)nh * hd, L.o, R);
                p.out = hid; p.ldo = H;
                launch_gemm(p, A_PLAIN, OUT_F32_ACC, wf, -1, c.s);
            }
            launch_rmsnorm(hid, L.post_norm, xn, R, H, cfg.rms_eps, nullptr, nullptr, 0, 0, c.s);
            {
                GemmParams p = gp(c, xn, H, L.gate_up, R);
                p.out = act; p.ldo = cfg.inter;
                launch_gemm(p, A_PLAIN, OUT_SWIGLU, wf, -1, c.s);
            }
            {
                GemmParams p = gp(c, act, cfg.inter, L.down, R);
                p.out = hid; p.ldo = H;
                launch_gemm(p, A_PLAIN, OUT_F32_ACC, wf, -1, c.s);
            }
        }
        launch_rmsnorm(hid, m->final_norm, x_last, b, H, cfg.rms_eps, d_last, nullptr, 0, 0, c.s);  // :386, last position only
        lm_head(x_last, 0, b, 0, c.s);
        step(0, 0, b, st_active, st_map[0], st_map[1], st_keys + 1, c.s);
        m->arena.restore(mark);
    }
    HIP_CHECK(hipEventRecord(ev.e[3], c.s));

    // ---- greedy decode (Qwen3ASR.swift:567-632) ----
    {
        // The step's attention streams the KV cache: it is HBM-bound, and one compute unit sustains ~55 GB/s of that
        // stream, so 160 of the 256 CUs already draw what the memory gives (6.8 against 7.0 TB/s in a bare read loop).
        // The step's projections are MFMA-bound.  A large group is therefore decoded as two halves: the attention
        // kernels of both run on a stream confined to `att_cus` CUs, everything else on a stream confined to the
        // remaining ones, chained by events so that one half's attention runs beside the other half's projections
        // (plain concurrent streams do not overlap them: each kernel fills every CU's registers on its own).
        // The share is a fixed function of the model and the group size (decode_att_cus).
        const int att_cus = decode_att_cus(m, b);
        const int mm_cus = att_cus ? device_cus() - att_cus : 0;
        int n_half = att_cus ? 2 : 1;
        const bool was_split = n_half == 2;
        bool chain = was_split;   // attention and the other kernels are on different streams
        // The halves' attention launches on one stream or on two (both confined to the attention's CUs): on two, a
        // launch starts filling CUs while the other half's previous launch drains, instead of ~11 us after its end.
        // (Q3ASR_ATT_STREAMS=1: one attention stream for both halves.  A second stream for the second half's projections
        // was measured as well and changes nothing: 12 379 / 12 194 ms of decode with it, 12 370 without.)
        const char* env_as = getenv("Q3ASR_ATT_STREAMS");
        const int att_streams = was_split && !(env_as && atoi(env_as) == 1) ? 2 : 1;
        if (was_split) ensure_split_streams(m, att_cus, att_streams == 2);
        // The reference computes nothing for a clip once it has stopped (:571-574).  Here a stopped utterance is skipped by
        // the attention kernel at once (its done flag) and leaves the step's rows at the next poll: the step kernel
        // writes the list of utterances still running, and switching the row -> utterance map to that list shrinks
        // every per-row kernel of the following steps (nothing is moved: the cache and all per-utterance state are
        // indexed by utterance).  Off for the logits tap (whose host copies are per row).
        const char* env_retire = getenv("Q3ASR_RETIRE");
        const bool skip_done = !(env_retire && atoi(env_retire) == 0);   // Q3ASR_RETIRE=0: every row is computed until the group ends
        const bool retire = !logits && skip_done;
        const char* env_gemv = getenv("Q3ASR_DECODE_GEMV");
        const char* env_auto = getenv("Q3ASR_DECODE_AUTOTUNE");
        const bool autotune = env_auto && atoi(env_auto) != 0;
        struct Half {
            int r0, n;      // first row and row count at the start (= utterances of the half)
            int rows;       // rows still decoded (0: every utterance of the half has stopped)
            int cur;        // which of st_map holds the half's row -> utterance list
            hipStream_t s, sa;   // stream of the attention kernels (sa) and of everything else (s)
            float* hid; bf16_t *xn, *qkvb, *qd, *att, *act, *x_last; float* slab;
            int* active_dev;
            long long* keys;
            q3asr_model::DecodePlan plan;
            bool gemv;  // a handful of rows: projections as fused GEMV kernels (gemv.hip), no partial-sum slabs
            bool pre_issued;   // the next step's embedding, input norm and q/k/v projection are already enqueued
        } hv[2];
        auto gemv_ok = [&](int rows) {
            return !(env_gemv && atoi(env_gemv) == 0) && gemv_supports(rows, QKV, H, GEMV_NORM_BF16, wf) &&
                   gemv_supports(rows, H, nh * hd, GEMV_ACC_F32, wf) && gemv_supports(rows, 2 * cfg.inter, H, GEMV_NORM_SWIGLU, wf) &&
                   gemv_supports(rows, H, cfg.inter, GEMV_ACC_F32, wf);
        };
        // Few rows, wide weights: a projection may be split along K so that enough workgroups stream the
        // weights; the partial sums are reduced by whichever kernel consumes them next.
        auto set_rows = [&](Half& x, int rows) {
            x.rows = rows;
            x.gemv = gemv_ok(rows);
            x.plan = x.gemv ? q3asr_model::DecodePlan{} : decode_plan(c, rows, wf, cur_cus, x.s);
            x.pre_issued = false;
        };
        auto slab_need = [&](const q3asr_model::DecodePlan& pl, int rows) {
            size_t e = (size_t)pl.ks[0] * rows * QKV;
            e = std::max(e, (size_t)pl.ks[1] * rows * H);
            e = std::max(e, (size_t)pl.ks[2] * rows * 2 * cfg.inter);
            return std::max(e, (size_t)pl.ks[3] * rows * H);
        };
        cur_cus = mm_cus;
        for (int h = 0; h < n_half; ++h) {
            Half& x = hv[h];
            x.r0 = h ? hv[0].n : 0;
            x.n = n_half == 1 ? b : (h ? b - hv[0].n : (b + 1) / 2);
            x.cur = 0;
            x.s = was_split ? m->stream_mm : c.s;
            x.sa = chain ? (h && att_streams == 2 ? m->stream_att2 : m->stream_att) : c.s;
            x.hid = c.get<float>((size_t)x.n * H);
            x.xn = c.get<bf16_t>((size_t)x.n * H);
            x.qd = c.get<bf16_t>((size_t)x.n * nh * hd);
            x.qkvb = c.get<bf16_t>((size_t)x.n * QKV);
            x.att = c.get<bf16_t>((size_t)x.n * nh * hd);
            x.act = c.get<bf16_t>((size_t)x.n * cfg.inter);
            x.x_last = x_last + (size_t)x.r0 * H;
            x.active_dev = st_active + h;
            x.keys = st_keys + (size_t)h * (max_tokens + 2);
            // the slab must hold the partial sums of every plan the half can pass through while it shrinks
            // (on its share of the CUs and, once the halves have been merged, on the whole device)
            size_t slab_elems = 256;
            if (autotune) {
                slab_elems = (size_t)8 * x.n * std::max(std::max(QKV, 2 * cfg.inter), H);
            } else {
                for (int top = 8;; top <<= 1) {
                    const int r = std::min(top, x.n);
                    if (!gemv_ok(r)) {
                        slab_elems = std::max(slab_elems, slab_need(decode_plan_rule(m, r, wf, mm_cus), r));
                        if (was_split) slab_elems = std::max(slab_elems, slab_need(decode_plan_rule(m, r, wf, 0), r));
                    }
                    if (top >= x.n) break;
                }
            }
            x.slab = c.get<float>(slab_elems);
        }
        int* st_tmp = was_split ? c.get<int>(b) : nullptr;   // staging of the merged row -> utterance list
        // One projection: split-K partial sums into the slab, or (fused plan) unsplit with the consumer's
        // work in the epilogue, writing `fused_dst` with row stride `fused_ld`.  Returns the number of slabs
        // the consumer still has to reduce (0 when fused).
        auto proj = [&](const Half& x, const bf16_t* A, int64_t lda, const LinearW& W, int which, GemmOut fused_kind, void* fused_dst,
                        int64_t fused_ld) {
            GemmParams p = gp(c, A, lda, W, x.rows);
            p.cus = cur_cus;
            if (x.plan.fused[which]) {
                p.out = fused_dst; p.ldo = fused_ld;
                launch_gemm(p, A_PLAIN, fused_kind, wf, x.plan.tile[which], x.s);
                return 0;
            }
            p.out = x.slab; p.ldo = W.N; p.ksplit = x.plan.ks[which]; p.slab_stride = (int64_t)x.rows * W.N;
            launch_gemm(p, A_PLAIN, OUT_SLAB, wf, x.plan.tile[which], x.s);
            return x.plan.ks[which];
        };
        // per half and layer: "q/k/v rows ready" and "attention done" events between the two streams
        if ((int)m->chain_events.size() < 4 * cfg.layers + 6) {
            const size_t old = m->chain_events.size();
            m->chain_events.resize(4 * cfg.layers + 6);
            for (size_t i = old; i < m->chain_events.size(); ++i) HIP_CHECK(hipEventCreateWithFlags(&m->chain_events[i], hipEventDisableTiming));
        }
        hipEvent_t* ev_pre[2] = {m->chain_events.data(), m->chain_events.data() + cfg.layers};
        hipEvent_t* ev_attn[2] = {m->chain_events.data() + 2 * cfg.layers, m->chain_events.data() + 3 * cfg.layers};
        hipEvent_t ev_fork = m->chain_events[4 * cfg.layers], ev_join = m->chain_events[4 * cfg.layers + 1];
        hipEvent_t* ev_xlast = m->chain_events.data() + 4 * cfg.layers + 2;   // per half: final-norm rows ready for the LM head
        hipEvent_t* ev_lm = m->chain_events.data() + 4 * cfg.layers + 4;      // per half: the attention stream's share of the LM head done
        int active = b;
        HIP_CHECK(hipMemcpyAsync(&active, st_active, 4, hipMemcpyDeviceToHost, c.s));
        HIP_CHECK(hipStreamSynchronize(c.s));
        if (was_split) {
            // the prefill's step kernel ran over the whole group: the halves start from their slices of the identity
            // list and drop the utterances that stopped on the prefill's token at the first poll (their done flags
            // already keep them out of the attention)
            for (int h = 0; h < 2; ++h) set_rows(hv[h], hv[h].n);
            HIP_CHECK(hipEventRecord(ev_fork, c.s));   // both streams start after everything queued so far
            HIP_CHECK(hipStreamWaitEvent(m->stream_att, ev_fork, 0));
            if (att_streams == 2) HIP_CHECK(hipStreamWaitEvent(m->stream_att2, ev_fork, 0));
            HIP_CHECK(hipStreamWaitEvent(m->stream_mm, ev_fork, 0));
            if (prof_active(c.s)) {
                prof_attach(c.s, m->stream_att);
                if (att_streams == 2) prof_attach(c.s, m->stream_att2);
                prof_attach(c.s, m->stream_mm);
            }
        } else {
            set_rows(hv[0], hv[0].n);
            if (retire && active > 0 && active < b) {   // utterances that stopped on the prefill's token
                hv[0].cur = 1;
                set_rows(hv[0], active);
            }
        }
        int pend_down[2] = {0, 0};  // slabs the last down-projection of each half left to reduce
        int qkv_slabs[2] = {0, 0};  // slabs of the current q/k/v projection (0: bf16 rows in qkvb)
        const char* env_fuse = getenv("Q3ASR_FUSE_ROPE");
        const bool fuse_rope = !(env_fuse && atoi(env_fuse) == 0) && nh / nkv <= 4 && (nh / nkv == 1 || nh / nkv == 2 || nh / nkv == 4);
        // profiling: the attention's algorithmic bytes are only known on the device (which utterances still run), so its
        // launches are recorded without work and the exact figure is added after the loop from the step kernel's key counts
        const size_t per_half = (size_t)max_tokens + 2;
        std::vector<int> attn_launched(2 * per_half, 0), attn_timed(2 * per_half, 0);
        // The step of one half is issued in four kinds of pieces.  Every per-step quantity is read from device arrays,
        // so the launches of all steps between two polls are identical; `it` only feeds the profiler and the logits tap.
        // pre: [token embedding,] input norm and q/k/v projection of layer l
        // Diagnostic (Q3ASR_SPLIT_SERIAL=1): the two CU-confined streams take turns instead of running side by side (every
        // attention launch waits for all projection work issued so far and the other way round), which shows what each
        // kernel reaches on its share of the compute units with the other share idle.
        const char* env_serial = getenv("Q3ASR_SPLIT_SERIAL");
        const bool serial = was_split && env_serial && atoi(env_serial) != 0;
        hipEvent_t ev_ser_mm = nullptr, ev_ser_att = nullptr;
        bool ser_mm_set = false, ser_att_set = false;
        if (serial) {
            HIP_CHECK(hipEventCreateWithFlags(&ev_ser_mm, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&ev_ser_att, hipEventDisableTiming));
        }
        auto ser_before_mm = [&](hipStream_t s) { if (serial && chain && ser_att_set) HIP_CHECK(hipStreamWaitEvent(s, ev_ser_att, 0)); };
        auto ser_after_mm = [&](hipStream_t s) { if (serial && chain) { HIP_CHECK(hipEventRecord(ev_ser_mm, s)); ser_mm_set = true; } };
        auto pre = [&](int h, int l) {
            Half& x = hv[h];
            const DecLayerW& L = m->dec[l];
            const int* map = st_map[x.cur] + x.r0;
            ser_before_mm(x.s);
            if (l == 0)
                launch_embed_tokens(st_next, map, x.rows, H, qemb ? nullptr : (const bf16_t*)m->embed.w,
                                    qemb ? (const uint32_t*)m->embed.w : nullptr, m->embed.scales, m->embed.biases, cfg.group_size, x.hid, x.s);
            if (x.gemv) {  // input norm inside the q/k/v kernel
                GemvParams g;
                g.w = L.qkv.w; g.scales = L.qkv.scales; g.biases = L.qkv.biases; g.qgroup = cfg.group_size; g.N = QKV; g.K = H; g.M = x.rows; g.hid = x.hid; g.norm_w = L.in_norm; g.eps = cfg.rms_eps;
                g.out = x.qkvb; g.ldo = QKV;
                launch_gemv(g, GEMV_NORM_BF16, x.s);
                qkv_slabs[h] = 0;
            } else {
                // residual += previous layer's down-projection partials (if it left any), then input norm
                launch_rmsnorm(x.hid, L.in_norm, x.xn, x.rows, H, cfg.rms_eps, nullptr, x.slab, l ? pend_down[h] : 0, (int64_t)x.rows * H, x.s);
                qkv_slabs[h] = proj(x, x.xn, H, L.qkv, 0, OUT_BF16, x.qkvb, QKV);
            }
            if (!fuse_rope) {
                const int ns = qkv_slabs[h];
                RopeParams rp{ns ? nullptr : x.qkvb, QKV, ns ? x.slab : nullptr, ns, (int64_t)x.rows * QKV, x.qd, (int64_t)nh * hd, x.rows,
                              nh, nkv, L.q_norm, L.k_norm, cfg.rms_eps, m->rope_cos, m->rope_sin, half, cfg.rope_interleaved,
                              m->rope_max_pos, map, nullptr, st_len, kc + l * layer_stride, vc + l * layer_stride, utt_stride, head_stride};
                launch_rope(rp, x.s);
            }
            if (chain) HIP_CHECK(hipEventRecord(ev_pre[h][l], x.s));
            ser_after_mm(x.s);
        };
        // Profiling with two attention streams: a launch whose duration is sampled runs alone on the attention's CUs (it
        // starts after the other half's latest launch has ended, and that half's next launch starts after it), so that
        // the sample says what the kernel reaches by itself; the unsampled launches overlap as they do without a profile.
        hipEvent_t last_attn_ev[2] = {nullptr, nullptr}, alone_ev = nullptr;
        int alone_half = -1;
        auto attn = [&](int h, int l, int it) {
            Half& x = hv[h];
            const DecLayerW& L = m->dec[l];
            if (chain) HIP_CHECK(hipStreamWaitEvent(x.sa, ev_pre[h][l], 0));
            if (serial && chain && ser_mm_set) HIP_CHECK(hipStreamWaitEvent(x.sa, ev_ser_mm, 0));
            if (alone_ev && alone_half != h) {
                HIP_CHECK(hipStreamWaitEvent(x.sa, alone_ev, 0));
                alone_ev = nullptr;
            }
            // two attention streams: a step's first launch of the second half starts behind the first half's, which sets
            // the alternation the later layers keep by themselves (each half's next launch waits for its own projections)
            if (chain && att_streams == 2 && h == 1 && l == 0 && hv[0].rows > 0) HIP_CHECK(hipStreamWaitEvent(x.sa, ev_attn[0][0], 0));
            if (prof_active(x.sa)) {
                const bool timed = prof_next_timed("decode_attn_kernel", x.sa);
                attn_launched[h * per_half + it] += 1;
                attn_timed[h * per_half + it] += timed ? 1 : 0;
                if (timed && chain && att_streams == 2) {
                    if (last_attn_ev[h ^ 1]) HIP_CHECK(hipStreamWaitEvent(x.sa, last_attn_ev[h ^ 1], 0));
                    alone_half = h;
                    alone_ev = ev_attn[h][l];
                }
                // a timed launch's start event would otherwise be stamped while the stream still waits for the q/k/v rows
                if (timed && chain) launch_noop(x.sa);
            }
            DecodeAttnParams dp{x.qd, (int64_t)nh * hd, kc + l * layer_stride, vc + l * layer_stride, utt_stride, head_stride,
                                st_len, fuse_rope ? 0 : 1, x.att, (int64_t)nh * hd, x.rows, nkv, nh / nkv, scale};
            dp.slot = st_map[x.cur] + x.r0;
            dp.done = skip_done ? st_done : nullptr;
            if (fuse_rope) {  // q/k norm, RoPE and the cache append happen inside the attention kernel
                const int ns = qkv_slabs[h];
                dp.rope = DecodeRopeIn{ns ? nullptr : x.qkvb, ns ? x.slab : nullptr, ns, (int64_t)x.rows * QKV, QKV, L.q_norm, L.k_norm,
                                       cfg.rms_eps, m->rope_cos, m->rope_sin, half, cfg.rope_interleaved, m->rope_max_pos};
            }
            launch_decode_attn(dp, hd, x.sa, 0.0);
            if (chain) {
                HIP_CHECK(hipEventRecord(ev_attn[h][l], x.sa));
                last_attn_ev[h] = ev_attn[h][l];
                if (serial) { HIP_CHECK(hipEventRecord(ev_ser_att, x.sa)); ser_att_set = true; }
            }
        };
        // post: o (+= residual), post norm + gate/up + SwiGLU, down (+= residual)
        auto post = [&](int h, int l) {
            Half& x = hv[h];
            const DecLayerW& L = m->dec[l];
            if (chain) HIP_CHECK(hipStreamWaitEvent(x.s, ev_attn[h][l], 0));
            ser_before_mm(x.s);
            if (x.gemv) {
                GemvParams g;
                g.M = x.rows; g.eps = cfg.rms_eps; g.qgroup = cfg.group_size;
                g.w = L.o.w; g.scales = L.o.scales; g.biases = L.o.biases; g.N = H; g.K = nh * hd; g.a = x.att; g.lda = (int64_t)nh * hd; g.out = x.hid; g.ldo = H;
                launch_gemv(g, GEMV_ACC_F32, x.s);
                g.w = L.gate_up.w; g.scales = L.gate_up.scales; g.biases = L.gate_up.biases; g.N = 2 * cfg.inter; g.K = H; g.hid = x.hid; g.norm_w = L.post_norm; g.out = x.act; g.ldo = cfg.inter;
                launch_gemv(g, GEMV_NORM_SWIGLU, x.s);
                g.w = L.down.w; g.scales = L.down.scales; g.biases = L.down.biases; g.N = H; g.K = cfg.inter; g.a = x.act; g.lda = cfg.inter; g.out = x.hid; g.ldo = H;
                launch_gemv(g, GEMV_ACC_F32, x.s);
                pend_down[h] = 0;
                return;
            }
            const int ns_o = proj(x, x.att, (int64_t)nh * hd, L.o, 1, OUT_F32_ACC, x.hid, H);
            launch_rmsnorm(x.hid, L.post_norm, x.xn, x.rows, H, cfg.rms_eps, nullptr, x.slab, ns_o, (int64_t)x.rows * H, x.s);
            const int ns_gu = proj(x, x.xn, H, L.gate_up, 2, OUT_SWIGLU, x.act, cfg.inter);
            if (ns_gu) launch_swiglu_reduce(x.slab, ns_gu, (int64_t)x.rows * 2 * cfg.inter, x.rows, 2 * cfg.inter, x.act, x.s);
            pend_down[h] = proj(x, x.act, cfg.inter, L.down, 3, OUT_F32_ACC, x.hid, H);
            ser_after_mm(x.s);
        };
        // tail: final norm, LM head, bookkeeping
        auto tail = [&](int h, int it) {
            Half& x = hv[h];
            launch_rmsnorm(x.hid, m->final_norm, x.x_last, x.rows, H, cfg.rms_eps, nullptr, x.slab, pend_down[h], (int64_t)x.rows * H, x.s);
            lm_head(x.x_last, x.r0, x.rows, it, x.s);
            step(1, x.r0, x.rows, x.active_dev, st_map[x.cur] + x.r0, st_map[x.cur ^ 1] + x.r0, x.keys + it + 1, x.s);
            res->decode_rows += x.rows;
        };
        // The LM head is the one long kernel of a step (vocabulary x hidden against every row), and with two halves it is
        // the stretch during which the attention stream runs dry: a half's next attention needs that half's next token.
        // So at the end of a step both streams share it: the attention stream's CUs take the first `ta` 256-column tiles
        // of the vocabulary for both halves, the other stream the rest, each writing its own range of the per-row
        // (max, index) partials; one reduction per half then picks the token (lowest index on ties, as before: every
        // logit is the same sum whichever stream computes it).  `ta` balances the two streams in tile rounds, the
        // attention stream starting `lm_lead` rounds early (it idles while the last layer's `post` runs).
        const char* env_lms = getenv("Q3ASR_LM_SPLIT");
        const char* env_lead = getenv("Q3ASR_LM_SPLIT_LEAD");
        // Q3ASR_LM_SPLIT=0: every half's LM head whole on the projections' stream; 2: shared also where the 256x256 tile
        // would not be the LM head's own choice (small test groups)
        const bool lm_split_on = !(env_lms && atoi(env_lms) == 0) && was_split && wf == W_BF16 && !logits;
        const bool lm_split_any = env_lms && atoi(env_lms) == 2;
        const int lm_lead = env_lead ? atoi(env_lead) : 7;
        auto lm_split_ok = [&]() {
            return lm_split_on && chain && n_half == 2 && hv[0].rows > 0 && hv[1].rows > 0 && !hv[0].gemv && !hv[1].gemv &&
                   (lm_split_any || (head_tile_for(hv[0].rows) == 3 && head_tile_for(hv[1].rows) == 3));
        };
        auto lm_split_tiles = [&]() {
            const int tn = cdiv(cfg.vocab, 256), ga = std::max(8, att_cus & ~7), gb = std::max(8, mm_cus & ~7);
            const int t0 = cdiv(hv[0].rows, 256), t1 = cdiv(hv[1].rows, 256);
            int best = 0;
            long best_cost = -1;
            for (int ta = 0; ta < tn; ++ta) {   // the other stream keeps at least the last (partial) tile
                const long ca = cdiv(t0 * ta, ga) + cdiv(t1 * ta, ga);
                const long cb = lm_lead + cdiv(t0 * (tn - ta), gb) + cdiv(t1 * (tn - ta), gb);
                const long cost = std::max(ca, cb);
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ta; }
            }
            return best;
        };
        const int lm_nparts = gemm_argmax_parts(cfg.vocab, 3);
        auto lm_part = [&](int h, int col0, int ncols, hipStream_t s, int cus) {
            Half& x = hv[h];
            GemmParams p = gp(c, x.x_last, H, m->embed, x.rows);
            p.W = (const bf16_t*)m->embed.w + (size_t)col0 * H;
            p.N = ncols; p.col_base = col0; p.part_base = col0 / 256 * 4; p.cus = cus;
            p.part_val = part_val + (size_t)x.r0 * max_parts; p.part_idx = part_idx + (size_t)x.r0 * max_parts; p.nparts = lm_nparts;
            launch_gemm(p, A_PLAIN, OUT_ARGMAX, wf, 3, s);
        };
        // last layer's post, final norm, shared LM head, bookkeeping and the next step's first `pre` of both halves
        auto tail_split = [&](int it, bool more) {
            const int l = cfg.layers - 1;
            const int ta = lm_split_tiles();
            for (int h = 0; h < 2; ++h) {
                Half& x = hv[h];
                post(h, l);
                launch_rmsnorm(x.hid, m->final_norm, x.x_last, x.rows, H, cfg.rms_eps, nullptr, x.slab, pend_down[h], (int64_t)x.rows * H, x.s);
                if (ta > 0) HIP_CHECK(hipEventRecord(ev_xlast[h], x.s));
            }
            if (ta > 0)
                for (int h = 0; h < 2; ++h) {
                    hipStream_t sl = hv[0].sa;   // both halves' shares on one stream: two persistent grids would contend for the same CUs
                    HIP_CHECK(hipStreamWaitEvent(sl, ev_xlast[h], 0));
                    if (prof_active(sl)) launch_noop(sl);   // keeps a timed launch's start event behind the wait
                    lm_part(h, 0, ta * 256, sl, att_cus);
                    HIP_CHECK(hipEventRecord(ev_lm[h], sl));
                }
            for (int h = 0; h < 2; ++h) lm_part(h, ta * 256, cfg.vocab - ta * 256, hv[h].s, mm_cus);
            for (int h = 0; h < 2; ++h) {
                Half& x = hv[h];
                if (ta > 0) HIP_CHECK(hipStreamWaitEvent(x.s, ev_lm[h], 0));
                launch_argmax_reduce(part_val + (size_t)x.r0 * max_parts, part_idx + (size_t)x.r0 * max_parts, lm_nparts, x.rows, st_argmax + x.r0, x.s);
                step(1, x.r0, x.rows, x.active_dev, st_map[x.cur] + x.r0, st_map[x.cur ^ 1] + x.r0, x.keys + it + 1, x.s);
                res->decode_rows += x.rows;
                x.pre_issued = more;
                if (more) pre(h, 0);
            }
        };
        // With two halves the attention stream runs A(0,l) A(1,l) A(0,l+1) ... and the other stream post(0,l) pre(0,l+1)
        // post(1,l) pre(1,l+1) ...: while one half's attention streams the cache, the other half's projections run.
        // A half's first `pre` of the next step follows its bookkeeping at once (`more`: another step will be issued),
        // so that the next step's first attention does not wait for the other half's LM head.
        auto issue_step = [&](int it, bool more) {
            for (int h = 0; h < n_half; ++h)
                if (hv[h].rows > 0 && !hv[h].pre_issued) pre(h, 0);
            for (int l = 0; l < cfg.layers; ++l) {
                for (int h = 0; h < n_half; ++h)
                    if (hv[h].rows > 0) attn(h, l, it);
                if (l + 1 == cfg.layers && lm_split_ok()) {
                    tail_split(it, more);
                    break;
                }
                for (int h = 0; h < n_half; ++h) {
                    if (hv[h].rows == 0) continue;
                    post(h, l);
                    if (l + 1 < cfg.layers) {
                        pre(h, l + 1);
                    } else {
                        tail(h, it);
                        hv[h].pre_issued = n_half == 2 && more;
                        if (hv[h].pre_issued) pre(h, 0);
                    }
                }
            }
        };
        // (Dealing one head's keys to several workgroups with a last-arriver merge was measured at one utterance: no gain,
        // the kernel's fixed prologue and the merge's fences cost what the shorter key loop saves.)
        // (Replaying the step as a HIP graph was measured at 1, 8, 32 and 64 utterances: no change -- even one
        // utterance is bound by streaming the 3.4 GB of weights per step, not by the host's launch rate.)
        const char* env_merge = getenv("Q3ASR_DECODE_MERGE_ROWS");
        const int merge_rows = std::min(env_merge ? atoi(env_merge) : SPLIT_MERGE_ROWS, hv[0].n);
        Q3_TRACE("decode: %d utterances, %d halves, attention CUs %d, shared LM head %d", b, n_half, att_cus, (int)lm_split_ok());
        for (int it = 1; it < max_tokens && active > 0; ++it) {
            const bool poll = it % POLL_STEPS == 0 || it + 1 == max_tokens;
            issue_step(it, it + 1 < max_tokens);
            res->decode_steps += 1;
            if (!poll) continue;
            Q3_TRACE("step %d issued, waiting (rows %d + %d)", it, hv[0].rows, n_half == 2 ? hv[1].rows : 0);
            int act_h[2] = {0, 0};
            for (int h = 0; h < n_half; ++h)
                if (hv[h].rows > 0) HIP_CHECK(hipMemcpyAsync(&act_h[h], hv[h].active_dev, 4, hipMemcpyDeviceToHost, hv[h].s));
            for (int h = 0; h < n_half; ++h) HIP_CHECK(hipStreamSynchronize(hv[h].s));   // (the attention stream has drained before the bookkeeping ran)
            active = act_h[0] + act_h[1];
            Q3_TRACE("step %d done, %d utterances running", it, active);
            if (!retire || active == 0) continue;
            for (int h = 0; h < n_half; ++h) {
                Half& x = hv[h];
                if (x.rows == 0 || act_h[h] == x.rows) continue;
                // the step just finished wrote the running utterances into the other list: decode those rows from now on
                x.cur ^= 1;
                if (act_h[h] > 0) set_rows(x, act_h[h]);
                else x.rows = 0;
            }
            if (n_half == 2 && (active <= merge_rows || hv[0].rows == 0 || hv[1].rows == 0)) {
                // Too few rows for two halves to pay (or one half has finished): one list, one stream, every CU again.
                // Both streams are idle here; the lists are joined through a staging array (source and destination
                // ranges of one list array may overlap).
                int o = 0;
                for (int h = 0; h < 2; ++h)
                    if (hv[h].rows > 0) {
                        HIP_CHECK(hipMemcpyAsync(st_tmp + o, st_map[hv[h].cur] + hv[h].r0, (size_t)hv[h].rows * 4, hipMemcpyDeviceToDevice, c.s));
                        o += hv[h].rows;
                    }
                HIP_CHECK(hipMemcpyAsync(st_map[0], st_tmp, (size_t)o * 4, hipMemcpyDeviceToDevice, c.s));
                n_half = 1;
                chain = false;
                cur_cus = 0;
                Half& x = hv[0];
                x.r0 = 0; x.cur = 0; x.s = x.sa = c.s; x.x_last = x_last;
                pend_down[0] = 0;
                set_rows(x, o);
            }
        }
        if (was_split && n_half == 2) {  // everything after the loop runs on the main stream
            HIP_CHECK(hipEventRecord(ev_join, m->stream_mm));
            HIP_CHECK(hipStreamWaitEvent(c.s, ev_join, 0));
            HIP_CHECK(hipEventRecord(ev_join, m->stream_att));
            HIP_CHECK(hipStreamWaitEvent(c.s, ev_join, 0));
            if (att_streams == 2) {
                HIP_CHECK(hipEventRecord(ev_join, m->stream_att2));
                HIP_CHECK(hipStreamWaitEvent(c.s, ev_join, 0));
            }
        }
        cur_cus = 0;
        if (serial) {
            HIP_CHECK(hipStreamSynchronize(c.s));
            (void)hipEventDestroy(ev_ser_mm);
            (void)hipEventDestroy(ev_ser_att);
        }
        if (prof_active(c.s)) {
            std::vector<long long> keys(2 * per_half);
            HIP_CHECK(hipMemcpyAsync(keys.data(), st_keys, keys.size() * sizeof(long long), hipMemcpyDeviceToHost, c.s));
            HIP_CHECK(hipStreamSynchronize(c.s));
            double total = 0, timed = 0;
            for (int h = 0; h < 2; ++h)
                for (int it = 1; it < max_tokens; ++it) {
                    const double bytes = 2.0 * 2.0 * nkv * hd * (double)keys[h * per_half + it];   // K and V rows, bf16, every kv head
                    total += bytes * attn_launched[h * per_half + it];
                    timed += bytes * attn_timed[h * per_half + it];
                }
            prof_add_work("decode_attn_kernel", total, timed, c.s);
        }
    }
    HIP_CHECK(hipEventRecord(ev.e[4], c.s));

    // ---- results ----
    std::vector<int> h_tokens((size_t)b * max_tokens), h_ntok(b);
    HIP_CHECK(hipMemcpyAsync(h_tokens.data(), st_tokens, h_tokens.size() * 4, hipMemcpyDeviceToHost, c.s));
    HIP_CHECK(hipMemcpyAsync(h_ntok.data(), st_ntok, (size_t)b * 4, hipMemcpyDeviceToHost, c.s));
    HIP_CHECK(hipStreamSynchronize(c.s));
    for (int i = 0; i < b; ++i) {
        // tokens appended after an utterance stopped are not part of its transcript
        res->n_tokens[idx[i]] = h_ntok[i];
        memcpy(res->tokens + (size_t)idx[i] * max_tokens, &h_tokens[(size_t)i * max_tokens], (size_t)h_ntok[i] * 4);
        res->tokens_total += h_ntok[i];
    }
    float ms;
    HIP_CHECK(hipEventElapsedTime(&ms, ev.e[0], ev.e[1])); res->ms_mel += ms;
    HIP_CHECK(hipEventElapsedTime(&ms, ev.e[1], ev.e[2])); res->ms_encoder += ms;
    HIP_CHECK(hipEventElapsedTime(&ms, ev.e[2], ev.e[3])); res->ms_prefill += ms;
    HIP_CHECK(hipEventElapsedTime(&ms, ev.e[3], ev.e[4])); res->ms_decode += ms;
    c.keep.clear();
}

}  // namespace

void model_transcribe(q3asr_model* m, const q3asr_request* req, q3asr_result* res) {
    Q3_REQUIRE(m->finalized, "transcribe: weights not finalized");
    Q3_REQUIRE(req && res && req->audio && req->offsets && res->tokens && res->n_tokens, "transcribe: null argument");
    Q3_REQUIRE(req->n_utt > 0, "transcribe: empty batch");
    Q3_REQUIRE(req->max_tokens >= 1 && req->max_tokens <= 2048, "transcribe: max_tokens must be in [1, 2048]");
    Q3_REQUIRE(req->sample_rate > 0, "transcribe: sample rate must be positive");
    HIP_CHECK(hipSetDevice(m->device));
    const q3asr_config& cfg = m->cfg;
    const int n = req->n_utt;
    res->audio_seconds = res->ms_mel = res->ms_encoder = res->ms_prefill = res->ms_decode = 0;
    res->decode_steps = res->tokens_total = res->decode_rows = 0;
    // per-utterance sizes in request order (tap layouts) and the length-sorted processing order
    std::vector<int> frames(n), order(n);
    std::vector<int64_t> tap_a_off(n + 1, 0), tap_s_off(n + 1, 0);
    const double ratio = 16000.0 / (double)req->sample_rate;
    for (int u = 0; u < n; ++u) {
        const int64_t len = req->offsets[u + 1] - req->offsets[u];
        Q3_REQUIRE(len >= 0, "transcribe: offsets must be non-decreasing");
        const int64_t n16 = req->sample_rate == 16000 ? len : (int64_t)((double)len * ratio);
        if (n16 < MEL_HOP) throw Q3Error("q3asr: transcribe: utterance " + std::to_string(u) + " is shorter than one 10 ms frame");
        frames[u] = mel_frames_out(n16);
        const int na = model_output_length(cfg, frames[u]);
        int nl = 0;
        if (req->lang_ids && req->lang_offsets) {
            nl = req->lang_offsets[u + 1] - req->lang_offsets[u];
            bool has = false;
            for (int k = req->lang_offsets[u]; k < req->lang_offsets[u + 1]; ++k) has |= req->lang_ids[k] == TOK_ASR_TEXT;
            if (nl > 0 && !has) nl += 1;
        }
        tap_a_off[u + 1] = tap_a_off[u] + na;
        tap_s_off[u + 1] = tap_s_off[u] + na + 15 + nl;
    }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return frames[a] > frames[b]; });
    // (Full groups and a rest: cutting a request a little above max_batch into even groups instead was measured on 2840
    // pieces of ragged recordings, 2 x 1420 against 2048 + 792, and is slower, 3524 x against 3579 x real time: a group
    // decodes the faster per utterance the larger it is.)
    const int mb = std::max(1, cfg.max_batch);
    for (int g0 = 0; g0 < n; g0 += mb) {
        std::vector<int> idx(order.begin() + g0, order.begin() + std::min(n, g0 + mb));
        m->arena.reset();
        Ctx c{m, m->stream, {}};
        transcribe_group(c, req, res, idx, tap_a_off, tap_s_off);
    }
}

int model_decode_att_cus(const q3asr_model* m, int n_utt) { return decode_att_cus(m, n_utt); }

void model_release_process_streams() {
    SplitStreamCache& cache = split_stream_cache();
    std::lock_guard<std::mutex> lk(cache.mu);
    for (auto& kv : cache.by_share) {
        (void)hipSetDevice(kv.first.first);
        for (hipStream_t s : {kv.second.att, kv.second.att2, kv.second.mm})
            if (s) (void)hipStreamDestroy(s);
    }
    cache.by_share.clear();
}
