// extern "C" surface of libq3asr (declared in include/q3asr.h).
#include "model.h"
#include "prof.h"
#include <cstdlib>
#include <cstring>
#include <set>

static thread_local std::string g_last_error;
void q3_set_error(const std::string& msg) { g_last_error = msg; }

static thread_local Q3Knobs g_knobs;
Q3Knobs& q3_knobs() { return g_knobs; }
void q3_knobs_refresh() {
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
    g_knobs.flash_pair = num("Q3ASR_FLASH_PAIR", -1);
    g_knobs.attn_waves = num("Q3ASR_ATTN_WAVES", 0);
    g_knobs.force_tile = num("Q3ASR_FORCE_TILE", -1);
}

#define Q3_API_BEGIN try { q3_knobs_refresh();
#define Q3_API_END                         \
    g_last_error.clear();                  \
    return 0;                              \
    }                                      \
    catch (const std::exception& e) {      \
        g_last_error = e.what();           \
        return 1;                          \
    }                                      \
    catch (...) {                          \
        g_last_error = "q3asr: unknown error"; \
        return 2;                          \
    }

// Models that exist.  A host may end without calling q3asr_destroy (a Swift deinit is not guaranteed at process exit,
// nor is a Python __del__), which would leave streams, CU-masked queues, events and device memory for the HIP
// runtime's own exit handlers to find alive; under a profiler that ended in a segmentation fault inside the runtime's
// static destruction.  So the library destroys what is left itself, from an exit handler registered right after the
// first successful HIP call of the first q3asr_create: handlers run in reverse order of registration, i.e. this one
// before anything the runtime registered while it initialised.  The registry and its mutex are never destructed
// (no static destructor of this library calls HIP), and q3asr_destroy ignores a handle that is not in the registry
// (destroyed twice, or destroyed by the exit handler before a late finaliser of the host got to it).
namespace {
struct LiveModels {
    std::mutex mu;
    std::set<q3asr_model*> set;
};
LiveModels& live_models() {
    static LiveModels* p = new LiveModels();
    return *p;
}
void destroy_model(q3asr_model* m) {
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    try { (void)prof_stop(m->stream); } catch (...) {}   // a profile left running must not outlive its streams
    mel_tables_destroy(m->mel);
    m->arena.release();
    for (void* p : m->owned) (void)hipFree(p);
    for (hipEvent_t e : m->chain_events) (void)hipEventDestroy(e);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}
void destroy_live_models_at_exit() {
    std::set<q3asr_model*> left;
    {
        std::lock_guard<std::mutex> lk(live_models().mu);
        left.swap(live_models().set);
    }
    bool busy = false;
    for (q3asr_model* m : left) {
        if (!m->call_mutex.try_lock()) { busy = true; continue; }   // a call still runs on another thread: leave that model alone
        m->call_mutex.unlock();
        destroy_model(m);
    }
    if (!busy) model_release_process_streams();
}
}  // namespace

extern "C" {

const char* q3asr_last_error(void) { return g_last_error.c_str(); }

int q3asr_create(const q3asr_config* cfg, int device, q3asr_model** out) {
    Q3_API_BEGIN
    Q3_REQUIRE(cfg && out, "create: null argument");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) throw Q3Error("q3asr: no HIP device available (this library has no CPU path)");
    Q3_REQUIRE(device >= 0 && device < ndev, "create: device ordinal out of range");
    Q3_REQUIRE(cfg->n_mels == 128 && cfg->n_window == 50, "create: kernels are built for 128 mel bins and 100-frame chunks");
    Q3_REQUIRE(cfg->head_dim == 128, "create: decoder head_dim must be 128");
    Q3_REQUIRE(cfg->enc_d_model % cfg->enc_heads == 0 && cfg->enc_d_model / cfg->enc_heads == 64, "create: encoder head_dim must be 64");
    Q3_REQUIRE(cfg->heads % cfg->kv_heads == 0, "create: heads must be a multiple of kv_heads");
    Q3_REQUIRE(cfg->enc_d_model % 32 == 0 && cfg->enc_ffn % 32 == 0 && cfg->hidden % 64 == 0 && cfg->inter % 64 == 0 &&
                   cfg->conv_channels % 32 == 0 && cfg->enc_out_dim % 4 == 0,
               "create: layer widths must be multiples of 32 (64 for the decoder)");
    Q3_REQUIRE(cfg->enc_d_model <= 2048 && cfg->hidden <= 2048, "create: model width above 2048 is not supported by the norm kernels");
    Q3_REQUIRE(cfg->rotary_dim > 0 && cfg->rotary_dim % 4 == 0 && cfg->rotary_dim <= cfg->head_dim, "create: rotary_dim must be a multiple of 4 within head_dim");
    Q3_REQUIRE(cfg->weight_format == 0 || cfg->weight_format == 1, "create: unknown weight_format");
    Q3_REQUIRE(cfg->weight_format == 0 || (cfg->group_size == 64 && cfg->bits == 4),
               "create: quantised weights must be MLX affine 4-bit with group size 64 (Configuration.swift:67-68)");
    Q3_REQUIRE(cfg->n_window_infer > 0 && cfg->n_window_infer % (2 * cfg->n_window) == 0, "create: n_window_infer must be a multiple of the chunk length");
    Q3_REQUIRE(cfg->conv_channels > 0 && cfg->enc_layers > 0 && cfg->layers > 0 && cfg->vocab > 0 && cfg->max_batch >= 0, "create: non-positive size");
    HIP_CHECK(hipSetDevice(device));
    static std::once_flag exit_hook;
    // (Q3ASR_NO_EXIT_HOOK=1 leaves the handler out: a diagnostic switch for showing what it is there to prevent)
    std::call_once(exit_hook, [] { if (!getenv("Q3ASR_NO_EXIT_HOOK")) atexit(destroy_live_models_at_exit); });
    q3asr_model* m = new q3asr_model();
    m->cfg = *cfg;
    m->device = device;
    if (const char* mb = getenv("Q3ASR_ARENA_BLOCK_MB"))
        if (atoi(mb) > 0) m->arena.DEFAULT_BLOCK = (size_t)atoi(mb) << 20;
    try {
        HIP_CHECK(hipStreamCreate(&m->stream));
        m->zero_page = m->dalloc<bf16_t>(512);
        HIP_CHECK(hipMemset(m->zero_page, 0, 1024));
        mel_tables_create(m->mel);
        mel_noalloc_init();
    } catch (...) {
        destroy_model(m);
        throw;
    }
    {
        std::lock_guard<std::mutex> lk(live_models().mu);
        live_models().set.insert(m);
    }
    *out = m;
    Q3_API_END
}

void q3asr_destroy(q3asr_model* m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> lk(live_models().mu);
        if (live_models().set.erase(m) == 0) return;   // not (or no longer) a live model
    }
    { std::lock_guard<std::mutex> wait_for_a_running_call(m->call_mutex); }
    destroy_model(m);
}

int q3asr_load_tensor(q3asr_model* m, const char* name, int dtype, const int64_t* shape, int ndim, const void* data) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && name && shape && data, "load_tensor: null argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    Q3_REQUIRE(!m->finalized, "load_tensor: model already finalized");
    Q3_REQUIRE(ndim >= 1 && ndim <= 4, "load_tensor: rank must be 1..4");
    Q3_REQUIRE(dtype == Q3ASR_F32 || dtype == Q3ASR_BF16 || dtype == Q3ASR_U32, "load_tensor: unknown dtype");
    HostTensor t;
    t.dtype = dtype;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        Q3_REQUIRE(shape[i] > 0, "load_tensor: non-positive dimension");
        t.shape.push_back(shape[i]);
        n *= shape[i];
    }
    const size_t bytes = (size_t)n * (dtype == Q3ASR_BF16 ? 2 : 4);
    t.data.assign((const uint8_t*)data, (const uint8_t*)data + bytes);
    m->host[name] = std::move(t);
    Q3_API_END
}

int q3asr_finalize(q3asr_model* m) {
    Q3_API_BEGIN
    Q3_REQUIRE(m, "finalize: null model");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    model_finalize(m);
    Q3_API_END
}

int q3asr_mel_filterbank(q3asr_model* m, float* out) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && out, "mel_filterbank: null argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    memcpy(out, m->mel.host_fb.data(), m->mel.host_fb.size() * 4);
    Q3_API_END
}

int q3asr_mel(q3asr_model* m, const float* audio, const int64_t* offsets, int n_utt, int sample_rate, int audio_space,
              float* mel_out, int t_stride, int32_t* n_frames, int out_space) {
    Q3_API_BEGIN
    Q3_REQUIRE(m, "mel: null model");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    model_mel(m, audio, offsets, n_utt, sample_rate, audio_space, mel_out, t_stride, n_frames, out_space);
    Q3_API_END
}

int q3asr_mel_noalloc(q3asr_model* m, const float* audio_dev, const int64_t* offsets, int n_utt, float* mel_out_dev, int t_stride,
                      int32_t* n_frames, void* stream) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && audio_dev && offsets && mel_out_dev && n_frames, "mel_noalloc: null argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    HIP_CHECK(hipSetDevice(m->device));
    hipStream_t s = stream ? (hipStream_t)stream : m->stream;
    mel_noalloc_run(m->mel, audio_dev, offsets, n_utt, mel_out_dev, t_stride, n_frames, s);
    if (!stream) HIP_CHECK(hipStreamSynchronize(s));
    Q3_API_END
}

int q3asr_encode(q3asr_model* m, const float* mel, int t_stride, const int32_t* n_frames, int n_utt, int mel_space,
                 float* embeds_out, int64_t out_rows_capacity, int32_t* cu_tokens, int out_space) {
    Q3_API_BEGIN
    Q3_REQUIRE(m, "encode: null model");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    model_encode(m, mel, t_stride, n_frames, n_utt, mel_space, embeds_out, out_rows_capacity, cu_tokens, out_space);
    Q3_API_END
}

int q3asr_output_length(const q3asr_model* m, int n_frames) { return m ? model_output_length(m->cfg, n_frames) : -1; }

int64_t q3asr_workspace_bytes(q3asr_model* m) {
    if (!m) return -1;
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    return (int64_t)m->arena.footprint();
}

int q3asr_decode_attention_cus(q3asr_model* m, int n_utt) {
    if (!m) return -1;
    try {
        std::lock_guard<std::mutex> call_lock(m->call_mutex);
        HIP_CHECK(hipSetDevice(m->device));
        return model_decode_att_cus(m, n_utt);
    } catch (...) {
        return -1;
    }
}

int q3asr_transcribe_tokens(q3asr_model* m, const q3asr_request* req, q3asr_result* res) {
    Q3_API_BEGIN
    Q3_REQUIRE(m, "transcribe: null model");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    model_transcribe(m, req, res);
    Q3_API_END
}

int q3asr_profile_start(q3asr_model* m, int stride) {
    Q3_API_BEGIN
    Q3_REQUIRE(m, "profile_start: null model");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    prof_start(m->stream, stride);
    Q3_API_END
}

int q3asr_profile_stop(q3asr_model* m, char* json_out, int capacity) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && json_out && capacity > 2, "profile_stop: null argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    HIP_CHECK(hipSetDevice(m->device));
    const std::string j = prof_stop(m->stream);
    Q3_REQUIRE((int)j.size() + 1 <= capacity, "profile_stop: buffer too small");
    memcpy(json_out, j.c_str(), j.size() + 1);
    Q3_API_END
}

// ---- kernel-level entry points for the parity tests ----
int q3asr_op_gemm(q3asr_model* m, const uint16_t* a, const uint16_t* w, const float* bias, int M, int N, int K, int gelu,
                  int tile, int ksplit, float* out) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && a && w && out, "op_gemm: null argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    bf16_t* da = m->arena.get<bf16_t>((size_t)M * K);
    bf16_t* dw = m->arena.get<bf16_t>((size_t)N * K);
    float* db = bias ? m->arena.get<float>(N) : nullptr;
    float* dout = m->arena.get<float>((size_t)M * N * std::max(1, ksplit));
    HIP_CHECK(hipMemcpyAsync(da, a, (size_t)M * K * 2, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(dw, w, (size_t)N * K * 2, hipMemcpyHostToDevice, m->stream));
    if (bias) HIP_CHECK(hipMemcpyAsync(db, bias, (size_t)N * 4, hipMemcpyHostToDevice, m->stream));
    GemmParams p;
    p.A = da; p.lda = K; p.W = dw; p.M = M; p.N = N; p.K = K; p.zero = m->zero_page;
    p.out = dout; p.ldo = N;
    if (ksplit > 1) {
        Q3_REQUIRE(!bias && !gelu, "op_gemm: split-K has no bias / activation");
        p.ksplit = ksplit; p.slab_stride = (int64_t)M * N;
        launch_gemm(p, A_PLAIN, OUT_SLAB, W_BF16, tile, m->stream);
        std::vector<float> h((size_t)M * N * ksplit);
        HIP_CHECK(hipMemcpyAsync(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost, m->stream));
        HIP_CHECK(hipStreamSynchronize(m->stream));
        for (size_t i = 0; i < (size_t)M * N; ++i) {
            float s = 0.f;
            for (int z = 0; z < ksplit; ++z) s += h[(size_t)z * M * N + i];
            out[i] = s;
        }
    } else {
        p.bias = db; p.act = gelu ? 1 : 0;
        launch_gemm(p, A_PLAIN, OUT_F32, W_BF16, tile, m->stream);
        HIP_CHECK(hipMemcpyAsync(out, dout, (size_t)M * N * 4, hipMemcpyDeviceToHost, m->stream));
        HIP_CHECK(hipStreamSynchronize(m->stream));
    }
    Q3_API_END
}

int q3asr_op_greedy_rules(q3asr_model* m, const int32_t* picks, const int32_t* seen0, int n_utt, int steps, int max_tokens,
                          int stop_rules, int32_t* tokens_out, int32_t* n_tokens_out) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && picks && tokens_out && n_tokens_out && n_utt > 0 && steps > 0 && max_tokens > 0, "op_greedy_rules: bad argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    const size_t B = (size_t)n_utt;
    int* d_picks = m->arena.get<int>(B * steps);
    int* d_tokens = m->arena.get<int>(B * max_tokens);
    int* d_state = m->arena.get<int>(B * 5 + 64);  // ntok, done, seen, next, len, n_active
    HIP_CHECK(hipMemcpyAsync(d_picks, picks, B * steps * 4, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemsetAsync(d_tokens, 0, B * max_tokens * 4, m->stream));
    HIP_CHECK(hipMemsetAsync(d_state, 0, (B * 5 + 64) * 4, m->stream));
    if (seen0) HIP_CHECK(hipMemcpyAsync(d_state + 2 * B, seen0, B * 4, hipMemcpyHostToDevice, m->stream));
    // picks are stored step-major so that each step's argmax row is contiguous
    for (int t = 0; t < steps; ++t) {
        StepParams sp{d_picks + (size_t)t * B, d_tokens, d_state, d_state + B, d_state + 2 * B, d_state + 3 * B, d_state + 4 * B,
                      d_state + 5 * B, nullptr, n_utt, max_tokens, t ? 1 : 0, stop_rules};
        launch_step(sp, m->stream);
    }
    HIP_CHECK(hipMemcpyAsync(tokens_out, d_tokens, B * max_tokens * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipMemcpyAsync(n_tokens_out, d_state, B * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    Q3_API_END
}

int q3asr_op_read_bandwidth(q3asr_model* m, int64_t bytes, float* gb_per_s) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && gb_per_s && bytes >= (1 << 20) && bytes % 16 == 0, "op_read_bandwidth: bad argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    char* buf = m->arena.get<char>((size_t)bytes);
    launch_fill_bf16((bf16_t*)buf, (size_t)bytes / 2, 0x77u, m->stream);
    float best = 0.f;
    for (int blocks : {4096, 8192, 16384}) {
        const float ms = read_probe_ms(buf, (size_t)bytes, blocks, 5, m->stream);
        best = std::max(best, (float)((double)bytes / (ms * 1e-3) / 1e9));
    }
    *gb_per_s = best;
    Q3_API_END
}

int q3asr_op_mfma_f32_rate(q3asr_model* m, int waves_per_simd, int iters, float* tflops) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && tflops && iters > 0 && (waves_per_simd == 1 || waves_per_simd == 2), "op_mfma_f32_rate: bad argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    *tflops = mfma_f32_probe_tflops(m->arena.get<float>(4), waves_per_simd, iters, m->stream);
    Q3_API_END
}

int q3asr_op_gemm_time(q3asr_model* m, int M, int N, int K, int tile, int ksplit, int out_kind, int iters, float* ms_per_call) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && ms_per_call && iters > 0 && out_kind >= 0 && out_kind <= 5, "op_gemm_time: bad argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    const int ks = std::max(1, ksplit);
    bf16_t* da = m->arena.get<bf16_t>((size_t)M * K);
    bf16_t* dw = m->arena.get<bf16_t>((size_t)N * K);
    float* dout = m->arena.get<float>((size_t)M * N * ks);
    launch_fill_bf16(da, (size_t)M * K, 0x1234u, m->stream);
    launch_fill_bf16(dw, (size_t)N * K, 0x9876u, m->stream);
    HIP_CHECK(hipMemsetAsync(dout, 0, (size_t)M * N * ks * 4, m->stream));
    GemmParams p;
    p.A = da; p.lda = K; p.W = dw; p.M = M; p.N = N; p.K = K; p.zero = m->zero_page;
    p.out = dout; p.ldo = out_kind == OUT_SWIGLU ? N / 2 : N;
    if (out_kind == OUT_SLAB) { p.ksplit = ks; p.slab_stride = (int64_t)M * N; }
    int use_tile = tile < 0 ? gemm_pick_tile(M, N) : tile;
    if (out_kind == OUT_ARGMAX) {
        p.nparts = gemm_argmax_parts(N, use_tile == 2 ? 1 : use_tile);
        p.part_val = m->arena.get<float>((size_t)M * p.nparts);
        p.part_idx = m->arena.get<int>((size_t)M * p.nparts);
    }
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch_gemm(p, A_PLAIN, (GemmOut)out_kind, W_BF16, use_tile, m->stream);
    HIP_CHECK(hipEventRecord(e0, m->stream));
    for (int i = 0; i < iters; ++i) launch_gemm(p, A_PLAIN, (GemmOut)out_kind, W_BF16, use_tile, m->stream);
    HIP_CHECK(hipEventRecord(e1, m->stream));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_call = ms / iters;
    Q3_API_END
}

int q3asr_op_gemm_q4(q3asr_model* m, const uint16_t* a, const uint32_t* wq, const uint16_t* scales, const uint16_t* biases,
                     int M, int N, int K, int tile, float* out) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && a && wq && scales && biases && out, "op_gemm_q4: null argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    bf16_t* da = m->arena.get<bf16_t>((size_t)M * K);
    uint32_t* dw = m->arena.get<uint32_t>((size_t)N * K / 8);
    bf16_t* ds = m->arena.get<bf16_t>((size_t)N * K / 64);
    bf16_t* db = m->arena.get<bf16_t>((size_t)N * K / 64);
    float* dout = m->arena.get<float>((size_t)M * N);
    HIP_CHECK(hipMemcpyAsync(da, a, (size_t)M * K * 2, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(dw, wq, (size_t)N * K / 8 * 4, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(ds, scales, (size_t)N * K / 64 * 2, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(db, biases, (size_t)N * K / 64 * 2, hipMemcpyHostToDevice, m->stream));
    GemmParams p;
    p.A = da; p.lda = K; p.W = dw; p.qscales = ds; p.qbiases = db; p.qgroup = 64;
    p.M = M; p.N = N; p.K = K; p.zero = m->zero_page; p.out = dout; p.ldo = N;
    launch_gemm(p, A_PLAIN, OUT_F32, W_Q4, tile, m->stream);
    HIP_CHECK(hipMemcpyAsync(out, dout, (size_t)M * N * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    Q3_API_END
}

int q3asr_op_attention(q3asr_model* m, const uint16_t* q, const uint16_t* k, const uint16_t* v, const int32_t* cu_seqlens,
                       int n_seq, int heads, int group, int head_dim, int causal, uint16_t* out) {
    Q3_API_BEGIN
    Q3_REQUIRE(m && q && k && v && cu_seqlens && out && n_seq > 0, "op_attention: null argument");
    std::lock_guard<std::mutex> call_lock(m->call_mutex);
    Q3_REQUIRE(group > 0 && heads % group == 0, "op_attention: heads must be a multiple of group");
    HIP_CHECK(hipSetDevice(m->device));
    m->arena.reset();
    const int rows = cu_seqlens[n_seq], kvh = heads / group;
    const size_t qn = (size_t)rows * heads * head_dim, kn = (size_t)rows * kvh * head_dim;
    bf16_t* dq = m->arena.get<bf16_t>(qn);
    bf16_t* dk = m->arena.get<bf16_t>(kn);
    bf16_t* dv = m->arena.get<bf16_t>(kn);
    bf16_t* dout = m->arena.get<bf16_t>(qn);
    HIP_CHECK(hipMemcpyAsync(dq, q, qn * 2, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(dk, k, kn * 2, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(dv, v, kn * 2, hipMemcpyHostToDevice, m->stream));
    std::vector<AttnSeq> seqs;
    std::vector<AttnTile> tiles;
    for (int s = 0; s < n_seq; ++s) {
        const int r0 = cu_seqlens[s], len = cu_seqlens[s + 1] - r0;
        Q3_REQUIRE(len > 0, "op_attention: empty sequence");
        seqs.push_back({r0, len, (int64_t)r0 * kvh * head_dim});
        for (int q0 = 0; q0 < len; q0 += 64) tiles.push_back({s, q0});
    }
    AttnSeq* dseq = m->arena.get<AttnSeq>(seqs.size());
    AttnTile* dtile = m->arena.get<AttnTile>(tiles.size());
    HIP_CHECK(hipMemcpyAsync(dseq, seqs.data(), seqs.size() * sizeof(AttnSeq), hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipMemcpyAsync(dtile, tiles.data(), tiles.size() * sizeof(AttnTile), hipMemcpyHostToDevice, m->stream));
    FlashParams fp{dq, (int64_t)heads * head_dim, dk, dv, (int64_t)kvh * head_dim, head_dim, dout, (int64_t)heads * head_dim,
                   dseq, dtile, (int)tiles.size(), heads, group, 1.0f / sqrtf((float)head_dim), causal};
    launch_flash_attn(fp, head_dim, m->stream);
    HIP_CHECK(hipMemcpyAsync(out, dout, qn * 2, hipMemcpyDeviceToHost, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));
    Q3_API_END
}

}  // extern "C"
