/* libq3asr -- C ABI of the MI355X (gfx950) Qwen3-ASR offline-transcription hot path.
 *
 * The reference (seichris/qwen3-asr-swift) has no FFI seam for this path: its Swift modules call
 * MLX directly.  This header is the seam a Swift host would bind through a system-library module map: the host
 * keeps Qwen3ASRModel.fromPretrained / transcribe(audio:sampleRate:) and its safetensors loader and
 * calls these entry points.  Every declaration cites the reference code it replaces, as
 * Sources/Qwen3ASR/<file>:<lines>.
 *
 * Conventions: plain C; every function returns 0 on success and a non-zero code on failure, with a
 * thread-local message available from q3asr_last_error().  The caller owns all buffers it passes;
 * the library owns all device memory it allocates.  A model's weights are immutable after q3asr_finalize.  Any
 * number of host threads may call into one model (the reference drives its model from detached tasks,
 * Qwen3ASR.swift:685-687, RealtimeTranslation.swift:459-477): calls on one model are serialised by the library
 * and each runs to completion, like the reference's synchronous transcribe; different models run concurrently.
 * "space" arguments say where a caller buffer lives: Q3ASR_HOST or Q3ASR_DEVICE (HIP device memory
 * of the model's GPU).
 */
#ifndef Q3ASR_H
#define Q3ASR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct q3asr_model q3asr_model;

enum { Q3ASR_HOST = 0, Q3ASR_DEVICE = 1 };
enum { Q3ASR_F32 = 0, Q3ASR_BF16 = 1, Q3ASR_U32 = 2 };

/* Model hyper-parameters: Qwen3AudioEncoderConfig (AudioEncoder.swift:7-43) and TextDecoderConfig
 * (Configuration.swift:46-110) plus the switches that change numerics. */
typedef struct q3asr_config {
    int enc_d_model, enc_heads, enc_ffn, enc_layers, n_mels, enc_out_dim, conv_channels, n_window, n_window_infer;
    float ln_eps;
    int vocab, hidden, layers, heads, kv_heads, head_dim, inter;
    float rms_eps, rope_theta;
    int rope_interleaved, rotary_dim; /* Configuration.swift:62,75-80 */
    int group_size, bits;             /* Configuration.swift:67-68 */
    int weight_format;                /* 0: dense bf16 decoder (TextDecoder.swift), 1: MLX affine q4 (QuantizedTextDecoder.swift) */
    int embed_scale;                  /* Qwen3ASR.swift:190-196, default 1 */
    int max_batch;                    /* utterances decoded together on this GPU */
} q3asr_config;

/* Last error message of the calling thread ("" if none). */
const char* q3asr_last_error(void);

/* Qwen3ASRModel.init (Qwen3ASR.swift:33-42): allocate a model for `cfg` on HIP device `device`. */
int q3asr_create(const q3asr_config* cfg, int device, q3asr_model** out);
void q3asr_destroy(q3asr_model* m);

/* WeightLoader (WeightLoading.swift:92-127, 151-176, 183-347): hand over one checkpoint tensor by
 * its safetensors key ("audio_tower.conv2d1.weight", "model.layers.0.self_attn.q_proj.scales", ...).
 * `data` is a host pointer borrowed for the duration of the call.  Conv weights arrive in the
 * checkpoint's [out, kH, kW, in] layout (WeightLoading.swift:277-284). */
int q3asr_load_tensor(q3asr_model* m, const char* name, int dtype, const int64_t* shape, int ndim, const void* data);
/* Ends loading: re-lays weights out for the kernels and moves them to HBM.  Unlike the reference,
 * which silently skips absent keys (WeightLoading.swift:190-203), a missing or mis-shaped tensor is an error. */
int q3asr_finalize(q3asr_model* m);

/* WhisperFeatureExtractor.setupMelFilterbank (AudioPreprocessing.swift:23-126): the [128][201] fp32 filterbank. */
int q3asr_mel_filterbank(q3asr_model* m, float* out);

/* WhisperFeatureExtractor.process (AudioPreprocessing.swift:330-344; resample :347-365, extractFeatures
 * :131-324), batched.  audio: concatenated mono samples, utterance u = audio[offsets[u] .. offsets[u+1]).
 * mel_out: [n_utt][128][t_stride] fp32, row u/m holds n_frames[u] values (t_stride >= 3000 is always enough). */
int q3asr_mel(q3asr_model* m, const float* audio, const int64_t* offsets, int n_utt, int sample_rate, int audio_space,
              float* mel_out, int t_stride, int32_t* n_frames, int out_space);

/* q3asr_mel for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid it).  No counterpart in the reference.  Same kernels and the same result as q3asr_mel on device buffers, but
 * no hipMalloc / hipFree / host-to-device copy happens inside: what q3asr_mel takes from the model's workspace (offsets on the
 * device, the tile list, per-utterance maxima, frame counts and per-cell minima) lives in a fixed 1.7 MB device array that every GPU holds
 * until the process ends.  Limits of that array: 16 kHz input only, at most 256 utterances and 24 064 tiles of 32 frames
 * (256 clips of 30 s) per call, at most 600 s per utterance; anything larger is rejected with an error.  audio_dev and
 * mel_out_dev ([n_utt][128][t_stride] fp32) are device pointers, offsets ([n_utt + 1]) and n_frames ([n_utt]) host
 * pointers.  stream: a hipStream_t to enqueue on (the call then returns without waiting), or NULL for the model's own
 * stream (the call then waits, like q3asr_mel).  All calls on one GPU share the array, through whichever model or stream
 * they come; the library runs them one after the other on the device (each call's stream first waits for an event
 * recorded behind the previous call's last kernel), so asynchronous calls on different streams do not race.  The one
 * exception is a stream under graph capture, which can neither wait for that event nor record it: replays of a captured
 * call must not overlap each other or any other call of this entry point on the same GPU. */
int q3asr_mel_noalloc(q3asr_model* m, const float* audio_dev, const int64_t* offsets, int n_utt, float* mel_out_dev, int t_stride,
                      int32_t* n_frames, void* stream);

/* Qwen3AudioEncoder.callAsFunction (AudioEncoder.swift:325-501), batched.  mel: [n_utt][128][t_stride]
 * as produced by q3asr_mel.  embeds_out: packed [sum N_u][enc_out_dim] fp32; cu_tokens[n_utt+1] (host)
 * receives the row offsets.  getOutputLength (AudioEncoder.swift:254-270) is q3asr_output_length. */
int q3asr_encode(q3asr_model* m, const float* mel, int t_stride, const int32_t* n_frames, int n_utt, int mel_space,
                 float* embeds_out, int64_t out_rows_capacity, int32_t* cu_tokens, int out_space);
int q3asr_output_length(const q3asr_model* m, int n_frames);
/* Device memory the model's per-call workspace currently holds (bytes; it is reused across calls and released by
 * q3asr_destroy).  Counterpart in kind of the reference's resident-memory figure (RuntimeMetrics.swift:9-28). */
int64_t q3asr_workspace_bytes(q3asr_model* m);
/* How a group of n_utt utterances is decoded on the model's device: the number of compute units its decode-attention
 * kernels are confined to while the step's other kernels run beside them on the remaining ones; 0 when the group
 * decodes on one stream (small groups).  No counterpart in the reference (its decode is one MLX stream). */
int q3asr_decode_attention_cus(q3asr_model* m, int n_utt);

typedef struct q3asr_request {
    const float* audio;          /* concatenated samples */
    const int64_t* offsets;      /* [n_utt + 1] (host) */
    int n_utt;
    int sample_rate;
    int audio_space;
    const int32_t* lang_ids;     /* tokenised " language X<asr_text>" prefixes, concatenated (host) or NULL: Qwen3ASR.swift:302-309 */
    const int32_t* lang_offsets; /* [n_utt + 1] or NULL */
    int max_tokens;              /* Qwen3ASR.swift:64, default 448 */
    int stop_rules;              /* 1: EOS / control-token / repetition stops of Qwen3ASR.swift:571-627; 0: run max_tokens steps;
                                    2 (with forced_tokens): the same rules evaluated on the teacher-forced inputs, i.e. scripted stop points */
    const int32_t* forced_tokens;/* [n_utt][max_tokens] teacher-forced decoder inputs (parity tests) or NULL; the recorded tokens stay
                                    the model's own choices, and with stop_rules 1 no rule is applied */
    /* debug taps (host pointers, may be NULL) */
    float* tap_logits;           /* [n_utt][tap_steps][vocab] fp32 logits of the first tap_steps steps */
    int tap_steps;
    float* tap_audio_embeds;     /* packed [sum N_u][enc_out_dim] in request order */
    float* tap_inputs_embeds;    /* packed [sum S_u][hidden] spliced prompt embeddings in request order */
    const float* inject_inputs_embeds; /* debug (host, may be NULL): packed [sum S_u][hidden] rows that replace the spliced prompt
                                    embeddings, i.e. the decoder is run on given inputsEmbeds (QuantizedTextDecoder.swift:384-446
                                    takes them as an argument); the prompt's token ids are then never looked up */
} q3asr_request;

typedef struct q3asr_result {
    int32_t* tokens;             /* [n_utt][max_tokens] (host, caller-allocated) */
    int32_t* n_tokens;           /* [n_utt] */
    int32_t* n_audio_tokens;     /* [n_utt] or NULL */
    double audio_seconds;        /* sum over utterances of the 16 kHz duration actually transcribed (<= 30 s each) */
    double ms_mel, ms_encoder, ms_prefill, ms_decode; /* device time per stage (HIP events) */
    int64_t decode_steps;        /* batched decode steps executed */
    int64_t tokens_total;
    int64_t decode_rows;         /* sum over those steps of the rows computed: an utterance leaves the steps soon after it stops */
} q3asr_result;

/* Qwen3ASRModel.transcribe up to the token list (Qwen3ASR.swift:60-145; _generateText :170-632):
 * mel -> encoder -> prompt ids / embedding / std rescale / splice (:229-363) -> prefill (:380-389) ->
 * greedy KV-cache loop with the stop rules (:562-632), for a batch of utterances. */
int q3asr_transcribe_tokens(q3asr_model* m, const q3asr_request* req, q3asr_result* res);

/* Per-kernel timing (no counterpart in the reference, which has no profiling: only wall-clock around
 * transcribe in Qwen3ASRIntegrationTests.swift:131-136).  Between start and stop every `stride`-th launch
 * of each kernel is bracketed by HIP events on the model's stream; stop writes a JSON array of
 * {"name","launches","timed","ms_timed","work","work_timed"} (work = algorithmic FLOPs or bytes). */
int q3asr_profile_start(q3asr_model* m, int stride);
int q3asr_profile_stop(q3asr_model* m, char* json_out, int capacity);

/* Kernel-level entry points used by the parity tests (host buffers).
 * q3asr_op_gemm: out[M][N] fp32 = act(A[M][K] * W[N][K]^T + bias), bf16 operands; tile / ksplit select the kernel variant. */
int q3asr_op_gemm(q3asr_model* m, const uint16_t* a, const uint16_t* w, const float* bias, int M, int N, int K, int gelu,
                  int tile, int ksplit, float* out);
/* q3asr_op_greedy_rules: the greedy loop's bookkeeping and stop rules alone (Qwen3ASR.swift:388-389, 567-627) on given
 * per-step argmax results: picks[step][utt] (step 0 = the prefill's token), seen0[utt] = 1 for a forced-language prompt (:565-566).
 * Writes each utterance's kept tokens [utt][max_tokens] and their count. */
int q3asr_op_greedy_rules(q3asr_model* m, const int32_t* picks, const int32_t* seen0, int n_utt, int steps, int max_tokens,
                          int stop_rules, int32_t* tokens_out, int32_t* n_tokens_out);
/* q3asr_op_read_bandwidth: GB/s of a plain streaming read of `bytes` bytes of device memory (the practical HBM read
 * ceiling reported beside the vendor peak in the benchmark's roofline; a measurement hook, no counterpart in the reference). */
int q3asr_op_read_bandwidth(q3asr_model* m, int64_t bytes, float* gb_per_s);
/* q3asr_op_mfma_f32_rate: TFLOP/s of a register-only stream of v_mfma_f32_16x16x4_f32 (4 independent accumulators per wave,
 * 1 or 2 waves per SIMD on every CU): the rate the fp32 matrix pipe sustains, i.e. the practical ceiling of the log-mel
 * kernel's folded DFT.  A measurement hook, no counterpart in the reference. */
int q3asr_op_mfma_f32_rate(q3asr_model* m, int waves_per_simd, int iters, float* tflops);
/* q3asr_op_gemm_time: average milliseconds per launch of the GEMM kernel variant (tile: 0 128x128, 1 64x64, 2 128x64,
 * 3 256x256, -1 automatic; out_kind: 0 bf16, 1 f32, 2 f32 accumulate, 3 SwiGLU, 4 argmax partials, 5 split-K slabs) on
 * device-resident pseudo-random bf16 operands, HIP events on the model's stream.  A measurement hook with no
 * counterpart in the reference (MLX's matmul is a library call there, QuantizedTextDecoder.swift:54-67). */
int q3asr_op_gemm_time(q3asr_model* m, int M, int N, int K, int tile, int ksplit, int out_kind, int iters, float* ms_per_call);
/* q3asr_op_gemm_q4: the dequant-fused variant, MLX affine 4-bit weights (QuantizedTextDecoder.swift:54-67). */
int q3asr_op_gemm_q4(q3asr_model* m, const uint16_t* a, const uint32_t* wq, const uint16_t* scales, const uint16_t* biases,
                     int M, int N, int K, int tile, float* out);
/* q3asr_op_attention: packed varlen attention; q/k/v [rows][heads*head_dim] bf16 (kv heads = heads / group). */
int q3asr_op_attention(q3asr_model* m, const uint16_t* q, const uint16_t* k, const uint16_t* v, const int32_t* cu_seqlens,
                       int n_seq, int heads, int group, int head_dim, int causal, uint16_t* out);

#ifdef __cplusplus
}
#endif
#endif
