"""Host-side mirror of the reference's model API over the C ABI.

Names and argument meaning follow the reference: ``Qwen3ASRModel.from_pretrained``
/ ``transcribe(audio, sample_rate, language, max_tokens)`` (Qwen3ASR.swift:60-145,
693-770), ``WhisperFeatureExtractor.process`` (AudioPreprocessing.swift:330-344)
and ``Qwen3AudioEncoder.__call__`` (AudioEncoder.swift:325-501).  The batched
entry points (``transcribe_batch``) are what the reference lacks: it is batch-1.
All arithmetic happens in libq3asr.so on the GPU; nothing here computes.
"""
import ctypes
import json
import os
from dataclasses import dataclass, field

import numpy as np

from . import lib as _lib
from .config import Q3Config, make_config

_TAG2DT = {"f32": _lib.F32, "bf16": _lib.BF16, "u32": _lib.U32}


def _ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


@dataclass
class TranscribeOutput:
    tokens: list                       # per utterance: list[int] (generated ids, EOS included when emitted)
    n_audio_tokens: np.ndarray
    audio_seconds: float
    ms: dict                           # device time per stage
    decode_steps: int
    tokens_total: int
    decode_rows: int = 0               # sum over the decode steps of the rows computed
    taps: dict = field(default_factory=dict)


def config_from_json(cj):
    """(preset, overrides) for make_config from a checkpoint's config.json, reading what the reference's
    loadTextConfigIfPresent reads (Qwen3ASR.swift:774-818) plus the audio_config the 1.7B model needs."""
    tc = cj.get("thinker_config", cj).get("text_config", {})
    quant = cj.get("quantization") or cj.get("quantization_config")
    # encoder shape when the checkpoint names none: the true 1.7B shape beside a 2048-wide decoder, else the
    # reference's only one (0.6B)
    preset = "1.7b" if tc.get("hidden_size", 1024) == 2048 else "0.6b"
    kw = dict(weight_format=1 if quant else 0)
    for src, dst in (("hidden_size", "hidden"), ("num_hidden_layers", "layers"), ("num_attention_heads", "heads"),
                     ("num_key_value_heads", "kv_heads"), ("head_dim", "head_dim"), ("intermediate_size", "inter"),
                     ("vocab_size", "vocab"), ("rms_norm_eps", "rms_eps"), ("rope_theta", "rope_theta")):
        if src in tc:
            kw[dst] = tc[src]
    # rope_scaling.interleaved / mrope_section and quantization.group_size / bits: Qwen3ASR.swift:806-815;
    # rotaryDim = 2 * sum(mrope_section) clamped to the head (Configuration.swift:75-80)
    rs = tc.get("rope_scaling") or {}
    if isinstance(rs.get("interleaved"), bool):
        kw["rope_interleaved"] = int(rs["interleaved"])
    if isinstance(rs.get("mrope_section"), list):
        head_dim = kw.get("head_dim", 128)
        rot = min(head_dim, 2 * max(0, min(int(sum(rs["mrope_section"])), head_dim // 2)))
        if rot == 0:
            raise ValueError("config.json: an empty mrope_section (no rotary embedding at all) is not supported")
        kw["rotary_dim"] = rot
    if isinstance(quant, dict):
        for src, dst in (("group_size", "group_size"), ("bits", "bits")):
            if isinstance(quant.get(src), int):
                kw[dst] = quant[src]
    # The reference keeps its built-in encoder shape (AudioEncoder.swift:26-43); a checkpoint's own audio_config,
    # when present (HF layout: thinker_config.audio_config), is what the 1.7B model needs.
    ac = cj.get("thinker_config", cj).get("audio_config", {})
    for src, dst in (("d_model", "enc_d_model"), ("encoder_layers", "enc_layers"), ("encoder_attention_heads", "enc_heads"),
                     ("encoder_ffn_dim", "enc_ffn"), ("output_dim", "enc_out_dim"), ("downsample_hidden_size", "conv_channels"),
                     ("num_mel_bins", "n_mels"), ("n_window", "n_window"), ("n_window_infer", "n_window_infer")):
        if src in ac:
            kw[dst] = ac[src]
    return preset, kw


class WhisperFeatureExtractor:
    """AudioPreprocessing.swift:8-366 (constants :9-13)."""
    sample_rate, n_fft, hop_length, n_mels, chunk_length = 16000, 400, 160, 128, 30

    def __init__(self, model):
        self._m = model

    def mel_filterbank(self):
        out = np.empty((128, 201), dtype=np.float32)
        _lib.check(self._m._lib, self._m._lib.q3asr_mel_filterbank(self._m._h, out.ctypes.data))
        return out

    def process(self, audio, sample_rate):
        """One utterance -> [128, T] fp32 log-mel."""
        return self.process_batch([audio], sample_rate)[0]

    def process_batch(self, audios, sample_rate, t_stride=3000):
        m = self._m
        audios = [np.ascontiguousarray(a, dtype=np.float32) for a in audios]
        off = np.zeros(len(audios) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(a) for a in audios])
        cat = np.concatenate(audios) if audios else np.zeros(0, np.float32)
        if cat.size == 0:
            cat = np.zeros(1, np.float32)
        out = np.empty((len(audios), 128, t_stride), dtype=np.float32)
        nfr = np.zeros(len(audios), dtype=np.int32)
        _lib.check(m._lib, m._lib.q3asr_mel(m._h, cat.ctypes.data, _ptr(off, ctypes.c_int64), len(audios), int(sample_rate),
                                            _lib.HOST, out.ctypes.data, t_stride, _ptr(nfr, ctypes.c_int32), _lib.HOST))
        return [np.ascontiguousarray(out[i, :, :nfr[i]]) for i in range(len(audios))]

    def process_device(self, audio_ptr, offsets, out_ptr, t_stride=3000, sample_rate=16000):
        """Same stage on buffers that already live in HBM: `audio_ptr` = device address of the concatenated fp32
        samples, `out_ptr` = device address of an [n][128][t_stride] fp32 buffer the log-mel is written into.
        Returns the frame count of every utterance."""
        m = self._m
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        nfr = np.zeros(len(off) - 1, dtype=np.int32)
        _lib.check(m._lib, m._lib.q3asr_mel(m._h, audio_ptr, _ptr(off, ctypes.c_int64), len(off) - 1, int(sample_rate),
                                            _lib.DEVICE, out_ptr, int(t_stride), _ptr(nfr, ctypes.c_int32), _lib.DEVICE))
        return nfr


    def process_device_noalloc(self, audio_ptr, offsets, out_ptr, t_stride=3000, stream=None):
        """process_device through the entry point that allocates nothing (fixed device scratch; 16 kHz, at most 256
        utterances per call).  stream: a hipStream_t address to enqueue on without waiting, or None."""
        m = self._m
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        nfr = np.zeros(len(off) - 1, dtype=np.int32)
        _lib.check(m._lib, m._lib.q3asr_mel_noalloc(m._h, audio_ptr, _ptr(off, ctypes.c_int64), len(off) - 1, out_ptr, int(t_stride),
                                                    _ptr(nfr, ctypes.c_int32), stream))
        return nfr


class Qwen3AudioEncoder:
    """AudioEncoder.swift:173-502."""

    def __init__(self, model):
        self._m = model

    def output_length(self, n_frames):
        return int(self._m._lib.q3asr_output_length(self._m._h, int(n_frames)))

    def __call__(self, mel):
        """mel [128, T] fp32 (or [1, 128, T]) -> [N, out_dim] fp32."""
        mel = np.asarray(mel, dtype=np.float32)
        if mel.ndim == 3:
            mel = mel[0]                       # AudioEncoder.swift:356 reads batch item 0 only
        return self.encode_batch([mel])[0]

    def encode_batch(self, mels):
        m = self._m
        t_stride = max(4, max(x.shape[1] for x in mels))
        buf = np.zeros((len(mels), 128, t_stride), dtype=np.float32)
        nfr = np.zeros(len(mels), dtype=np.int32)
        for i, x in enumerate(mels):
            buf[i, :, :x.shape[1]] = x
            nfr[i] = x.shape[1]
        cap = int(sum(self.output_length(int(n)) for n in nfr))
        out = np.empty((max(cap, 1), m.cfg.enc_out_dim), dtype=np.float32)
        cu = np.zeros(len(mels) + 1, dtype=np.int32)
        _lib.check(m._lib, m._lib.q3asr_encode(m._h, buf.ctypes.data, t_stride, _ptr(nfr, ctypes.c_int32), len(mels), _lib.HOST,
                                               out.ctypes.data, cap, _ptr(cu, ctypes.c_int32), _lib.HOST))
        return [out[cu[i]:cu[i + 1]].copy() for i in range(len(mels))]


class Qwen3ASRModel:
    def __init__(self, cfg: Q3Config, device=0):
        self._lib = _lib.load()
        self.cfg = cfg
        self.device = device
        h = ctypes.c_void_p()
        _lib.check(self._lib, self._lib.q3asr_create(ctypes.byref(cfg), int(device), ctypes.byref(h)))
        self._h = h
        self.feature_extractor = WhisperFeatureExtractor(self)
        self.audio_encoder = Qwen3AudioEncoder(self)
        self.tokenizer = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.q3asr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights (WeightLoading.swift) ----
    def load_tensor(self, name, arr, tag):
        arr = np.ascontiguousarray(arr)
        shape = (ctypes.c_int64 * arr.ndim)(*arr.shape)
        _lib.check(self._lib, self._lib.q3asr_load_tensor(self._h, name.encode(), _TAG2DT[tag], shape, arr.ndim, arr.ctypes.data))

    def load_checkpoint(self, ck):
        for name, (arr, tag) in ck.items():
            self.load_tensor(name, arr, tag)
        _lib.check(self._lib, self._lib.q3asr_finalize(self._h))

    @classmethod
    def from_checkpoint(cls, cfg, ck, device=0):
        m = cls(cfg, device)
        m.load_checkpoint(ck)
        return m

    @classmethod
    def from_pretrained(cls, model_dir, device=0, **overrides):
        """Local-directory form of Qwen3ASRModel.fromPretrained (Qwen3ASR.swift:693-770): reads
        config.json (text_config + quantization, :774-818) and every *.safetensors file."""
        from .safetensors_io import load_dir
        with open(os.path.join(model_dir, "config.json")) as f:
            cj = json.load(f)
        preset, kw = config_from_json(cj)
        kw.update(overrides)
        m = cls(make_config(preset, **kw), device)
        m.load_checkpoint(load_dir(model_dir))
        vocab = os.path.join(model_dir, "vocab.json")
        if os.path.exists(vocab):                       # Qwen3ASR.swift:728-739
            from .tokenizer import Qwen3Tokenizer
            m.tokenizer = Qwen3Tokenizer.load(vocab)
        return m

    # ---- transcription ----
    def transcribe(self, audio, sample_rate=24000, language=None, max_tokens=448):
        """Qwen3ASRModel.transcribe (Qwen3ASR.swift:60-145).  Without a tokenizer the reference
        returns the generated ids joined by spaces (:647-650); so does this."""
        lang_ids = None
        if language is not None:
            if self.tokenizer is None:
                raise ValueError("forcing a language needs a tokenizer to encode ' language X<asr_text>'")
            lang_ids = [self.tokenizer.encode(f" language {language}<asr_text>")]
        out = self.transcribe_batch([audio], sample_rate, max_tokens=max_tokens, lang_ids=lang_ids)
        toks = out.tokens[0]
        if self.tokenizer is not None:
            return self.tokenizer.post_process(self.tokenizer.decode(toks))
        return " ".join(str(t) for t in toks)

    def transcribe_long(self, audios, sample_rate=16000, mode="truncate", max_tokens=448, batch=None):
        """Recordings of any length.  mode="truncate" is the reference's behaviour (everything after 30 s is
        dropped, AudioPreprocessing.swift:307-316); "segment" / "segment_fixed" cut each recording into pieces of
        at most 30 s (q3asr.segment), run all pieces as one sharded batch and join them per recording.
        Returns (texts or joined id lists, per-recording list of piece token lists)."""
        from .segment import join_tokens, segment_batch
        pieces, owner = segment_batch(audios, sample_rate, mode)
        step = batch or max(1, int(self.cfg.max_batch))
        piece_tokens = []
        for i in range(0, len(pieces), step):
            piece_tokens += self.transcribe_batch(pieces[i:i + step], sample_rate, max_tokens=max_tokens).tokens
        per_utt = [[] for _ in audios]
        for toks, i in zip(piece_tokens, owner):
            per_utt[i].append(toks)
        if self.tokenizer is not None:
            joined = [" ".join(t for t in (self.tokenizer.post_process(self.tokenizer.decode(p)) for p in ps) if t) for ps in per_utt]
        else:
            joined = join_tokens(piece_tokens, owner, len(audios))
        return joined, per_utt

    def transcribe_batch(self, audios, sample_rate=16000, max_tokens=448, lang_ids=None, stop_rules=True, forced_tokens=None,
                         tap_steps=0, tap_embeds=False, device_audio=None, inject_inputs_embeds=None):
        """audios: list of fp32 arrays (host).  device_audio: (device pointer, offsets int64[n+1]) of samples
        already resident in HBM, used instead of `audios`."""
        if device_audio is not None:
            ptr, off = device_audio
            off = np.ascontiguousarray(off, dtype=np.int64)
            n, space = len(off) - 1, _lib.DEVICE
            keep = None
        else:
            audios = [np.ascontiguousarray(a, dtype=np.float32) for a in audios]
            n = len(audios)
            off = np.zeros(n + 1, dtype=np.int64)
            off[1:] = np.cumsum([len(a) for a in audios])
            keep = np.concatenate(audios) if n else np.zeros(1, np.float32)
            ptr, space = keep.ctypes.data, _lib.HOST
        req = _lib.Q3Request()
        req.audio, req.offsets, req.n_utt = ptr, _ptr(off, ctypes.c_int64), n
        req.sample_rate, req.audio_space = int(sample_rate), space
        # stop_rules: True / False, or "forced" = the reference's rules evaluated on the teacher-forced inputs (scripted stops)
        req.max_tokens, req.stop_rules = int(max_tokens), 2 if stop_rules == "forced" else (1 if stop_rules else 0)
        holders = [off, keep]
        if lang_ids is not None:
            lo = np.zeros(n + 1, dtype=np.int32)
            lo[1:] = np.cumsum([len(x) if x is not None else 0 for x in lang_ids])
            li = np.asarray([t for x in lang_ids if x is not None for t in x] or [0], dtype=np.int32)
            req.lang_ids, req.lang_offsets = _ptr(li, ctypes.c_int32), _ptr(lo, ctypes.c_int32)
            holders += [lo, li]
        if forced_tokens is not None:
            ft = np.ascontiguousarray(forced_tokens, dtype=np.int32).reshape(n, max_tokens)
            req.forced_tokens = _ptr(ft, ctypes.c_int32)
            holders.append(ft)
        if inject_inputs_embeds is not None:      # packed [sum S_u][hidden] rows that replace the spliced prompt embeddings
            inj = np.ascontiguousarray(inject_inputs_embeds, dtype=np.float32)
            req.inject_inputs_embeds = _ptr(inj, ctypes.c_float)
            holders.append(inj)
        taps = {}
        if tap_steps:
            taps["logits"] = np.zeros((n, tap_steps, self.cfg.vocab), dtype=np.float32)
            req.tap_logits, req.tap_steps = _ptr(taps["logits"], ctypes.c_float), int(tap_steps)
        if tap_embeds:
            cap = n * (390 + 64)
            taps["audio_embeds"] = np.zeros((cap, self.cfg.hidden), dtype=np.float32)
            taps["inputs_embeds"] = np.zeros((cap, self.cfg.hidden), dtype=np.float32)
            req.tap_audio_embeds = _ptr(taps["audio_embeds"], ctypes.c_float)
            req.tap_inputs_embeds = _ptr(taps["inputs_embeds"], ctypes.c_float)
        tokens = np.zeros((n, max_tokens), dtype=np.int32)
        ntok = np.zeros(n, dtype=np.int32)
        naud = np.zeros(n, dtype=np.int32)
        res = _lib.Q3Result()
        res.tokens, res.n_tokens, res.n_audio_tokens = _ptr(tokens, ctypes.c_int32), _ptr(ntok, ctypes.c_int32), _ptr(naud, ctypes.c_int32)
        _lib.check(self._lib, self._lib.q3asr_transcribe_tokens(self._h, ctypes.byref(req), ctypes.byref(res)))
        del holders
        return TranscribeOutput(
            tokens=[tokens[i, :ntok[i]].tolist() for i in range(n)], n_audio_tokens=naud, audio_seconds=res.audio_seconds,
            ms=dict(mel=res.ms_mel, encoder=res.ms_encoder, prefill=res.ms_prefill, decode=res.ms_decode),
            decode_steps=int(res.decode_steps), tokens_total=int(res.tokens_total), decode_rows=int(res.decode_rows), taps=taps)

    def workspace_bytes(self):
        """Device memory the per-call workspace holds at the moment."""
        return int(self._lib.q3asr_workspace_bytes(self._h))

    def decode_attention_cus(self, n_utt):
        """Compute units the decode attention of a group of n_utt utterances is confined to (0: one stream)."""
        return int(self._lib.q3asr_decode_attention_cus(self._h, int(n_utt)))

    # ---- per-kernel timing ----
    def profile_start(self, stride=1):
        _lib.check(self._lib, self._lib.q3asr_profile_start(self._h, int(stride)))

    def profile_stop(self):
        buf = ctypes.create_string_buffer(1 << 16)
        _lib.check(self._lib, self._lib.q3asr_profile_stop(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    # ---- kernel-level entry points (parity tests) ----
    def op_gemm(self, a_bits, w_bits, bias=None, gelu=False, tile=-1, ksplit=1):
        a_bits, w_bits = np.ascontiguousarray(a_bits, np.uint16), np.ascontiguousarray(w_bits, np.uint16)
        (M, K), N = a_bits.shape, w_bits.shape[0]
        out = np.empty((M, N), dtype=np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        _lib.check(self._lib, self._lib.q3asr_op_gemm(self._h, a_bits.ctypes.data, w_bits.ctypes.data, None if b is None else b.ctypes.data,
                                                      M, N, K, int(gelu), tile, ksplit, out.ctypes.data))
        return out

    def op_greedy_rules(self, picks, seen0=None, max_tokens=448, stop_rules=True):
        """picks: int32 [n_utt, steps] argmax results per step (column 0 = the prefill's token).  Returns the token lists
        the device-side greedy bookkeeping keeps."""
        picks = np.ascontiguousarray(np.asarray(picks, dtype=np.int32))
        n, steps = picks.shape
        by_step = np.ascontiguousarray(picks.T)
        seen = None if seen0 is None else np.ascontiguousarray(seen0, dtype=np.int32)
        toks = np.zeros((n, max_tokens), dtype=np.int32)
        cnt = np.zeros(n, dtype=np.int32)
        _lib.check(self._lib, self._lib.q3asr_op_greedy_rules(self._h, by_step.ctypes.data, None if seen is None else seen.ctypes.data,
                                                              n, steps, int(max_tokens), int(stop_rules), toks.ctypes.data, cnt.ctypes.data))
        return [toks[i, :cnt[i]].tolist() for i in range(n)]

    def op_read_bandwidth(self, nbytes=1 << 32):
        """GB/s of a plain streaming read of device memory (practical HBM read ceiling)."""
        v = ctypes.c_float()
        _lib.check(self._lib, self._lib.q3asr_op_read_bandwidth(self._h, int(nbytes), ctypes.byref(v)))
        return v.value

    def op_mfma_f32_rate(self, waves_per_simd=2, iters=20000):
        """TFLOP/s of back-to-back fp32 16x16x4 MFMAs from registers (the log-mel transform's practical ceiling)."""
        out = ctypes.c_float()
        _lib.check(self._lib, self._lib.q3asr_op_mfma_f32_rate(self._h, int(waves_per_simd), int(iters), ctypes.byref(out)))
        return float(out.value)

    def op_gemm_time(self, M, N, K, tile=-1, ksplit=1, out_kind=0, iters=20):
        """Average ms per launch of one GEMM kernel variant on device-resident random operands."""
        ms = ctypes.c_float()
        _lib.check(self._lib, self._lib.q3asr_op_gemm_time(self._h, M, N, K, tile, ksplit, out_kind, iters, ctypes.byref(ms)))
        return ms.value

    def op_gemm_q4(self, a_bits, wq, scales_bits, biases_bits, tile=-1):
        a_bits = np.ascontiguousarray(a_bits, np.uint16)
        wq = np.ascontiguousarray(wq, np.uint32)
        s, b = np.ascontiguousarray(scales_bits, np.uint16), np.ascontiguousarray(biases_bits, np.uint16)
        (M, K), N = a_bits.shape, wq.shape[0]
        out = np.empty((M, N), dtype=np.float32)
        _lib.check(self._lib, self._lib.q3asr_op_gemm_q4(self._h, a_bits.ctypes.data, wq.ctypes.data, s.ctypes.data, b.ctypes.data,
                                                         M, N, K, tile, out.ctypes.data))
        return out

    def op_attention(self, q_bits, k_bits, v_bits, cu_seqlens, heads, group, head_dim, causal):
        q, k, v = (np.ascontiguousarray(x, np.uint16) for x in (q_bits, k_bits, v_bits))
        cu = np.ascontiguousarray(cu_seqlens, np.int32)
        out = np.empty_like(q)
        _lib.check(self._lib, self._lib.q3asr_op_attention(self._h, q.ctypes.data, k.ctypes.data, v.ctypes.data,
                                                           _ptr(cu, ctypes.c_int32), len(cu) - 1, heads, group, head_dim,
                                                           int(causal), out.ctypes.data))
        return out
