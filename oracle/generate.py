"""Oracle: prompt build, embedding splice, greedy decode (test infrastructure).

Follows Sources/Qwen3ASR/Qwen3ASR.swift of the reference.
"""
import numpy as np
from . import mel as mel_mod
from . import encoder as enc_mod
from .decoder import Decoder

f32 = np.float32

IM_START, IM_END, ENDOFTEXT = 151644, 151645, 151643      # Qwen3ASR.swift:11-19,236-237
AUDIO_START, AUDIO_END, AUDIO_PAD = 151669, 151670, 151676  # :238-240
ASR_TEXT, NEWLINE = 151704, 198                            # :241-242
SYSTEM, USER, ASSISTANT = 8948, 872, 77091                 # :246-248


def prompt_ids(n_audio, lang_ids=None):
    """input_ids of the audio prompt -- Qwen3ASR.swift:282-317.
    Returns (ids, audio_start, audio_end)."""
    ids = [IM_START, SYSTEM, NEWLINE, IM_END, NEWLINE, IM_START, USER, NEWLINE, AUDIO_START]
    a0 = len(ids)
    ids += [AUDIO_PAD] * n_audio
    a1 = len(ids)
    ids += [AUDIO_END, IM_END, NEWLINE, IM_START, ASSISTANT, NEWLINE]
    if lang_ids is not None:                            # :302-309
        lang_ids = list(lang_ids)
        if ASR_TEXT not in lang_ids:
            lang_ids.append(ASR_TEXT)
        ids += lang_ids
    return ids, a0, a1


def splice(dec, ids, a0, a1, audio_embeds, embed_scale=True):
    """embedTokens + std rescale + concat -- Qwen3ASR.swift:324-363.
    std = sqrt(population variance) over ALL elements (all S rows incl. the
    audio_pad rows for the text side), computed in fp32."""
    x = dec.embed(ids)
    a = np.asarray(audio_embeds, dtype=f32)
    if a.shape[0] > 0:
        if embed_scale:
            tstd = f32(np.sqrt(np.var(x.astype(np.float64)))) if x.size else f32(0)
            astd = f32(np.sqrt(np.var(a.astype(np.float64))))
            if astd > 0:                                # :345-349
                a = (a * f32(tstd / astd)).astype(f32)
        x = np.concatenate([x[:a0], a, x[a1:]], axis=0)
    return x.astype(f32)


def stop_after_append(tokens, seen_asr):
    """The three break conditions evaluated right after a token is appended
    -- Qwen3ASR.swift:597-627."""
    new = tokens[-1]
    if new in (IM_START, AUDIO_START, AUDIO_END, AUDIO_PAD):
        return True
    if seen_asr and len(tokens) >= 32:
        run = 0
        for t in reversed(tokens):
            if t == new:
                run += 1
            else:
                break
        if run >= 16:
            return True
    if seen_asr and len(tokens) >= 64 and len(set(tokens[-64:])) <= 4:
        return True
    return False


def greedy(dec, inputs_embeds, max_tokens=448, forced_language=False, trace=None):
    """Prefill + greedy KV-cache loop -- Qwen3ASR.swift:380-389,562-632.
    argmax takes the lowest index among equal maxima.  trace: optional list
    that receives the fp32 logits of every step."""
    h, cache = dec.forward(inputs_embeds, None)
    lg = dec.logits(h[-1:])[0]
    if trace is not None:
        trace.append(lg.copy())
    nxt = int(np.argmax(lg))
    tokens = [nxt]
    seen = bool(forced_language)                        # :565-566
    for _ in range(1, max_tokens):
        if nxt in (IM_END, ENDOFTEXT):                  # :571-574
            break
        if nxt == ASR_TEXT:                             # :576-578
            seen = True
        h, cache = dec.forward(dec.embed([nxt]), cache)  # :581-585
        lg = dec.logits(h)[0]
        if trace is not None:
            trace.append(lg.copy())
        nxt = int(np.argmax(lg))
        tokens.append(nxt)
        if stop_after_append(tokens, seen):
            break
    return tokens


def transcribe_tokens(audio, sample_rate, w, enc_cfg, dec_cfg, max_tokens=448, lang_ids=None,
                      embed_scale=True, mel_mode="exact", dec=None, trace=None, stages=None):
    """Qwen3ASRModel.transcribe up to the token list -- Qwen3ASR.swift:60-145,170-632."""
    feats = mel_mod.process(audio, sample_rate, mel_mode)
    a = enc_mod.encode(feats, w, enc_cfg)
    dec = dec or Decoder(w, dec_cfg)
    ids, a0, a1 = prompt_ids(a.shape[0], lang_ids)
    x = splice(dec, ids, a0, a1, a, embed_scale)
    if stages is not None:
        stages.update(mel=feats, audio_embeds=a, inputs_embeds=x)
    return greedy(dec, x, max_tokens, forced_language=lang_ids is not None, trace=trace)


def post_process(text):
    """Keep what follows the last "<asr_text>" -- Qwen3ASR.swift:655-662."""
    tag = "<asr_text>"
    i = text.rfind(tag)
    return text if i < 0 else text[i + len(tag):].strip()
