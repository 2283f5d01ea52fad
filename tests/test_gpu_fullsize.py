"""The benchmark configurations at their own size: 0.6B bf16 x 128 clips of 30 s, 0.6B MLX-q4 x 256 clips, a
1.7B slice of 5-300 s recordings cut into pieces, and the headline's 2048 clips of 30 s through the 1.7B model.  The CPU restatement cannot cover these in seconds, so each run is
checked through properties (audio-token counts by the reference's length algebra, AudioEncoder.swift:254-270; a
clip's tokens do not depend on its slot or on its neighbours; finite logits), three utterances are checked against
their own single-utterance runs under one pinned kernel variant, and one utterance's first-step logits are checked
against the restatement at the model's full depth.  These are the row counts at which the four-wave decode
attention, the table-driven decode plan and the large-tile epilogues run in the benchmark."""
import numpy as np
import pytest

import oracle
from parity import LOGIT_RTOL, check_rel, check_tokens
from q3asr import Qwen3ASRModel, make_config, oracle_cfgs, synth
from q3asr.segment import segment_batch

pytestmark = pytest.mark.gpu
PINS = (("Q3ASR_DECODE_TILE", "1"), ("Q3ASR_DECODE_SPLIT", "1"), ("Q3ASR_DECODE_NO_FUSE", "1"), ("Q3ASR_DECODE_GEMV", "0"),
        ("Q3ASR_ATTN_WAVES", "4"))      # the decode attention merges four or eight partial results depending on the group size


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)


def _oracle_first_logits(cfg, ck, audio):
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, dec_cfg = oracle_cfgs(cfg)
    trace = []
    oracle.generate.transcribe_tokens(audio, 16000, w, enc_cfg, dec_cfg, max_tokens=1, trace=trace, embed_scale=bool(cfg.embed_scale))
    return trace[0]


def _full_size_30s(preset, q4, n_clips, monkeypatch):
    cfg = make_config(preset, max_batch=n_clips, weight_format=1 if q4 else 0)
    ck = synth.make_checkpoint(cfg, seed=0)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        base = [synth.synth_audio(480000, 2000 + i, "noise" if i % 8 == 7 else "speechlike") for i in range(64)]
        clips = [base[i % 64] for i in range(n_clips)]
        steps = 6
        out = m.transcribe_batch(clips, 16000, max_tokens=steps, stop_rules=False)
        assert (np.asarray(out.n_audio_tokens) == 390).all()
        assert all(len(t) == steps and all(0 <= x < cfg.vocab for x in t) for t in out.tokens)
        for i in range(64, n_clips):
            assert out.tokens[i] == out.tokens[i % 64], i          # same clip, other slot: same tokens
        # finite logits and the restatement's first-step logits at full depth, one clip
        tap = m.transcribe_batch(clips[:3], 16000, max_tokens=2, stop_rules=False, tap_steps=2, tap_embeds=True)
        assert np.isfinite(tap.taps["logits"]).all()
        # the clips do differ from one another (with random weights the prompt dominates the argmax, so compare values)
        emb = tap.taps["audio_embeds"]
        assert _rel(emb[:390], emb[390:780]) > 1e-2 and not np.array_equal(tap.taps["logits"][0], tap.taps["logits"][1])
        ref = _oracle_first_logits(cfg, ck, clips[0])
        check_rel(f"first-step logits at full depth, {preset}{' q4' if q4 else ''}", tap.taps["logits"][0, 0], ref, LOGIT_RTOL)
        check_tokens(f"first token at full depth, {preset}{' q4' if q4 else ''}", out.tokens[0][:1], ref[None])
        # one kernel variant at every row count: an utterance decodes alone exactly as inside the full batch
        for k, v in PINS:
            monkeypatch.setenv(k, v)
        pinned = m.transcribe_batch(clips, 16000, max_tokens=steps, stop_rules=False)
        for u in (0, 17, 63):
            alone = m.transcribe_batch([clips[u]], 16000, max_tokens=steps, stop_rules=False)
            assert alone.tokens[0] == pinned.tokens[u], u
    finally:
        m.close()


def test_06b_bf16_128_clips_of_30s(monkeypatch):
    _full_size_30s("0.6b", False, 128, monkeypatch)


def test_06b_q4_256_clips_of_30s(monkeypatch):
    _full_size_30s("0.6b", True, 256, monkeypatch)


def test_17b_ragged_recordings_in_pieces(monkeypatch):
    """32 distinct recordings of 5-300 s (the ragged configuration's duration law, seed 4321), each present twice, cut
    every 30 s into pieces that run as one length-sorted group of the 1.7B model."""
    cfg = make_config("1.7b", max_batch=2048)
    ck = synth.make_checkpoint(cfg, seed=0)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        lens = (np.random.default_rng(4321).uniform(5.0, 300.0, 32) * 16000).astype(np.int64)
        lens[0] = 16000 * 5                                            # the shortest case exactly
        tile = [synth.synth_audio(480000, 3000 + i) for i in range(4)]
        recs = [np.tile(tile[i % 4], 11)[:int(n)] for i, n in enumerate(lens)]
        pieces, owner = segment_batch(recs + recs, 16000, "segment_fixed")
        n_half = len(pieces) // 2
        steps = 4
        out = m.transcribe_batch(pieces, 16000, max_tokens=steps, stop_rules=False)
        for p, n_a in zip(pieces, out.n_audio_tokens):
            assert int(n_a) == oracle.encoder.output_length(min(len(p) // 160, 3000))
        assert abs(out.audio_seconds - 2 * float(lens.sum()) / 16000.0) < 1e-3
        for i in range(n_half):
            assert out.tokens[i] == out.tokens[i + n_half], i        # the second copy of every recording
        tap = m.transcribe_batch(pieces[:1], 16000, max_tokens=2, stop_rules=False, tap_steps=2)
        assert np.isfinite(tap.taps["logits"]).all()
        ref = _oracle_first_logits(cfg, ck, pieces[0])                # the 5 s recording, full depth
        check_rel("first-step logits at full depth, 1.7b 5 s recording", tap.taps["logits"][0, 0], ref, LOGIT_RTOL)
        check_tokens("first token at full depth, 1.7b 5 s recording", out.tokens[0][:1], ref[None])
        for k, v in PINS:
            monkeypatch.setenv(k, v)
        pinned = m.transcribe_batch(pieces, 16000, max_tokens=steps, stop_rules=False)
        for u in (0, 5, n_half - 1):
            alone = m.transcribe_batch([pieces[u]], 16000, max_tokens=steps, stop_rules=False)
            assert alone.tokens[0] == pinned.tokens[u], u
    finally:
        m.close()


def test_17b_2048_clips_of_30s_split_decode(monkeypatch):
    """The headline configuration at its own size: 2048 clips of 30 s through the 1.7B model, decoded as two halves
    on CU-confined streams with the LM head shared between them.  A clip's tokens do not depend on its half or slot,
    keeping each half's LM head whole changes nothing, and under one pinned projection variant the two-half decode
    gives the one-stream tokens."""
    cfg = make_config("1.7b", max_batch=2048)
    ck = synth.make_checkpoint(cfg, seed=0)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        base = [synth.synth_audio(480000, 2000 + i, "noise" if i % 8 == 7 else "speechlike") for i in range(64)]
        clips = [base[i % 64] for i in range(2048)]
        steps = 6
        monkeypatch.delenv("Q3ASR_DECODE_ATT_CUS", raising=False)
        monkeypatch.delenv("Q3ASR_LM_SPLIT", raising=False)
        assert m.decode_attention_cus(2048) == 160
        out = m.transcribe_batch(clips, 16000, max_tokens=steps, stop_rules=False)
        assert (np.asarray(out.n_audio_tokens) == 390).all()
        assert all(len(t) == steps and all(0 <= x < cfg.vocab for x in t) for t in out.tokens)
        for i in range(64, 2048):
            assert out.tokens[i] == out.tokens[i % 64], i          # same clip, other slot or other half: same tokens
        monkeypatch.setenv("Q3ASR_LM_SPLIT", "0")
        whole_head = m.transcribe_batch(clips, 16000, max_tokens=steps, stop_rules=False)
        assert whole_head.tokens == out.tokens
        monkeypatch.delenv("Q3ASR_LM_SPLIT")
        for k, v in PINS:
            monkeypatch.setenv(k, v)
        two = m.transcribe_batch(clips, 16000, max_tokens=steps, stop_rules=False)
        monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "0")
        one = m.transcribe_batch(clips, 16000, max_tokens=steps, stop_rules=False)
        assert two.tokens == one.tokens
    finally:
        m.close()
