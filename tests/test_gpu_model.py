"""GPU parity of the encoder, prompt splice, prefill and greedy decode against the CPU restatement."""
import numpy as np
import pytest

import oracle
import parity
from parity import ENC_RTOL, LOGIT_RTOL, check_rel, check_tokens
from q3asr import Qwen3ASRModel, make_config, oracle_cfgs, synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)


@pytest.mark.parametrize("frames", [250, 100, 37, 3000, 1234])
def test_encoder_matches_oracle(tiny_model, frames):
    m, cfg, ck = tiny_model
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, _ = oracle_cfgs(cfg)
    mel = oracle.mel.process(synth.synth_audio(frames * 160, frames), 16000)
    assert mel.shape[1] == frames
    got = m.audio_encoder(mel)
    ref = oracle.encoder.encode(mel, w, enc_cfg)
    assert got.shape == ref.shape == (oracle.encoder.output_length(frames), cfg.enc_out_dim)
    check_rel(f"encoder vs restatement, {frames} frames", got, ref, ENC_RTOL)


def test_encoder_batch_equals_single(tiny_model):
    m, cfg, _ = tiny_model
    mels = [oracle.mel.process(synth.synth_audio(n * 160, n), 16000) for n in (250, 1000, 99, 100)]
    batch = m.audio_encoder.encode_batch(mels)
    for x, b in zip(mels, batch):
        assert np.array_equal(m.audio_encoder(x), b)


def _oracle_run(cfg, ck, audio, max_tokens, embed_scale=True):
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, dec_cfg = oracle_cfgs(cfg)
    trace, stages = [], {}
    toks = oracle.generate.transcribe_tokens(audio, 16000, w, enc_cfg, dec_cfg, max_tokens=max_tokens, trace=trace,
                                             stages=stages, embed_scale=embed_scale)
    return toks, np.stack(trace), stages


@pytest.mark.parametrize("variant", ["default", "rotate_half", "partial_rotary", "partial_rotate_half", "no_embed_scale", "q4",
                                     "attn4"])
def test_transcribe_teacher_forced_logits(variant, monkeypatch):
    """Per-step logits under teacher forcing with the oracle's own tokens, every decode path
    (interleaved / rotate-half RoPE over the whole head or its first 64 channels, embed rescale on / off,
    dense / MLX-q4 weights)."""
    kw = dict(rotate_half=dict(rope_interleaved=0), partial_rotary=dict(rotary_dim=64),
              partial_rotate_half=dict(rope_interleaved=0, rotary_dim=64), no_embed_scale=dict(embed_scale=0),
              q4=dict(weight_format=1)).get(variant, {})
    if variant == "attn4":      # the large groups' four-wave decode attention (a single utterance would take eight)
        monkeypatch.setenv("Q3ASR_ATTN_WAVES", "4")
    cfg = make_config("tiny", **kw)
    ck = synth.make_checkpoint(cfg, seed=3)
    audio = synth.synth_audio(16000 * 2 + 4000, 21)
    steps = 10
    toks, ref_logits, stages = _oracle_run(cfg, ck, audio, steps, embed_scale=bool(cfg.embed_scale))
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        forced = np.zeros((1, steps), dtype=np.int32)
        forced[0, :len(toks)] = toks
        out = m.transcribe_batch([audio], 16000, max_tokens=steps, forced_tokens=forced, tap_steps=steps, tap_embeds=True)
        n_a = int(out.n_audio_tokens[0])
        assert n_a == stages["audio_embeds"].shape[0]
        check_rel(f"audio embeds [{variant}]", out.taps["audio_embeds"][:n_a], stages["audio_embeds"], ENC_RTOL)
        s = stages["inputs_embeds"].shape[0]
        check_rel(f"inputs embeds [{variant}]", out.taps["inputs_embeds"][:s], stages["inputs_embeds"], ENC_RTOL)
        got = out.taps["logits"][0, :len(toks)]
        check_rel(f"teacher-forced logits [{variant}]", got, ref_logits[:len(toks)], LOGIT_RTOL)
        # tokens agree wherever the oracle's top-2 margin exceeds the gate
        check_tokens(f"teacher-forced tokens [{variant}]", out.tokens[0], ref_logits[:len(toks)])
    finally:
        m.close()


@pytest.mark.parametrize("tile", ["0", "2", "3"])
def test_every_gemm_tile_through_the_model(tiny_model, tile, monkeypatch):
    """The automatic tile choice depends on the row count, so a small model never reaches the large tiles on its
    own: pin each tile in turn and run the encoder, the prefill epilogues (bf16, residual add, SwiGLU) and the
    fused LM-head argmax through it."""
    m, cfg, ck = tiny_model
    monkeypatch.setenv("Q3ASR_FORCE_TILE", tile)
    audio = synth.synth_audio(16000 * 3 + 1234, 5)
    steps = 6
    toks, ref_logits, stages = _oracle_run(cfg, ck, audio, steps)
    forced = np.zeros((1, steps), dtype=np.int32)
    forced[0, :len(toks)] = toks
    out = m.transcribe_batch([audio], 16000, max_tokens=steps, forced_tokens=forced, tap_steps=steps, tap_embeds=True)
    n_a = int(out.n_audio_tokens[0])
    check_rel(f"audio embeds, tile {tile}", out.taps["audio_embeds"][:n_a], stages["audio_embeds"], ENC_RTOL)
    check_rel(f"logits, tile {tile}", out.taps["logits"][0, :len(toks)], ref_logits[:len(toks)], LOGIT_RTOL)
    check_tokens(f"tokens, tile {tile}", out.tokens[0], ref_logits[:len(toks)])
    audios = [synth.synth_audio(n, 50 + i) for i, n in enumerate((16000, 40000, 52345))]
    a = m.transcribe_batch(audios, 16000, max_tokens=8, stop_rules=False)
    b = m.transcribe_batch(audios, 16000, max_tokens=8, stop_rules=False, tap_steps=8)
    assert a.tokens == b.tokens


def test_fused_argmax_equals_logit_tap(tiny_model):
    """The LM-head + argmax epilogue must pick the same ids as argmax over the tapped fp32 logits."""
    m, cfg, _ = tiny_model
    audios = [synth.synth_audio(n, 50 + i) for i, n in enumerate((16000, 40000, 52345))]
    a = m.transcribe_batch(audios, 16000, max_tokens=12, stop_rules=False)
    b = m.transcribe_batch(audios, 16000, max_tokens=12, stop_rules=False, tap_steps=12)
    assert a.tokens == b.tokens
    for i in range(len(audios)):
        assert np.array_equal(np.argmax(b.taps["logits"][i], axis=-1), np.array(b.tokens[i]))


def test_batch_equals_single_and_order(pinned_model):
    """Utterances are sorted by length internally; results come back in request order and do not
    depend on what else is in the batch."""
    m, cfg, _ = pinned_model
    audios = [synth.synth_audio(n, 70 + i) for i, n in enumerate((20000, 100000, 16000, 64000))]
    batch = m.transcribe_batch(audios, 16000, max_tokens=8, stop_rules=False)
    for i, a in enumerate(audios):
        single = m.transcribe_batch([a], 16000, max_tokens=8, stop_rules=False)
        assert single.tokens[0] == batch.tokens[i]
    assert abs(batch.audio_seconds - sum(len(a) for a in audios) / 16000.0) < 1e-6


def test_ragged_corpus_in_groups(monkeypatch):
    """40 utterances of 0.2-33 s through a model that decodes at most 12 at a time: several length-sorted groups,
    results back in request order, audio-token counts by the reference's length algebra (AudioEncoder.swift:254-270),
    and every fifth utterance identical to its own single-utterance run."""
    for k, v in (("Q3ASR_DECODE_TILE", "1"), ("Q3ASR_DECODE_SPLIT", "1"), ("Q3ASR_DECODE_NO_FUSE", "1"), ("Q3ASR_DECODE_GEMV", "0")):
        monkeypatch.setenv(k, v)        # one kernel variant for every group size (see conftest.pinned_model)
    cfg = make_config("tiny", max_batch=12)
    ck = synth.make_checkpoint(cfg, seed=1)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        rng = np.random.default_rng(77)
        lens = [int(x) for x in rng.uniform(0.2, 33.0, size=40) * 16000]
        lens[3], lens[17] = 160 * 100, 160 * 3000 + 159            # exactly one chunk; exactly 3000 frames
        audios = [synth.synth_audio(n, 400 + i) for i, n in enumerate(lens)]
        out = m.transcribe_batch(audios, 16000, max_tokens=6, stop_rules=False)
        assert len(out.tokens) == 40 and all(len(t) == 6 for t in out.tokens)
        for i, n in enumerate(lens):
            assert int(out.n_audio_tokens[i]) == oracle.encoder.output_length(min(n // 160, 3000))
        assert abs(out.audio_seconds - sum(min(n, 480000) for n in lens) / 16000.0) < 1e-6
        for i in range(0, 40, 5):
            single = m.transcribe_batch([audios[i]], 16000, max_tokens=6, stop_rules=False)
            assert single.tokens[0] == out.tokens[i], i
    finally:
        m.close()


def test_split_decode_equals_one_stream(pinned_model, monkeypatch):
    """Large groups are decoded as two halves, their attention kernels on a stream confined to part of the compute
    units and everything else on a stream confined to the rest; a small group is forced through that path here and
    must give the tokens (and stop points) of the single-stream loop."""
    m, cfg, _ = pinned_model
    audios = [synth.synth_audio(n, 80 + i) for i, n in enumerate((20000, 100000, 16000, 64000, 33333))]
    # the library's own choice: one stream for small groups, 5/8 of the compute units for the attention of large ones
    monkeypatch.delenv("Q3ASR_DECODE_ATT_CUS", raising=False)
    assert m.decode_attention_cus(5) == 0 and m.decode_attention_cus(639) == 0 and m.decode_attention_cus(640) > 0
    big = m.decode_attention_cus(2048)
    assert big > 0 and big % 8 == 0
    monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "64")
    assert m.decode_attention_cus(5) == 64 and m.decode_attention_cus(1) == 0
    monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "0")
    one = m.transcribe_batch(audios, 16000, max_tokens=24)
    for att_cus in ("160", "64"):
        monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", att_cus)
        two = m.transcribe_batch(audios, 16000, max_tokens=24)
        assert one.tokens == two.tokens, att_cus
        assert one.decode_steps == two.decode_steps
    # the step's LM head shared between the two streams by vocabulary range (forced here: a group this small would
    # keep it on one stream), with the cut at the default place, at the far end and with nothing on the attention stream
    monkeypatch.setenv("Q3ASR_LM_SPLIT", "2")
    for lead in (None, "-4000", "4000"):
        if lead is not None:
            monkeypatch.setenv("Q3ASR_LM_SPLIT_LEAD", lead)
        two = m.transcribe_batch(audios, 16000, max_tokens=24)
        assert one.tokens == two.tokens, lead
    monkeypatch.delenv("Q3ASR_LM_SPLIT_LEAD")
    monkeypatch.setenv("Q3ASR_LM_SPLIT", "0")
    two = m.transcribe_batch(audios, 16000, max_tokens=24)
    assert one.tokens == two.tokens
    # each half's attention runs on a stream of its own (both confined to the attention's compute units) by default:
    # both halves on one attention stream must give the same
    monkeypatch.setenv("Q3ASR_LM_SPLIT", "2")
    monkeypatch.setenv("Q3ASR_ATT_STREAMS", "1")
    two = m.transcribe_batch(audios, 16000, max_tokens=24)
    assert one.tokens == two.tokens
    monkeypatch.delenv("Q3ASR_ATT_STREAMS")
    monkeypatch.delenv("Q3ASR_LM_SPLIT")
    forced = np.tile(np.arange(100, 110, dtype=np.int32), (5, 1))
    a = m.transcribe_batch(audios, 16000, max_tokens=10, forced_tokens=forced, tap_steps=10)
    monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "0")
    b = m.transcribe_batch(audios, 16000, max_tokens=10, forced_tokens=forced, tap_steps=10)
    assert _rel(a.taps["logits"], b.taps["logits"]) <= 1e-3


def test_large_group_takes_the_split_decode_by_itself(monkeypatch):
    """1100 short clips in one group: the library's own choices apply (two halves, attention on its share of the
    compute units, 256x256 LM head shared between the two streams by vocabulary range) and must give, clip by clip,
    the free-running tokens of the one-stream decode.  Projections pinned to one summation order as in pinned_model."""
    for k, v in (("Q3ASR_DECODE_TILE", "1"), ("Q3ASR_DECODE_SPLIT", "1"), ("Q3ASR_DECODE_NO_FUSE", "1"), ("Q3ASR_DECODE_GEMV", "0")):
        monkeypatch.setenv(k, v)
    cfg = make_config("tiny", max_batch=1100)
    ck = synth.make_checkpoint(cfg, seed=1)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck, device=0)
    try:
        base = [synth.synth_audio(9600 + 800 * i, 300 + i) for i in range(11)]
        audios = [base[i % 11] for i in range(1100)]
        monkeypatch.delenv("Q3ASR_DECODE_ATT_CUS", raising=False)
        assert m.decode_attention_cus(1100) > 0
        two = m.transcribe_batch(audios, 16000, max_tokens=12)
        monkeypatch.setenv("Q3ASR_LM_SPLIT", "0")
        whole_head = m.transcribe_batch(audios, 16000, max_tokens=12)
        monkeypatch.setenv("Q3ASR_DECODE_ATT_CUS", "0")
        one = m.transcribe_batch(audios, 16000, max_tokens=12)
        assert two.tokens == one.tokens and whole_head.tokens == one.tokens
        for i in range(11, 1100):                       # the same clip in another slot (and the other half)
            assert two.tokens[i] == two.tokens[i % 11]
    finally:
        m.close()


@pytest.mark.parametrize("variant", ["default", "rotate_half"])
def test_rope_inside_attention_equals_separate_kernel(variant, monkeypatch):
    """Decode steps normalise / rotate q and k and append to the cache inside the attention kernel; the separate
    kernel (the prefill's, Q3ASR_FUSE_ROPE=0) must give the same tokens and logits."""
    cfg = make_config("tiny", **(dict(rope_interleaved=0) if variant == "rotate_half" else {}))
    ck = synth.make_checkpoint(cfg, seed=11)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        audios = [synth.synth_audio(n, 90 + i) for i, n in enumerate((24000, 70000, 16000))]
        forced = np.tile(np.arange(200, 212, dtype=np.int32), (3, 1))
        fused = m.transcribe_batch(audios, 16000, max_tokens=12, forced_tokens=forced, tap_steps=12)
        monkeypatch.setenv("Q3ASR_FUSE_ROPE", "0")
        plain = m.transcribe_batch(audios, 16000, max_tokens=12, forced_tokens=forced, tap_steps=12)
        # the new key's score is an fp32 VALU dot product in one path and an MFMA in the other; a flipped bf16
        # rounding of an attention output moves a logit by ~1e-3 of the largest
        assert _rel(fused.taps["logits"], plain.taps["logits"]) <= 5e-3
    finally:
        m.close()


@pytest.mark.parametrize("q4", [False, True])
@pytest.mark.parametrize("n_utt", [1, 3, 4])
def test_few_row_projection_kernels_equal_the_mfma_path(n_utt, q4, monkeypatch):
    """Groups of up to four utterances decode through the fused GEMV kernels (norm + q/k/v, o += residual,
    norm + gate/up + SwiGLU, down += residual); the MFMA path (Q3ASR_DECODE_GEMV=0) must give the same logits up to
    bf16 rounding noise on a teacher-forced run (parity with the restatement itself: test_transcribe_teacher_forced_logits,
    whose single utterance takes these kernels)."""
    cfg = make_config("tiny", weight_format=1 if q4 else 0)      # q4: MLX affine 4-bit decoder weights
    ck = synth.make_checkpoint(cfg, seed=21)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        audios = [synth.synth_audio(n, 60 + i) for i, n in enumerate((30000, 52000, 16000, 81000)[:n_utt])]
        forced = np.tile(np.arange(300, 310, dtype=np.int32), (n_utt, 1))
        gemv = m.transcribe_batch(audios, 16000, max_tokens=10, forced_tokens=forced, tap_steps=10)
        monkeypatch.setenv("Q3ASR_DECODE_GEMV", "0")
        mfma = m.transcribe_batch(audios, 16000, max_tokens=10, forced_tokens=forced, tap_steps=10)
        assert np.isfinite(gemv.taps["logits"]).all()
        # Two bf16 pipelines that round their activations at different sums: measured 5e-3 of the largest logit at every
        # step (no growth), the same as either path's distance to the fp32 restatement on this model.
        assert _rel(gemv.taps["logits"], mfma.taps["logits"]) <= 1e-2
        assert np.array_equal(gemv.taps["logits"][:, 0], mfma.taps["logits"][:, 0])   # step 0 is the prefill, shared by both
    finally:
        m.close()


def test_decode_attention_wave_counts_agree(monkeypatch):
    """Groups with fewer (utterance, kv head) pairs than CUs stream each pair's keys with eight waves instead of four
    (every test-sized group is that small, so the four-wave kernel of the large groups is pinned here explicitly):
    same logits up to the order in which the partial softmax results are merged."""
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=29)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        audios = [synth.synth_audio(n, 40 + i) for i, n in enumerate((32000, 480000, 90000))]
        forced = np.tile(np.arange(500, 512, dtype=np.int32), (3, 1))
        runs = {}
        for waves in ("4", "8"):
            monkeypatch.setenv("Q3ASR_ATTN_WAVES", waves)
            runs[waves] = m.transcribe_batch(audios, 16000, max_tokens=12, forced_tokens=forced, tap_steps=12).taps["logits"]
            assert np.isfinite(runs[waves]).all()
        assert _rel(runs["4"], runs["8"]) <= 5e-3      # a flipped bf16 rounding of an attention output
        assert np.array_equal(runs["4"][:, 0], runs["8"][:, 0])   # the prefill does not use this kernel
    finally:
        m.close()


@pytest.mark.parametrize("preset", ["0.6b", "1.7b", "0.6b-q4"])
def test_true_widths_two_layers(preset):
    """The real layer widths (896 / 1024 encoder, 1024 / 2048 decoder, 14 / 16 encoder heads, 16 Q / 8 KV heads,
    vocabulary 151 936) with two layers per stack, so that the CPU restatement finishes in seconds: catches tile-edge
    and head-count assumptions the tiny configuration cannot."""
    cfg = make_config(preset.split("-")[0], enc_layers=2, layers=2, weight_format=1 if preset.endswith("q4") else 0)
    ck = synth.make_checkpoint(cfg, seed=5)
    audio = synth.synth_audio(16000 * 2 + 800, 31)
    steps = 4
    toks, ref_logits, stages = _oracle_run(cfg, ck, audio, steps)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        forced = np.zeros((1, steps), dtype=np.int32)
        forced[0, :len(toks)] = toks
        out = m.transcribe_batch([audio], 16000, max_tokens=steps, forced_tokens=forced, tap_steps=steps, tap_embeds=True)
        n_a = int(out.n_audio_tokens[0])
        assert n_a == stages["audio_embeds"].shape[0]
        check_rel(f"audio embeds, true widths {preset}", out.taps["audio_embeds"][:n_a], stages["audio_embeds"], ENC_RTOL)
        check_rel(f"logits, true widths {preset}", out.taps["logits"][0, :len(toks)], ref_logits[:len(toks)], LOGIT_RTOL)
        check_tokens(f"tokens, true widths {preset}", out.tokens[0], ref_logits[:len(toks)])
    finally:
        m.close()


def test_forced_language_prefix_matches_oracle(pinned_model):
    """Forced-language prompts (Qwen3ASR.swift:302-309): the prefix ids follow the assistant header, <asr_text> is
    appended when missing, and the repetition rules are armed from the first token (:565-566)."""
    m, cfg, ck = pinned_model
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, dec_cfg = oracle_cfgs(cfg)
    audio = synth.synth_audio(26000, 17)
    for prefix in ([4128, 6364], [4128, 6364, 151704]):
        trace = []
        toks = oracle.generate.transcribe_tokens(audio, 16000, w, enc_cfg, dec_cfg, max_tokens=8, lang_ids=prefix, trace=trace)
        ref = np.stack(trace)
        forced = np.zeros((1, 8), dtype=np.int32)
        forced[0, :len(toks)] = toks
        out = m.transcribe_batch([audio], 16000, max_tokens=8, lang_ids=[prefix], forced_tokens=forced, tap_steps=8)
        check_rel(f"logits, forced language prefix of {len(prefix)}", out.taps["logits"][0, :len(toks)], ref[:len(toks)], LOGIT_RTOL)
        check_tokens(f"tokens, forced language prefix of {len(prefix)}", out.tokens[0], ref[:len(toks)])
    # a batch mixing forced and free utterances keeps them apart
    mixed = m.transcribe_batch([audio, audio], 16000, max_tokens=6, lang_ids=[[4128, 6364], None], stop_rules=False)
    alone = m.transcribe_batch([audio], 16000, max_tokens=6, stop_rules=False)
    assert mixed.tokens[1] == alone.tokens[0]


def test_long_audio_modes(pinned_model):
    """A 33 s recording: "truncate" keeps the reference's behaviour (3000 frames -> 390 audio tokens, the clamp's
    maximum taken over the whole recording, AudioPreprocessing.swift:269-316); "segment_fixed" runs 30 s + 3 s pieces
    as independent utterances and joins them."""
    m, cfg, ck = pinned_model
    audio = synth.synth_audio(16000 * 33, 3)
    w = synth.checkpoint_to_f32(ck)
    enc_cfg, _ = oracle_cfgs(cfg)
    out = m.transcribe_batch([audio], 16000, max_tokens=4, tap_embeds=True)
    assert int(out.n_audio_tokens[0]) == 390
    ref = oracle.encoder.encode(oracle.mel.process(audio, 16000), w, enc_cfg)
    assert ref.shape[0] == 390
    check_rel("audio embeds, 33 s recording truncated", out.taps["audio_embeds"][:390], ref, ENC_RTOL)
    joined, pieces = m.transcribe_long([audio], 16000, mode="truncate", max_tokens=4)
    assert pieces[0] == [out.tokens[0]]
    joined, pieces = m.transcribe_long([audio, audio[:16000 * 4]], 16000, mode="segment_fixed", max_tokens=4)
    a = m.transcribe_batch([audio[:16000 * 30]], 16000, max_tokens=4).tokens[0]
    b = m.transcribe_batch([audio[16000 * 30:]], 16000, max_tokens=4).tokens[0]
    c = m.transcribe_batch([audio[:16000 * 4]], 16000, max_tokens=4).tokens[0]
    assert pieces == [[a, b], [c]]
    strip = lambda t: [x for x in t if x not in (151645, 151643)]
    assert joined == [strip(a) + strip(b), strip(c)]


def test_greedy_free_running_matches_oracle_prefix(tiny_model):
    """Free-running greedy decode: identical ids up to the first step whose oracle margin is within tolerance."""
    m, cfg, ck = tiny_model
    audio = synth.synth_audio(30000, 99)
    toks, ref_logits, _ = _oracle_run(cfg, ck, audio, 16)
    out = m.transcribe_batch([audio], 16000, max_tokens=16)
    got = out.tokens[0]
    _, mar = parity.margins(ref_logits)
    n = min(len(toks), len(got))
    below = np.nonzero(mar[:n] <= parity.TOKEN_GATE)[0]
    upto = int(below[0]) if below.size else n
    parity.note("free-running greedy prefix", f"{upto} of {n} steps before the first margin below the gate")
    assert got[:upto] == toks[:upto]


@pytest.mark.parametrize("preset", ["tiny", "1.7b-2layers"])
def test_last_prefill_layer_for_the_last_row_only(preset, monkeypatch):
    """The prefill's last layer runs attention, o and the MLP for the last prompt row of every utterance only (nothing
    else of that layer is ever read: Qwen3ASR.swift:386-388 takes the last position's logits; its K / V rows are cached
    as before).  Computing every row (Q3ASR_PREFILL_FULL_LAST=1) must give the same bits: first-step logits, and the
    tokens of a free-running decode over the cache the prefill left."""
    if preset == "tiny":
        cfg = make_config("tiny")
        lens = (16000, 52345, 480000, 30001, 99999)
    else:
        cfg = make_config("1.7b", enc_layers=2, layers=2)
        lens = (20000, 64000, 33333)
    ck = synth.make_checkpoint(cfg, seed=17)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        audios = [synth.synth_audio(n, 310 + i) for i, n in enumerate(lens)]
        short = m.transcribe_batch(audios, 16000, max_tokens=10, stop_rules=False, tap_steps=2)
        short_free = m.transcribe_batch(audios, 16000, max_tokens=10, stop_rules=False)
        prefix = m.transcribe_batch(audios[:2], 16000, max_tokens=4, lang_ids=[[4128, 6364], None], stop_rules=False, tap_steps=1)
        monkeypatch.setenv("Q3ASR_PREFILL_FULL_LAST", "1")
        full = m.transcribe_batch(audios, 16000, max_tokens=10, stop_rules=False, tap_steps=2)
        full_free = m.transcribe_batch(audios, 16000, max_tokens=10, stop_rules=False)
        prefix_full = m.transcribe_batch(audios[:2], 16000, max_tokens=4, lang_ids=[[4128, 6364], None], stop_rules=False, tap_steps=1)
        assert np.array_equal(short.taps["logits"], full.taps["logits"])
        assert short.tokens == full.tokens and short_free.tokens == full_free.tokens
        assert np.array_equal(prefix.taps["logits"], prefix_full.taps["logits"]) and prefix.tokens == prefix_full.tokens
    finally:
        m.close()
