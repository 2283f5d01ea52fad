"""GPU tests of the boundary's behaviour around the hot path: loud failures (the reference silently skips missing
weights, WeightLoading.swift:190-203; this library must not), a model that stays usable after a rejected call, and
loading from a checkpoint directory as `fromPretrained` does (Qwen3ASR.swift:693-770)."""
import json
import os

import numpy as np
import pytest

from q3asr import Qwen3ASRModel, make_config, synth
from q3asr.lib import Q3Error
from q3asr.safetensors_io import save_file

pytestmark = pytest.mark.gpu


def test_missing_or_misshaped_weights_fail_at_load():
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=1)
    m = Qwen3ASRModel(cfg)
    try:
        for name, (arr, tag) in ck.items():
            if name != "model.layers.1.mlp.down_proj.weight":
                m.load_tensor(name, arr, tag)
        with pytest.raises(Q3Error, match="model.layers.1.mlp.down_proj.weight"):
            m.load_checkpoint({})                         # finalize: the missing key is named
    finally:
        m.close()
    m = Qwen3ASRModel(cfg)
    try:
        bad = dict(ck)
        arr, tag = bad["audio_tower.layers.0.fc1.weight"]
        bad["audio_tower.layers.0.fc1.weight"] = (np.ascontiguousarray(arr[:, :-8]), tag)
        with pytest.raises(Q3Error):
            m.load_checkpoint(bad)
        with pytest.raises(Q3Error):                      # nothing runs on a model that never finalized
            m.transcribe_batch([synth.synth_audio(16000, 1)], 16000, max_tokens=2)
    finally:
        m.close()


def test_rejected_calls_leave_the_model_usable(tiny_model):
    m, cfg, _ = tiny_model
    audio = synth.synth_audio(20000, 9)
    good = m.transcribe_batch([audio], 16000, max_tokens=4, stop_rules=False).tokens
    with pytest.raises(Q3Error, match="utterance 1 is shorter than one"):
        m.transcribe_batch([audio, audio[:100]], 16000, max_tokens=4)
    with pytest.raises(Q3Error, match="max_tokens"):
        m.transcribe_batch([audio], 16000, max_tokens=0)
    with pytest.raises(Q3Error, match="max_tokens"):
        m.transcribe_batch([audio], 16000, max_tokens=5000)
    with pytest.raises(Q3Error, match="sample rate"):
        m.transcribe_batch([audio], 0, max_tokens=4)
    with pytest.raises(Q3Error, match="language prefix"):
        m.transcribe_batch([audio], 16000, max_tokens=4, lang_ids=[[cfg.vocab + 5]])
    with pytest.raises(Q3Error, match="multiple of 32"):
        m.op_gemm(np.zeros((4, 40), np.uint16), np.zeros((8, 40), np.uint16))
    with pytest.raises(Q3Error, match="forced token id out of range"):
        m.transcribe_batch([audio], 16000, max_tokens=4, forced_tokens=np.full((1, 4), 1 << 30, np.int32))
    assert m.transcribe_batch([audio], 16000, max_tokens=4, stop_rules=False).tokens == good


def test_from_pretrained_reads_rope_scaling(tmp_path):
    """rope_scaling.interleaved = false and an mrope_section that rotates only the first 64 channels
    (Qwen3ASR.swift:806-810, Configuration.swift:75-80) must reach the kernels: logits against the CPU restatement
    run with the same switches, and different from the default (interleaved, full rotary) model's."""
    import oracle
    from q3asr import oracle_cfgs
    cfg = make_config("tiny", rope_interleaved=0, rotary_dim=64)
    ck = synth.make_checkpoint(cfg, seed=4)
    save_file(tmp_path / "model.safetensors", ck)
    config = {"thinker_config": {
        "text_config": dict(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                            num_key_value_heads=cfg.kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.inter,
                            vocab_size=cfg.vocab, rms_norm_eps=1e-6, rope_theta=1e6,
                            rope_scaling=dict(interleaved=False, mrope_interleaved=True, mrope_section=[12, 10, 10])),
        "audio_config": dict(d_model=cfg.enc_d_model, encoder_layers=cfg.enc_layers, encoder_attention_heads=cfg.enc_heads,
                             encoder_ffn_dim=cfg.enc_ffn, output_dim=cfg.enc_out_dim, downsample_hidden_size=cfg.conv_channels,
                             num_mel_bins=128, n_window=50, n_window_infer=800)}}
    (tmp_path / "config.json").write_text(json.dumps(config))
    audio = synth.synth_audio(30000, 12)
    enc_cfg, dec_cfg = oracle_cfgs(cfg)
    assert dec_cfg["rotary_dim"] == 64 and not dec_cfg["rope_interleaved"]
    trace = []
    toks = oracle.generate.transcribe_tokens(audio, 16000, synth.checkpoint_to_f32(ck), enc_cfg, dec_cfg, max_tokens=5, trace=trace)
    ref = np.stack(trace)
    forced = np.zeros((1, 5), dtype=np.int32)
    forced[0, :len(toks)] = toks
    m = Qwen3ASRModel.from_pretrained(str(tmp_path))
    plain = Qwen3ASRModel.from_checkpoint(make_config("tiny"), ck)
    try:
        assert m.cfg.rope_interleaved == 0 and m.cfg.rotary_dim == 64
        got = m.transcribe_batch([audio], 16000, max_tokens=5, forced_tokens=forced, tap_steps=5).taps["logits"][0, :len(toks)]
        other = plain.transcribe_batch([audio], 16000, max_tokens=5, forced_tokens=forced, tap_steps=5).taps["logits"][0, :len(toks)]
        rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
        assert rel(got, ref[:len(toks)]) <= 4e-2
        assert rel(other, ref[:len(toks)]) > rel(got, ref[:len(toks)])
    finally:
        m.close()
        plain.close()
    with pytest.raises(Q3Error, match="group size 64"):
        Qwen3ASRModel(make_config("tiny", weight_format=1, group_size=32))


def test_from_pretrained_directory(tmp_path):
    """config.json + model.safetensors on disk -> the same transcription as the in-memory checkpoint."""
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=4)
    save_file(tmp_path / "model.safetensors", ck)
    config = {"thinker_config": {
        "text_config": dict(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                            num_key_value_heads=cfg.kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.inter,
                            vocab_size=cfg.vocab, rms_norm_eps=1e-6, rope_theta=1e6),
        "audio_config": dict(d_model=cfg.enc_d_model, encoder_layers=cfg.enc_layers, encoder_attention_heads=cfg.enc_heads,
                             encoder_ffn_dim=cfg.enc_ffn, output_dim=cfg.enc_out_dim, downsample_hidden_size=cfg.conv_channels,
                             num_mel_bins=128, n_window=50, n_window_infer=800)}}
    (tmp_path / "config.json").write_text(json.dumps(config))
    audio = synth.synth_audio(30000, 12)
    a = Qwen3ASRModel.from_checkpoint(cfg, ck)
    b = Qwen3ASRModel.from_pretrained(str(tmp_path))
    try:
        ta = a.transcribe_batch([audio], 16000, max_tokens=8, stop_rules=False)
        tb = b.transcribe_batch([audio], 16000, max_tokens=8, stop_rules=False)
        assert ta.tokens == tb.tokens
        assert b.tokenizer is None and isinstance(b.transcribe(audio, 16000, max_tokens=3), str)   # ids joined by spaces (:647-650)
    finally:
        a.close()
        b.close()


def test_cli_transcribes_wav_files_from_a_model_directory(tmp_path):
    """scripts/transcribe.py (the reference CLI's `transcribe`, main.swift:297-315): model directory + WAV files in,
    one line per file out; equals the in-process result on the reference's own 24 kHz test clip."""
    import subprocess
    import sys
    from conftest import GOLDEN, read_wav_pcm16
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=4)
    save_file(tmp_path / "model.safetensors", ck)
    config = {"thinker_config": {
        "text_config": dict(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                            num_key_value_heads=cfg.kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.inter,
                            vocab_size=cfg.vocab, rms_norm_eps=1e-6, rope_theta=1e6),
        "audio_config": dict(d_model=cfg.enc_d_model, encoder_layers=cfg.enc_layers, encoder_attention_heads=cfg.enc_heads,
                             encoder_ffn_dim=cfg.enc_ffn, output_dim=cfg.enc_out_dim, downsample_hidden_size=cfg.conv_channels,
                             num_mel_bins=128, n_window=50, n_window_infer=800)}}
    (tmp_path / "config.json").write_text(json.dumps(config))
    wav = os.path.join(GOLDEN, "test_audio.wav")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "scripts", "transcribe.py"), str(tmp_path), wav, wav, "--max-tokens", "6"],
                         capture_output=True, text=True, timeout=240)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith(wav + ": ")]
    assert len(lines) == 2 and lines[0] == lines[1]
    audio, rate = read_wav_pcm16(wav)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck)
    try:
        want = m.transcribe_batch([audio], rate, max_tokens=6).tokens[0]
    finally:
        m.close()
    while want and want[-1] in (151645, 151643):      # the joined form drops a closing EOS
        want.pop()
    assert lines[0] == wav + ": " + " ".join(str(t) for t in want)


def test_two_host_threads_share_one_model(tiny_model):
    """The reference's model is called from detached tasks (Qwen3ASR.swift:685-687, RealtimeTranslation.swift:459-477).
    Calls on one handle are serialised by the library: two threads transcribing at once get what one thread gets."""
    import threading
    m, cfg, _ = tiny_model
    sets = [[synth.synth_audio(20000 + 9000 * i, 900 + 10 * t + i) for i in range(5)] for t in range(2)]
    want = [m.transcribe_batch(a, 16000, max_tokens=12, stop_rules=False).tokens for a in sets]
    mels = [m.feature_extractor.process(sets[t][0], 16000) for t in range(2)]
    got, errs = [[None] * 6, [None] * 6], []

    def work(t):
        try:
            for r in range(6):
                got[t][r] = m.transcribe_batch(sets[t], 16000, max_tokens=12, stop_rules=False).tokens
                assert np.array_equal(m.feature_extractor.process(sets[t][0], 16000), mels[t])
        except Exception as e:      # surfaced in the main thread below
            errs.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=240)
    assert not errs, errs
    for t in range(2):
        assert all(g == want[t] for g in got[t])


def test_workspace_stays_bounded_when_requests_grow(monkeypatch):
    """The KV cache is one large workspace request.  A later call that needs a slightly larger one must replace the
    old block, not keep it beside the new one (at the benchmark's size the two together would not fit the GPU)."""
    monkeypatch.setenv("Q3ASR_ARENA_BLOCK_MB", "1")        # every request above 1 MiB takes the large-request path
    cfg = make_config("tiny")
    m = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=1))
    try:
        audios = [synth.synth_audio(480000, 30 + i) for i in range(6)]     # 405 prompt rows each
        sizes = []
        for max_tokens in (8, 24, 40, 56, 72, 88, 104, 120):
            forced = np.full((6, max_tokens), 151645, dtype=np.int32)      # every clip stops at once: the cache is only sized
            m.transcribe_batch(audios, 16000, max_tokens=max_tokens, forced_tokens=forced, stop_rules="forced")
            sizes.append(m.workspace_bytes())
        again = m.transcribe_batch(audios, 16000, max_tokens=8, stop_rules=False).tokens
        assert m.workspace_bytes() <= sizes[-1]
        # without the fix every step keeps the previous K and V blocks: the footprint grows with the SUM of the requests
        k_or_v = lambda mt: (405 + mt + 16) * 128 * cfg.kv_heads * 6 * cfg.layers * 2      # bytes of one cache (bf16) at max_tokens
        assert sizes[-1] - sizes[0] <= 2 * (k_or_v(120) - k_or_v(8)) + (4 << 20), sizes
        fresh = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=1))
        try:
            assert fresh.transcribe_batch(audios, 16000, max_tokens=8, stop_rules=False).tokens == again
        finally:
            fresh.close()
    finally:
        m.close()
