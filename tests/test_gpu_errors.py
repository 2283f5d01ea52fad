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
    with pytest.raises(Q3Error, match="shorter than one"):
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
