"""Model configurations accepted by the C ABI.

Mirrors the reference's ``Qwen3AudioEncoderConfig`` (AudioEncoder.swift:7-43) and
``TextDecoderConfig`` (Configuration.swift:46-110).  "0.6b" is the reference's
only runnable shape (encoder ``.default``, decoder ``.small``); "1.7b" uses the
true Qwen3-ASR-1.7B dimensions (the reference's ``.large`` preset is not that
shape); "tiny" is a test-sized model that exercises every kernel.
"""
import ctypes


class Q3Config(ctypes.Structure):
    """Field-for-field image of ``q3asr_config`` in include/q3asr.h."""
    _fields_ = [
        # audio encoder
        ("enc_d_model", ctypes.c_int), ("enc_heads", ctypes.c_int), ("enc_ffn", ctypes.c_int),
        ("enc_layers", ctypes.c_int), ("n_mels", ctypes.c_int), ("enc_out_dim", ctypes.c_int),
        ("conv_channels", ctypes.c_int), ("n_window", ctypes.c_int), ("n_window_infer", ctypes.c_int),
        ("ln_eps", ctypes.c_float),
        # text decoder
        ("vocab", ctypes.c_int), ("hidden", ctypes.c_int), ("layers", ctypes.c_int), ("heads", ctypes.c_int),
        ("kv_heads", ctypes.c_int), ("head_dim", ctypes.c_int), ("inter", ctypes.c_int),
        ("rms_eps", ctypes.c_float), ("rope_theta", ctypes.c_float),
        ("rope_interleaved", ctypes.c_int), ("rotary_dim", ctypes.c_int),
        ("group_size", ctypes.c_int), ("bits", ctypes.c_int),
        # switches
        ("weight_format", ctypes.c_int),   # 0 = dense bf16, 1 = MLX affine q4 decoder
        ("embed_scale", ctypes.c_int),     # std(text)/std(audio) rescale, reference default on
        ("max_batch", ctypes.c_int),       # utterances decoded together on one GPU
    ]


_PRESETS = {
    "0.6b": dict(enc_d_model=896, enc_heads=14, enc_ffn=3584, enc_layers=18, enc_out_dim=1024,
                 hidden=1024, layers=28, heads=16, kv_heads=8, head_dim=128, inter=3072),
    "1.7b": dict(enc_d_model=1024, enc_heads=16, enc_ffn=4096, enc_layers=24, enc_out_dim=2048,
                 hidden=2048, layers=28, heads=16, kv_heads=8, head_dim=128, inter=6144),
    "tiny": dict(enc_d_model=128, enc_heads=2, enc_ffn=256, enc_layers=2, enc_out_dim=256, conv_channels=32,
                 hidden=256, layers=2, heads=4, kv_heads=2, head_dim=128, inter=512),
}


def make_config(preset="0.6b", **overrides):
    base = dict(n_mels=128, conv_channels=480, n_window=50, n_window_infer=800, ln_eps=1e-5,
                vocab=151936, rms_eps=1e-6, rope_theta=1e6, rope_interleaved=1, rotary_dim=0,
                group_size=64, bits=4, weight_format=0, embed_scale=1, max_batch=128)
    base.update(_PRESETS[preset])
    base.update(overrides)
    if not base["rotary_dim"]:
        base["rotary_dim"] = base["head_dim"]       # Configuration.swift:75-80 (no mrope_section)
    cfg = Q3Config()
    for k, v in base.items():
        setattr(cfg, k, v)
    return cfg


def oracle_cfgs(cfg):
    """The two dicts the numpy oracle takes, derived from a Q3Config."""
    enc = dict(d_model=cfg.enc_d_model, heads=cfg.enc_heads, ffn=cfg.enc_ffn, layers=cfg.enc_layers,
               out_dim=cfg.enc_out_dim, n_window=cfg.n_window, n_window_infer=cfg.n_window_infer,
               ln_eps=cfg.ln_eps, conv_channels=cfg.conv_channels)
    dec = dict(vocab=cfg.vocab, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads, kv_heads=cfg.kv_heads,
               head_dim=cfg.head_dim, inter=cfg.inter, rms_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
               rope_interleaved=bool(cfg.rope_interleaved), rotary_dim=cfg.rotary_dim, group_size=cfg.group_size)
    return enc, dec
