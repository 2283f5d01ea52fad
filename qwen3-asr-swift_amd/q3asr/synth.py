"""Seeded synthetic checkpoints and synthetic audio.

No model weights exist offline, so throughput runs and parity tests use random
weights of the true architecture, named and laid out exactly like the tensors
the reference loader reads (WeightLoading.swift:92-127, 151-176, 243-347): conv
weights in MLX layout [out, kH, kW, in], dense tensors bf16, MLX-q4 linears as
(weight u32 [out, in/8], scales bf16 [out, in/64], biases bf16 [out, in/64]).

A checkpoint is a dict ``name -> (numpy array, dtype tag)`` with tags "bf16"
(array holds the uint16 bit patterns), "u32" and "f32".
"""
import numpy as np

f32 = np.float32


def to_bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=f32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def from_bf16_bits(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(f32)


def _quantize_q4(w, group):
    o, i = w.shape
    g = w.reshape(o, i // group, group)
    lo, hi = g.min(axis=2), g.max(axis=2)
    scale = from_bf16_bits(to_bf16_bits(np.maximum((hi - lo) / f32(15.0), f32(1e-8))))
    bias = from_bf16_bits(to_bf16_bits(lo))
    q = np.clip(np.rint((g - bias[:, :, None]) / scale[:, :, None]), 0, 15).astype(np.uint32).reshape(o, i // 8, 8)
    sh = (np.arange(8, dtype=np.uint32) * 4)[None, None, :]
    packed = np.bitwise_or.reduce(q << sh, axis=2).astype(np.uint32)
    return packed, to_bf16_bits(scale), to_bf16_bits(bias)


def make_checkpoint(cfg, seed=0, std=0.02, q4=None, workers=0):
    """Random checkpoint for a Q3Config.  q4 defaults to cfg.weight_format == 1.  workers > 0: the tensors are drawn
    and converted on that many threads from per-tensor generators (other values than the sequential default, same
    distributions: for benchmarks, where drawing 1.7 billion parameters on one core takes most of a minute)."""
    if workers > 0:
        return _make_checkpoint_threaded(cfg, seed, std, q4, workers)
    rng = np.random.default_rng(seed)
    q4 = bool(cfg.weight_format) if q4 is None else q4
    ck = {}

    def dense(name, shape, s=std):
        ck[name] = (to_bf16_bits(rng.standard_normal(shape, dtype=f32) * f32(s)), "bf16")

    def norm_w(name, n):
        ck[name] = (to_bf16_bits(f32(1.0) + rng.standard_normal(n, dtype=f32) * f32(0.05)), "bf16")

    def linear(name, out, inp, bias, quant=False, s=std):
        w = rng.standard_normal((out, inp), dtype=f32) * f32(s)
        if quant:
            p, sc, bi = _quantize_q4(w, cfg.group_size)
            ck[name + ".weight"], ck[name + ".scales"], ck[name + ".biases"] = (p, "u32"), (sc, "bf16"), (bi, "bf16")
        else:
            ck[name + ".weight"] = (to_bf16_bits(w), "bf16")
        if bias:
            dense(name + ".bias", (out,))

    a, d, c = "audio_tower.", cfg.enc_d_model, cfg.conv_channels
    f3 = ((cfg.n_mels + 1) // 2 + 1) // 2
    f3 = (f3 + 1) // 2                                  # mel bins after three stride-2 convs
    dense(a + "conv2d1.weight", (c, 3, 3, 1), 0.3)
    dense(a + "conv2d1.bias", (c,))
    for n in ("conv2d2", "conv2d3"):
        dense(a + n + ".weight", (c, 3, 3, c), 1.0 / np.sqrt(9 * c))
        dense(a + n + ".bias", (c,))
    dense(a + "conv_out.weight", (d, c * f3), 1.0 / np.sqrt(c * f3))
    for i in range(cfg.enc_layers):
        p = f"{a}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            linear(p + "self_attn." + n, d, d, True, s=1.0 / np.sqrt(d))
        linear(p + "fc1", cfg.enc_ffn, d, True, s=1.0 / np.sqrt(d))
        linear(p + "fc2", d, cfg.enc_ffn, True, s=1.0 / np.sqrt(cfg.enc_ffn))
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            norm_w(p + n + ".weight", d)
            dense(p + n + ".bias", (d,))
    norm_w(a + "ln_post.weight", d)
    dense(a + "ln_post.bias", (d,))
    linear(a + "proj1", d, d, True, s=1.0 / np.sqrt(d))
    linear(a + "proj2", cfg.enc_out_dim, d, True, s=1.0 / np.sqrt(d))

    h, hd = cfg.hidden, cfg.head_dim
    linear("model.embed_tokens", cfg.vocab, h, False, quant=q4)
    norm_w("model.norm.weight", h)
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        linear(p + "self_attn.q_proj", cfg.heads * hd, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "self_attn.k_proj", cfg.kv_heads * hd, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "self_attn.v_proj", cfg.kv_heads * hd, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "self_attn.o_proj", h, cfg.heads * hd, False, q4, 1.0 / np.sqrt(cfg.heads * hd))
        norm_w(p + "self_attn.q_norm.weight", hd)
        norm_w(p + "self_attn.k_norm.weight", hd)
        norm_w(p + "input_layernorm.weight", h)
        norm_w(p + "post_attention_layernorm.weight", h)
        linear(p + "mlp.gate_proj", cfg.inter, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "mlp.up_proj", cfg.inter, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "mlp.down_proj", h, cfg.inter, False, q4, 1.0 / np.sqrt(cfg.inter))
    return ck


def _make_checkpoint_threaded(cfg, seed, std, q4, workers):
    """make_checkpoint with every tensor's draw, scaling and conversion as one task on a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    q4 = bool(cfg.weight_format) if q4 is None else q4
    tasks = []          # (names, function of a generator -> tuple of (array, tag))

    def dense(name, shape, s=std):
        tasks.append(((name,), lambda g: ((to_bf16_bits(g.standard_normal(shape, dtype=f32) * f32(s)), "bf16"),)))

    def norm_w(name, n):
        tasks.append(((name,), lambda g: ((to_bf16_bits(f32(1.0) + g.standard_normal(n, dtype=f32) * f32(0.05)), "bf16"),)))

    def linear(name, out, inp, bias, quant=False, s=std):
        if quant:
            def job(g):
                p, sc, bi = _quantize_q4(g.standard_normal((out, inp), dtype=f32) * f32(s), cfg.group_size)
                return (p, "u32"), (sc, "bf16"), (bi, "bf16")
            tasks.append(((name + ".weight", name + ".scales", name + ".biases"), job))
        else:
            dense(name + ".weight", (out, inp), s)
        if bias:
            dense(name + ".bias", (out,))

    a, d, c = "audio_tower.", cfg.enc_d_model, cfg.conv_channels
    f3 = (((cfg.n_mels + 1) // 2 + 1) // 2 + 1) // 2
    dense(a + "conv2d1.weight", (c, 3, 3, 1), 0.3)
    dense(a + "conv2d1.bias", (c,))
    for n in ("conv2d2", "conv2d3"):
        dense(a + n + ".weight", (c, 3, 3, c), 1.0 / np.sqrt(9 * c))
        dense(a + n + ".bias", (c,))
    dense(a + "conv_out.weight", (d, c * f3), 1.0 / np.sqrt(c * f3))
    for i in range(cfg.enc_layers):
        p = f"{a}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            linear(p + "self_attn." + n, d, d, True, s=1.0 / np.sqrt(d))
        linear(p + "fc1", cfg.enc_ffn, d, True, s=1.0 / np.sqrt(d))
        linear(p + "fc2", d, cfg.enc_ffn, True, s=1.0 / np.sqrt(cfg.enc_ffn))
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            norm_w(p + n + ".weight", d)
            dense(p + n + ".bias", (d,))
    norm_w(a + "ln_post.weight", d)
    dense(a + "ln_post.bias", (d,))
    linear(a + "proj1", d, d, True, s=1.0 / np.sqrt(d))
    linear(a + "proj2", cfg.enc_out_dim, d, True, s=1.0 / np.sqrt(d))
    h, hd = cfg.hidden, cfg.head_dim
    # the embedding is the largest tensor by far: drawn in row blocks so that it does not end up as one thread's task
    eblocks = [(r, min(r + 8192, cfg.vocab)) for r in range(0, cfg.vocab, 8192)]
    if q4:
        linear("model.embed_tokens", cfg.vocab, h, False, quant=True)
    else:
        for r0, r1 in eblocks:
            dense(f"model.embed_tokens.weight#{r0}", (r1 - r0, h))
    norm_w("model.norm.weight", h)
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        linear(p + "self_attn.q_proj", cfg.heads * hd, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "self_attn.k_proj", cfg.kv_heads * hd, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "self_attn.v_proj", cfg.kv_heads * hd, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "self_attn.o_proj", h, cfg.heads * hd, False, q4, 1.0 / np.sqrt(cfg.heads * hd))
        norm_w(p + "self_attn.q_norm.weight", hd)
        norm_w(p + "self_attn.k_norm.weight", hd)
        norm_w(p + "input_layernorm.weight", h)
        norm_w(p + "post_attention_layernorm.weight", h)
        linear(p + "mlp.gate_proj", cfg.inter, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "mlp.up_proj", cfg.inter, h, False, q4, 1.0 / np.sqrt(h))
        linear(p + "mlp.down_proj", h, cfg.inter, False, q4, 1.0 / np.sqrt(cfg.inter))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        futs = [pool.submit(job, np.random.default_rng((seed, k))) for k, (_, job) in enumerate(tasks)]
        ck = {}
        for (names, _), fut in zip(tasks, futs):
            for name, val in zip(names, fut.result()):
                ck[name] = val
    if not q4:
        ck["model.embed_tokens.weight"] = (np.concatenate([ck.pop(f"model.embed_tokens.weight#{r0}")[0] for r0, _ in eblocks]), "bf16")
    return ck


def checkpoint_to_f32(ck):
    """name -> fp32 array (bf16 decoded) / uint32 array, the form the numpy
    restatement of the reference takes."""
    out = {}
    for k, (arr, tag) in ck.items():
        out[k] = from_bf16_bits(arr) if tag == "bf16" else arr
    return out


def synth_audio(n_samples, seed, kind="speechlike"):
    """Seeded 16 kHz mono fp32 clip in [-1, 1].  "speechlike": 3-5 harmonics of
    f0 in [90, 250] Hz with 4 Hz amplitude modulation plus white noise
    (sigma 0.01), peak ~0.3.  "noise": white noise, sigma 0.1."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return np.clip(rng.standard_normal(n_samples, dtype=f32) * f32(0.1), -1, 1).astype(f32)
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    f0 = rng.uniform(90.0, 250.0)
    nh = int(rng.integers(3, 6))
    x = np.zeros(n_samples, dtype=np.float64)
    for k in range(1, nh + 1):
        x += rng.uniform(0.3, 1.0) / k * np.sin(2 * np.pi * f0 * k * t + rng.uniform(0, 2 * np.pi))
    x *= 0.5 * (1.0 + 0.8 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi)))
    x = 0.3 * x / max(np.abs(x).max(), 1e-9)
    x += rng.standard_normal(n_samples) * 0.01
    return np.clip(x, -1, 1).astype(f32)
