"""Oracle: audio encoder (test infrastructure, numpy fp32).

Follows Sources/Qwen3ASR/AudioEncoder.swift of the reference.  Weights are a
dict keyed by the checkpoint names the reference loader reads
(Sources/Qwen3ASR/WeightLoading.swift:92-127, 270-347), fp32 arrays whose values
are bf16-representable; conv weights are in the checkpoint's MLX layout
[out, kH, kW, in] (WeightLoading.swift:277-284).
"""
import numpy as np
from scipy.special import erf

f32 = np.float32


def gelu(x):
    """Exact (erf) GELU, as MLXNN.gelu used at AudioEncoder.swift:128,380-384,492."""
    x = x.astype(f32)
    return (f32(0.5) * x * (f32(1.0) + erf(x / f32(np.sqrt(2.0))).astype(f32))).astype(f32)


def layer_norm(x, w, b, eps):
    mu = x.mean(axis=-1, keepdims=True, dtype=f32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=f32)
    return (xc / np.sqrt(var + f32(eps)) * w + b).astype(f32)


def conv_len(n):
    """One 3x3 stride-2 pad-1 conv: (n-1)/2+1 -- AudioEncoder.swift:259-261."""
    return (n - 1) // 2 + 1


def feat_len(clen):
    """Valid tokens of a chunk of clen frames -- AudioEncoder.swift:416-423."""
    return conv_len(conv_len(conv_len(clen)))


def output_length(t, chunk=100):
    """getOutputLength -- AudioEncoder.swift:254-270."""
    rem = t % chunk
    full = (t // chunk) * 13
    return full + (max(feat_len(rem), 1) if rem > 0 else 0)


def chunk_lengths(t, chunk=100):
    """AudioEncoder.swift:335-346."""
    n = (t + chunk - 1) // chunk
    out = []
    for i in range(n):
        if i == n - 1:
            rem = t % chunk
            out.append(chunk if rem == 0 else rem)
        else:
            out.append(chunk)
    return out


def window_cu_seqlens(t, cfg):
    """cu_seqlens of the attention windows -- AudioEncoder.swift:441-466."""
    chunk = cfg["n_window"] * 2
    clens = chunk_lengths(t, chunk)
    max_len = max(feat_len(c) for c in clens)
    window = max_len * (cfg["n_window_infer"] // chunk)
    total = output_length(t, chunk)
    lens = [window] * (total // window)
    if total % window:
        lens.append(total % window)
    return [0] + list(np.cumsum(lens))


def conv2d_nhwc(x, w, b, stride=2, pad=1):
    """x [B,H,W,Cin], w [Cout,kH,kW,Cin] (MLX layout) -> [B,Ho,Wo,Cout]."""
    bsz, h, wd, cin = x.shape
    cout, kh, kw, _ = w.shape
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    xp = np.zeros((bsz, h + 2 * pad, wd + 2 * pad, cin), dtype=f32)
    xp[:, pad:pad + h, pad:pad + wd, :] = x
    cols = np.empty((bsz, ho, wo, kh, kw, cin), dtype=f32)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + stride * ho:stride, j:j + stride * wo:stride, :]
    y = cols.reshape(bsz * ho * wo, kh * kw * cin) @ w.reshape(cout, -1).T
    return (y.reshape(bsz, ho, wo, cout) + b).astype(f32)


def sinusoid_pe(seq_len, d_model):
    """concat[sin, cos] -- AudioEncoder.swift:141-169."""
    half = d_model // 2
    inc = f32(np.log(f32(10000.0))) / f32(half - 1)
    inv = np.exp(np.arange(half, dtype=f32) * (-inc)).astype(f32)
    st = np.arange(seq_len, dtype=f32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(f32)


def frontend(mel, w, cfg):
    """Chunking, conv stack, conv_out, per-chunk PE, valid-token packing
    -- AudioEncoder.swift:325-437.  mel: [128, T].  Returns [N, d_model]."""
    p = "audio_tower."
    chunk = cfg["n_window"] * 2
    t = mel.shape[1]
    clens = chunk_lengths(t, chunk)
    max_len = max(clens)
    x = np.zeros((len(clens), mel.shape[0], max_len), dtype=f32)
    pos = 0
    for i, c in enumerate(clens):                      # :351-368 (pad to the longest chunk)
        x[i, :, :c] = mel[:, pos:pos + c]
        pos += c
    x = x[..., None]                                    # NHWC, :374
    for name in ("conv2d1", "conv2d2", "conv2d3"):      # :379-384
        x = gelu(conv2d_nhwc(x, w[p + name + ".weight"], w[p + name + ".bias"]))
    nb, fr, tc, ch = x.shape
    x = x.transpose(0, 2, 3, 1).reshape(nb, tc, ch * fr)  # channel-major flatten, :395-396
    x = (x @ w[p + "conv_out.weight"].T).astype(f32)       # no bias, :401
    x = x + sinusoid_pe(tc, cfg["d_model"])[None]          # :407-408
    return np.concatenate([x[i, :feat_len(c)] for i, c in enumerate(clens)], axis=0)  # :416-436


def layer(h, w, pre, cu, cfg):
    """AudioEncoderLayer + AudioSelfAttention -- AudioEncoder.swift:70-95,117-133.
    The block-diagonal -1e9 mask (:303-320) is applied literally."""
    nh = cfg["heads"]
    d = cfg["d_model"]
    hd = d // nh
    n = h.shape[0]
    x = layer_norm(h, w[pre + "self_attn_layer_norm.weight"], w[pre + "self_attn_layer_norm.bias"], cfg["ln_eps"])
    q = x @ w[pre + "self_attn.q_proj.weight"].T + w[pre + "self_attn.q_proj.bias"]
    k = x @ w[pre + "self_attn.k_proj.weight"].T + w[pre + "self_attn.k_proj.bias"]
    v = x @ w[pre + "self_attn.v_proj.weight"].T + w[pre + "self_attn.v_proj.bias"]
    q = q.reshape(n, nh, hd).transpose(1, 0, 2)
    k = k.reshape(n, nh, hd).transpose(1, 0, 2)
    v = v.reshape(n, nh, hd).transpose(1, 0, 2)
    mask = np.full((n, n), f32(-1e9), dtype=f32)
    for a, b in zip(cu[:-1], cu[1:]):
        mask[a:b, a:b] = 0
    s = (q @ k.transpose(0, 2, 1)) * f32(1.0 / np.sqrt(hd)) + mask[None]
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s).astype(f32)
    a = e / e.sum(axis=-1, keepdims=True, dtype=f32)
    o = (a @ v).transpose(1, 0, 2).reshape(n, d)
    o = o @ w[pre + "self_attn.out_proj.weight"].T + w[pre + "self_attn.out_proj.bias"]
    h = (h + o).astype(f32)
    x = layer_norm(h, w[pre + "final_layer_norm.weight"], w[pre + "final_layer_norm.bias"], cfg["ln_eps"])
    x = gelu(x @ w[pre + "fc1.weight"].T + w[pre + "fc1.bias"])
    x = x @ w[pre + "fc2.weight"].T + w[pre + "fc2.bias"]
    return (h + x).astype(f32)


def encode(mel, w, cfg, return_pre_proj=False):
    """Qwen3AudioEncoder.callAsFunction -- AudioEncoder.swift:325-501.
    mel [128, T] fp32 -> [N, out_dim] fp32."""
    p = "audio_tower."
    h = frontend(mel, w, cfg)
    cu = window_cu_seqlens(mel.shape[1], cfg)
    for i in range(cfg["layers"]):                      # :475-477
        h = layer(h, w, f"{p}layers.{i}.", cu, cfg)
    h = layer_norm(h, w[p + "ln_post.weight"], w[p + "ln_post.bias"], cfg["ln_eps"])  # :483
    if return_pre_proj:
        return h
    h = gelu(h @ w[p + "proj1.weight"].T + w[p + "proj1.bias"])                       # :491-492
    return (h @ w[p + "proj2.weight"].T + w[p + "proj2.bias"]).astype(f32)            # :493
