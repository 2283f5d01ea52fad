"""Oracle: Qwen3 text decoder (test infrastructure, numpy fp32).

Follows Sources/Qwen3ASR/QuantizedTextDecoder.swift (4-bit) and its dense twin
Sources/Qwen3ASR/TextDecoder.swift (the written spec of the bf16 variant).
Weights: dict keyed by checkpoint names (WeightLoading.swift:151-176,243-266);
a linear named X is dense when only ``X.weight`` (fp32, bf16-valued) exists and
MLX-q4 when ``X.scales`` / ``X.biases`` are present as well.
"""
import numpy as np
from . import q4

f32 = np.float32


class Linear:
    """Caches the fp32 [out, in] matrix of a (possibly quantised) linear."""

    def __init__(self, w, name, group=64):
        if name + ".scales" in w:
            self.m = q4.dequantize(w[name + ".weight"], w[name + ".scales"], w[name + ".biases"], group)
        else:
            self.m = np.asarray(w[name + ".weight"], dtype=f32)

    def __call__(self, x):
        return (x @ self.m.T).astype(f32)


def rms_norm(x, w, eps):
    """MLXNN.RMSNorm -- QuantizedTextDecoder.swift:124-125,312-313,358."""
    ms = (x * x).mean(axis=-1, keepdims=True, dtype=f32)
    return (x / np.sqrt(ms + f32(eps)) * w).astype(f32)


def rope(x, offset, cfg):
    """applyRoPE -- QuantizedTextDecoder.swift:211-268.  x: [heads, T, hd]."""
    hd = x.shape[-1]
    t = x.shape[1]
    rot = min(cfg["rotary_dim"], hd)
    half = rot // 2
    inv = np.exp(-np.arange(half, dtype=f32) * (np.log(f32(cfg["rope_theta"])) / f32(half))).astype(f32)  # :219-220
    pos = np.arange(offset, offset + t, dtype=f32)                                                    # :223
    ang = (pos[:, None] * inv[None, :]).astype(f32)
    c, s = np.cos(ang).astype(f32)[None], np.sin(ang).astype(f32)[None]
    xr = x[..., :rot]
    if cfg["rope_interleaved"]:                         # :245-252
        x0, x1 = xr[..., 0::2], xr[..., 1::2]
        r = np.empty_like(xr)
        r[..., 0::2] = x0 * c - x1 * s
        r[..., 1::2] = x1 * c + x0 * s
    else:                                               # :253-259
        x0, x1 = xr[..., :half], xr[..., half:]
        r = np.concatenate([x0 * c - x1 * s, x1 * c + x0 * s], axis=-1)
    return np.concatenate([r, x[..., rot:]], axis=-1).astype(f32)


class Decoder:
    """QuantizedTextModel / TextModel -- QuantizedTextDecoder.swift:340-446."""

    def __init__(self, w, cfg):
        self.cfg = cfg
        self.w = w
        g = cfg.get("group_size", 64)
        self.lin = {}
        for i in range(cfg["layers"]):
            pre = f"model.layers.{i}."
            for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                      "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
                self.lin[pre + n] = Linear(w, pre + n, g)
        self.quant_embed = "model.embed_tokens.scales" in w
        if self.quant_embed:
            e = "model.embed_tokens"
            # asLinear -> quantizedMM keeps q*s+b in fp32 (:54-67)
            self.head = q4.dequantize(w[e + ".weight"], w[e + ".scales"], w[e + ".biases"], g)
            # callAsFunction -> dequantized(...) in the scales' dtype, bf16 (:44-51)
            self.table = q4.bf16_round(self.head)
        else:
            self.head = self.table = np.asarray(w["model.embed_tokens.weight"], dtype=f32)

    def embed(self, ids):
        return self.table[np.asarray(ids, dtype=np.int64)].astype(f32)

    def logits(self, h):
        """Tied LM head -- PreQuantizedEmbedding.asLinear, :54-67."""
        return (h @ self.head.T).astype(f32)

    def forward(self, x, cache):
        """x [T, hidden] fp32, cache: list of (K [kvh, L, hd], V) or None.
        Returns (hidden [T, hidden], new cache) -- :384-446."""
        cfg = self.cfg
        nh, nkv, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
        t = x.shape[0]
        clen = 0 if cache is None else cache[0][0].shape[1]
        mask = np.triu(np.ones((t, t + clen), dtype=f32), k=clen + 1) * f32(-1e9)   # :364-381
        new_cache = []
        h = x.astype(f32)
        for i in range(cfg["layers"]):
            pre = f"model.layers.{i}."
            xn = rms_norm(h, self.w[pre + "input_layernorm.weight"], cfg["rms_eps"])         # :325
            q = self.lin[pre + "self_attn.q_proj"](xn).reshape(t, nh, hd)
            k = self.lin[pre + "self_attn.k_proj"](xn).reshape(t, nkv, hd)
            v = self.lin[pre + "self_attn.v_proj"](xn).reshape(t, nkv, hd)
            q = rms_norm(q, self.w[pre + "self_attn.q_norm.weight"], cfg["rms_eps"])           # :148-149
            k = rms_norm(k, self.w[pre + "self_attn.k_norm.weight"], cfg["rms_eps"])
            q, k, v = q.transpose(1, 0, 2), k.transpose(1, 0, 2), v.transpose(1, 0, 2)
            q, k = rope(q, clen, cfg), rope(k, clen, cfg)                                       # :157-160
            if cache is not None:                                                               # :166-169
                k = np.concatenate([cache[i][0], k], axis=1)
                v = np.concatenate([cache[i][1], v], axis=1)
            new_cache.append((k, v))
            rep = nh // nkv                                                                     # :171-185
            ke, ve = np.repeat(k, rep, axis=0), np.repeat(v, rep, axis=0)
            s = (q @ ke.transpose(0, 2, 1)) * f32(1.0 / np.sqrt(hd)) + mask[None]               # :188-192
            s = s - s.max(axis=-1, keepdims=True)
            e = np.exp(s).astype(f32)
            a = e / e.sum(axis=-1, keepdims=True, dtype=f32)
            o = (a @ ve).transpose(1, 0, 2).reshape(t, nh * hd)                                 # :201-204
            h = (h + self.lin[pre + "self_attn.o_proj"](o)).astype(f32)                         # :206,327
            xn = rms_norm(h, self.w[pre + "post_attention_layernorm.weight"], cfg["rms_eps"])   # :331
            g = self.lin[pre + "mlp.gate_proj"](xn)
            u = self.lin[pre + "mlp.up_proj"](xn)
            act = (g / (f32(1.0) + np.exp(-g))).astype(f32) * u                                 # silu(gate)*up, :294-298
            h = (h + self.lin[pre + "mlp.down_proj"](act)).astype(f32)                          # :333
        return rms_norm(h, self.w["model.norm.weight"], cfg["rms_eps"]), new_cache             # :439
