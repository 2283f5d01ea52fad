"""Oracle: MLX affine 4-bit group quantisation (test infrastructure).

The arithmetic lives in mlx-swift (>= 0.29.1, un-vendored dependency of the
reference, Package.swift:21; no version is pinned because Package.resolved is
git-ignored).  The published format, as used at
Sources/Qwen3ASR/QuantizedTextDecoder.swift:44-67 (group 64, 4 bits, affine):
``w[o, i] = q[o, i] * scales[o, i // g] + biases[o, i // g]`` with eight 4-bit
codes per uint32 along the input dimension, lowest nibble first.
"""
import numpy as np

f32 = np.float32


def bf16_round(x):
    """Round fp32 to the nearest bf16 (ties to even), return fp32."""
    u = np.ascontiguousarray(x, dtype=f32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(f32).reshape(np.shape(x))


def to_bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=f32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).reshape(np.shape(x))


def from_bf16_bits(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(f32)


def unpack(packed):
    """uint32 [out, in/8] -> uint8 codes [out, in] (low nibble first)."""
    packed = np.asarray(packed, dtype=np.uint32)
    shifts = (np.arange(8, dtype=np.uint32) * 4)[None, None, :]
    q = (packed[:, :, None] >> shifts) & np.uint32(0xF)
    return q.reshape(packed.shape[0], -1).astype(np.uint8)


def pack(codes):
    """uint8 codes [out, in] -> uint32 [out, in/8]."""
    codes = np.asarray(codes, dtype=np.uint32).reshape(codes.shape[0], -1, 8)
    shifts = (np.arange(8, dtype=np.uint32) * 4)[None, None, :]
    return np.bitwise_or.reduce(codes << shifts, axis=2).astype(np.uint32)


def dequantize(packed, scales, biases, group=64, round_bf16=False):
    """fp32 [out, in].  scales / biases: fp32 arrays with bf16-representable
    values.  round_bf16=True models ``dequantized(...)`` whose output dtype is
    the scales' dtype (the embedding gather, QuantizedTextDecoder.swift:44-51);
    False models the fused ``quantizedMM`` which keeps q*s+b in fp32 (:54-67)."""
    q = unpack(packed).astype(f32)
    s = np.repeat(np.asarray(scales, dtype=f32), group, axis=1)
    b = np.repeat(np.asarray(biases, dtype=f32), group, axis=1)
    w = (q * s + b).astype(f32)
    return bf16_round(w) if round_bf16 else w


def quantize(w, group=64):
    """Affine min/max quantiser used only to mint synthetic q4 checkpoints
    (any (codes, scale, bias) triple is a valid checkpoint; parity concerns the
    dequantisation above).  Returns (packed u32, scales fp32-bf16, biases fp32-bf16)."""
    w = np.asarray(w, dtype=f32)
    o, i = w.shape
    g = w.reshape(o, i // group, group)
    lo = g.min(axis=2)
    hi = g.max(axis=2)
    scale = bf16_round(np.maximum((hi - lo) / f32(15.0), f32(1e-8)))
    bias = bf16_round(lo)
    q = np.clip(np.rint((g - bias[:, :, None]) / scale[:, :, None]), 0, 15).astype(np.uint8)
    return pack(q.reshape(o, i)), scale.astype(f32), bias.astype(f32)
