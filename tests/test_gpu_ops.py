"""GPU parity of the kernel-level entry points (GEMM variants, attention) through the C ABI."""
import numpy as np
import pytest

from conftest import bf16_bits, bf16_vals

pytestmark = pytest.mark.gpu


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


@pytest.mark.parametrize("M,N,K,tile,ksplit,gelu,bias", [
    (128, 128, 64, 0, 1, False, False),
    (300, 480, 4320, 0, 1, True, True),      # conv-shaped: N, K not multiples of the tile, K tail of 32
    (405, 2048, 1024, 0, 1, False, True),
    (7, 896, 896, 1, 1, False, True),        # ragged M on the 64x64 tile
    (130, 192, 96, 1, 1, True, False),
    (64, 1024, 3072, 1, 4, False, False),    # split-K slabs
    (200, 320, 512, 2, 1, False, True),      # 128x64 tile
    (1, 151936, 256, 0, 1, False, False),    # LM-head shape, one row
    (256, 256, 64, 3, 1, False, False),      # 256x256 tile: a single K tile
    (600, 520, 4192, 3, 1, True, True),      # ragged M and N, odd number of K tiles, K tail of 32
    (1000, 768, 128, 3, 1, False, True),
    (300, 1024, 1984, 3, 3, False, False),   # split-K slabs, uneven split
    (2100, 2304, 2048, 3, 1, False, False),  # more tiles than the XCD grouping width
])
def test_gemm_matches_fp64(tiny_model, M, N, K, tile, ksplit, gelu, bias):
    m = tiny_model[0]
    rng = np.random.default_rng(M * 7 + N)
    a = bf16_bits(rng.standard_normal((M, K)) * 0.5)
    w = bf16_bits(rng.standard_normal((N, K)) / np.sqrt(K))
    b = (rng.standard_normal(N) * 0.1).astype(np.float32) if bias else None
    out = m.op_gemm(a, w, b, gelu, tile, ksplit)
    ref = bf16_vals(a).astype(np.float64) @ bf16_vals(w).astype(np.float64).T
    if bias:
        ref = ref + b
    if gelu:
        ref = _gelu(ref)
    tol = 2e-5 * np.sqrt(K) + 1e-6       # fp32 accumulation of exact bf16 products
    assert np.abs(out - ref).max() <= tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,N,K,tile", [(2, 256, 128, 1), (1, 2048, 1024, 1), (1, 3072, 1024, 1), (96, 512, 1024, 0)])
def test_gemm_q4_matches_dequant_reference(tiny_model, M, N, K, tile):
    """Fused quantised GEMM vs dequantise + matmul on the shapes of Diagnostics.swift:50-105."""
    import oracle
    m = tiny_model[0]
    rng = np.random.default_rng(N + K)
    wq, s, b = oracle.q4.quantize(rng.standard_normal((N, K)).astype(np.float32) * 0.05)
    a = bf16_bits(rng.standard_normal((M, K)) * 0.5)
    out = m.op_gemm_q4(a, wq, bf16_bits(s), bf16_bits(b), tile)
    wd = oracle.q4.dequantize(wq, s, b, 64, round_bf16=True).astype(np.float64)   # the kernel feeds bf16 to the MFMA
    ref = bf16_vals(a).astype(np.float64) @ wd.T
    assert np.abs(out - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())
    exact = bf16_vals(a).astype(np.float64) @ oracle.q4.dequantize(wq, s, b, 64).astype(np.float64).T
    assert np.abs(out - exact).max() <= 2e-2 * max(1.0, np.abs(exact).max())       # bf16 rounding of q*s+b


def _ref_attention(q, k, v, cu, heads, group, hd, causal):
    rows = q.shape[0]
    kvh = heads // group
    q = q.reshape(rows, heads, hd).astype(np.float64)
    k = k.reshape(rows, kvh, hd).astype(np.float64)
    v = v.reshape(rows, kvh, hd).astype(np.float64)
    out = np.zeros((rows, heads, hd))
    for a, b in zip(cu[:-1], cu[1:]):
        for h in range(heads):
            s = q[a:b, h] @ k[a:b, h // group].T / np.sqrt(hd)
            if causal:
                s = s + np.triu(np.ones_like(s), 1) * -1e9
            s = s - s.max(axis=1, keepdims=True)
            p = np.exp(s)
            p /= p.sum(axis=1, keepdims=True)
            out[a:b, h] = p @ v[a:b, h // group]
    return out.reshape(rows, heads * hd)


@pytest.mark.parametrize("lens,heads,group,hd,causal", [
    ([104, 104, 104, 78], 14, 1, 64, False),     # encoder windows of a 30 s clip
    ([13, 7, 1], 2, 1, 64, False),
    ([405, 275], 16, 2, 128, True),              # decoder prompts, GQA
    ([33, 65, 64], 4, 2, 128, True),
])
def test_attention_matches_fp64(tiny_model, lens, heads, group, hd, causal):
    m = tiny_model[0]
    rng = np.random.default_rng(sum(lens) + heads)
    rows = sum(lens)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    q = bf16_bits(rng.standard_normal((rows, heads * hd)))
    k = bf16_bits(rng.standard_normal((rows, heads // group * hd)))
    v = bf16_bits(rng.standard_normal((rows, heads // group * hd)))
    out = bf16_vals(m.op_attention(q, k, v, cu, heads, group, hd, causal))
    ref = _ref_attention(bf16_vals(q), bf16_vals(k), bf16_vals(v), cu, heads, group, hd, causal)
    assert np.abs(out - ref).max() <= 2e-2     # P and O are rounded to bf16


def _rules_reference(picks, seen, max_tokens):
    """The reference's loop (Qwen3ASR.swift:388-389, 567-627) driven by given argmax results, via the restatement's rules."""
    import oracle
    g = oracle.generate
    tokens = [int(picks[0])]
    nxt = tokens[0]
    for it in range(1, max_tokens):
        if nxt in (g.IM_END, g.ENDOFTEXT):
            break
        if nxt == g.ASR_TEXT:
            seen = True
        if it >= len(picks):
            break
        nxt = int(picks[it])
        tokens.append(nxt)
        if g.stop_after_append(tokens, seen):
            break
    return tokens


def test_greedy_stop_rules_on_device(tiny_model):
    """Device-side bookkeeping of the greedy loop against the restatement on crafted and random argmax sequences:
    EOS kept then stop, control tokens (not applied to the prefill's token), 16-run after 32 tokens, <= 4 unique in
    the last 64, both only once <asr_text> was seen or the language was forced, and the max_tokens cap."""
    import oracle
    g = oracle.generate
    m = tiny_model[0]
    rng = np.random.default_rng(2024)
    T = 100
    rows, seen0 = [], []

    def add(seq, seen=0):
        seq = list(seq) + [7] * T
        rows.append(seq[:T])
        seen0.append(seen)

    add([5, 6, g.IM_END, 9, 9])                                   # EOS appended, then stop
    add([g.ENDOFTEXT, 3, 3])                                      # EOS as the prefill's token
    add([g.AUDIO_PAD, 3, g.AUDIO_PAD, 4])                         # control token: ignored at step 0, stops at step 2
    add([1, 2, g.IM_START, 4])
    add([1, g.AUDIO_START, 4])
    add([1, 2, 3, g.AUDIO_END])
    add([g.ASR_TEXT] + list(range(100, 120)) + [55] * 40)         # run of 16 reached at n = 37 >= 32
    add(list(range(100, 121)) + [55] * 40)                        # same without <asr_text>: no stop
    add(list(range(100, 121)) + [55] * 40, seen=1)                # forced language arms the rules from the start
    add([g.ASR_TEXT] + [55] * 40)                                 # run long before n = 32: stops exactly at n = 32
    add([g.ASR_TEXT] + [t for _ in range(40) for t in (11, 12, 13)])        # <= 4 unique in the last 64
    add([g.ASR_TEXT] + [t for _ in range(40) for t in (11, 12, 13, 14, 15)])  # 5 unique: no stop
    add([t for _ in range(40) for t in (11, 12, 13)], seen=0)     # rule not armed
    add(list(rng.integers(0, 6, size=T)), seen=1)                 # random low-diversity streams
    add(list(rng.integers(0, 3, size=T)), seen=1)
    add([g.ASR_TEXT] + list(rng.integers(0, 4, size=T)))
    for _ in range(40):
        seq = rng.integers(0, 30, size=T)
        seq[rng.integers(0, T, size=3)] = rng.choice([g.IM_END, g.ENDOFTEXT, g.ASR_TEXT, g.AUDIO_PAD, g.IM_START], size=3)
        add(list(seq), seen=int(rng.integers(0, 2)))
    picks = np.array(rows, dtype=np.int32)
    for max_tokens in (100, 40, 1):
        got = m.op_greedy_rules(picks, seen0, max_tokens=max_tokens)
        for i in range(len(rows)):
            assert got[i] == _rules_reference(picks[i], bool(seen0[i]), max_tokens), (i, max_tokens)
    free = m.op_greedy_rules(picks, seen0, max_tokens=100, stop_rules=False)
    assert all(len(t) == 100 for t in free)


def test_gelu_epilogue_against_exact_erf(tiny_model):
    """The GEMM epilogue's GELU (conv2 / conv3 / fc1 / proj1 of the encoder) on a sweep of bf16 inputs over [-8, 8]
    against 0.5 x (1 + erf(x / sqrt 2)) in float64 (AudioEncoder.swift:379-384, 127-129): within 1e-6 absolute everywhere,
    and within one bf16 step (2^-8 relative) down to x = -5.5, below which the value is under 1.1e-7.  The input rides in
    as a weight column against an activation of one, so the product is exact and the epilogue alone is measured."""
    from scipy.special import erfc
    m = tiny_model[0]
    xs = bf16_vals(bf16_bits(np.concatenate([np.linspace(-8, 8, 3201), -np.logspace(-3, 0.9, 400), np.logspace(-3, 0.9, 400)])))
    N = (len(xs) + 3) // 4 * 4
    w = np.zeros((N, 32), dtype=np.float32)
    w[:len(xs), 0] = xs
    a = np.zeros((16, 32), dtype=np.float32)
    a[:, 0] = 1.0
    got = m.op_gemm(bf16_bits(a), bf16_bits(w), gelu=True, tile=1)[0, :len(xs)].astype(np.float64)
    x64 = xs.astype(np.float64)
    want = 0.5 * x64 * erfc(-x64 / np.sqrt(2.0))
    assert np.abs(got - want).max() <= 1e-6
    near = x64 >= -5.5
    rel = np.abs(got - want)[near] / np.maximum(np.abs(want[near]), 1e-300)
    assert rel[want[near] != 0].max() <= 2.0 ** -8
    assert np.abs(got[~near]).max() <= 1.5e-7
