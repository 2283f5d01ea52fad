"""CPU tests: the numpy restatement (oracle/) against the committed golden vectors and hand-computed cases."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, read_wav_pcm16
from q3asr import make_config, oracle_cfgs, synth


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


# ---------------------------------------------------------------- mel (G1, G2)
@pytest.fixture(scope="module")
def test_clip_16k():
    audio, rate = read_wav_pcm16(os.path.join(GOLDEN, "test_audio.wav"))
    assert rate == 24000 and len(audio) == 480000          # the reference's test resource: 20.0 s mono
    return oracle.mel.resample_linear(audio, rate, 16000)


def test_resample_length_and_endpoints(test_clip_16k):
    assert len(test_clip_16k) == 320000                     # Int(480000 * 16000 / 24000)
    x = np.arange(12, dtype=np.float32)
    y = oracle.mel.resample_linear(x, 24000, 16000)
    assert len(y) == 8 and np.allclose(y, np.arange(8) * 1.5)
    assert len(oracle.mel.resample_linear(x, 16000, 16000 * 3)) == 36


def test_filterbank_matches_hf_float64():
    fb = oracle.mel.mel_filterbank()
    ref = np.load(os.path.join(GOLDEN, "g2_filterbank_hf.npy"))
    assert fb.shape == ref.shape == (128, 201)
    assert np.abs(fb - ref).max() <= 5e-7


def test_mel_matches_hf_on_reference_clip(test_clip_16k):
    """fp32 restatement vs the float64 extractor: within 5e-5 when both use the same filterbank table,
    within 1e-4 with the restatement's own fp32 table (filter 53 holds a 1.2e-4 edge weight that moves
    by 0.2 % between fp32 and float64 evaluation of the slaney formula)."""
    got = oracle.mel.extract_features(test_clip_16k, "exact")
    same_fb = oracle.mel.extract_features(test_clip_16k, "exact", fb=np.load(os.path.join(GOLDEN, "g2_filterbank_hf.npy")))
    assert got.shape == (128, 2000)
    for name, sl in (("g1_mel_hf_f450_600.npy", slice(450, 600)), ("g1_mel_hf_f700_800.npy", slice(700, 800)),
                     ("g1_mel_hf_f1900_2000.npy", slice(1900, 2000))):
        ref = np.load(os.path.join(GOLDEN, name))
        assert np.abs(got[:, sl] - ref).max() <= 1e-4
        assert np.abs(same_fb[:, sl] - ref).max() <= 5e-5


def test_mel_literal_twiddles_residual(test_clip_16k):
    """The reference's fp32 angle -2*pi*k*n/400 (k*n up to 80 000) moves the result by more than the
    1e-4 parity target on its own; recorded so the choice of exact twiddles is visible."""
    exact = oracle.mel.extract_features(test_clip_16k[80000:128000], "exact")
    literal = oracle.mel.extract_features(test_clip_16k[80000:128000], "literal")
    d = np.abs(exact - literal)
    assert 1e-5 < d.max() < 2e-2


@pytest.mark.parametrize("n,frames", [(160, 1), (161, 1), (480000, 3000), (480160, 3000), (800000, 3000), (399, 2), (16000, 100)])
def test_mel_frame_counts(n, frames):
    x = synth.synth_audio(n, n)
    assert oracle.mel.extract_features(x).shape == (128, frames) == (128, oracle.mel.n_frames_out(n))


def test_mel_max_is_taken_over_untruncated_clip():
    """Clamp floor uses the maximum over ALL frames (dropped last frame and frames past 30 s included)."""
    x = synth.synth_audio(16000 * 31, 5) * 0.01
    x[-8000:] *= 100.0                                      # loud tail beyond the 3000-frame cut
    feats = oracle.mel.extract_features(x)
    quiet = oracle.mel.extract_features(x[:480000])
    assert feats.shape == quiet.shape == (128, 3000)
    assert feats.min() > quiet.min() + 0.5


# ---------------------------------------------------------------- encoder (G3)
@pytest.mark.parametrize("frames", [250, 300, 1000])
def test_encoder_matches_hf(frames):
    cfg = make_config("tiny")
    w = synth.checkpoint_to_f32(synth.make_checkpoint(cfg, seed=11))
    g = np.load(os.path.join(GOLDEN, f"g3_encoder_hf_{frames}.npz"))
    enc_cfg, _ = oracle_cfgs(cfg)
    out = oracle.encoder.encode(g["mel"], w, enc_cfg)
    assert out.shape == g["out"].shape == (oracle.encoder.output_length(frames), cfg.enc_out_dim)
    assert _rel(out, g["out"]) <= 2e-4


def test_encoder_length_algebra():
    ol = oracle.encoder.output_length
    assert [ol(t) for t in (3000, 2000, 500, 100, 50, 1, 101, 250)] == [390, 260, 65, 13, 7, 1, 14, 33]
    cfg = oracle_cfgs(make_config("0.6b"))[0]
    assert oracle.encoder.window_cu_seqlens(3000, cfg) == [0, 104, 208, 312, 390]
    assert oracle.encoder.window_cu_seqlens(2000, cfg) == [0, 104, 208, 260]     # AudioEncoder.swift:458
    assert oracle.encoder.window_cu_seqlens(50, cfg) == [0, 7]                    # single short chunk: window 7 * 8
    assert oracle.encoder.chunk_lengths(250) == [100, 100, 50]


def test_single_short_chunk_is_not_padded():
    """A lone chunk shorter than 100 frames is convolved at its own length (AudioEncoder.swift:348-368):
    its result differs from the same frames followed by a second chunk, whose presence pads it to 100."""
    cfg = make_config("tiny")
    w = synth.checkpoint_to_f32(synth.make_checkpoint(cfg, seed=11))
    enc_cfg, _ = oracle_cfgs(cfg)
    mel = oracle.mel.process(synth.synth_audio(160 * 150, 3), 16000)
    alone = oracle.encoder.frontend(mel[:, 100:150], w, enc_cfg)
    second = oracle.encoder.frontend(mel, w, enc_cfg)[13:]
    assert alone.shape == second.shape == (7, cfg.enc_d_model)
    assert np.abs(alone[:5] - second[:5]).max() < 1e-5       # receptive fields inside the valid frames agree
    assert np.abs(alone[6] - second[6]).max() > 1e-4         # the last token sees the padding difference


# ---------------------------------------------------------------- decoder (G4)
def test_decoder_matches_hf_rotate_half():
    cfg = make_config("tiny", vocab=512, rope_interleaved=0)
    w = synth.checkpoint_to_f32(synth.make_checkpoint(cfg, seed=13))
    g = np.load(os.path.join(GOLDEN, "g4_decoder_hf.npz"))
    dec = oracle.decoder.Decoder(w, oracle_cfgs(cfg)[1])
    h, cache = dec.forward(g["x"], None)
    assert _rel(h, g["prefill_hidden"]) <= 2e-4
    logits = [dec.logits(h[-1:])[0]]
    ids = [int(np.argmax(logits[-1]))]
    for _ in range(8):
        h, cache = dec.forward(dec.embed([ids[-1]]), cache)
        logits.append(dec.logits(h)[0])
        ids.append(int(np.argmax(logits[-1])))
    assert ids == g["ids"].tolist()
    assert _rel(np.stack(logits), g["logits"]) <= 5e-4


def test_interleaved_rope_is_a_permutation_of_rotate_half():
    """Interleaved pairs (2i, 2i+1) == rotate-half pairs (i, i+half) on de-interleaved channels."""
    cfg_i = dict(rotary_dim=128, rope_theta=1e6, rope_interleaved=True)
    cfg_h = dict(cfg_i, rope_interleaved=False)
    x = np.random.default_rng(0).standard_normal((3, 5, 128)).astype(np.float32)
    perm = np.concatenate([np.arange(0, 128, 2), np.arange(1, 128, 2)])
    a = oracle.decoder.rope(x, 7, cfg_i)[..., perm]
    b = oracle.decoder.rope(x[..., perm], 7, cfg_h)
    assert np.abs(a - b).max() < 1e-6
    part = oracle.decoder.rope(x, 2, dict(cfg_i, rotary_dim=64))
    assert np.array_equal(part[..., 64:], x[..., 64:])        # partial rotary leaves the tail untouched


# ---------------------------------------------------------------- MLX q4 (G5)
def test_q4_known_answer():
    codes = (np.arange(64) % 16).astype(np.uint8)[None, :]
    packed = oracle.q4.pack(codes)
    assert packed.shape == (1, 8) and packed[0, 0] == 0x76543210 and packed[0, 1] == 0xFEDCBA98
    assert np.array_equal(oracle.q4.unpack(packed), codes)
    w = oracle.q4.dequantize(packed, np.array([[0.5]], np.float32), np.array([[-2.0]], np.float32))
    assert np.array_equal(w[0], codes[0] * 0.5 - 2.0)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((256, 128)).astype(np.float32)
    p, s, b = oracle.q4.quantize(x)
    assert p.shape == (256, 16) and s.shape == b.shape == (256, 2)
    assert np.abs(oracle.q4.dequantize(p, s, b) - x).max() <= 0.51 * s.max() + 1e-2


# ---------------------------------------------------------------- prompt, splice, stop rules (G6, G7)
def test_prompt_layout():
    ids, a0, a1 = oracle.generate.prompt_ids(260)
    assert len(ids) == 275 and (a0, a1) == (9, 269)
    assert ids[:9] == [151644, 8948, 198, 151645, 198, 151644, 872, 198, 151669]
    assert set(ids[9:269]) == {151676}
    assert ids[269:] == [151670, 151645, 198, 151644, 77091, 198]
    ids, _, _ = oracle.generate.prompt_ids(390, lang_ids=[4128, 6364])
    assert len(ids) == 405 + 3 and ids[-3:] == [4128, 6364, 151704]      # <asr_text> appended when absent
    ids, _, _ = oracle.generate.prompt_ids(0, lang_ids=[4128, 151704])
    assert ids[-2:] == [4128, 151704]


def test_splice_scales_audio_to_text_std():
    class Dec:
        def embed(self, ids):
            rng = np.random.default_rng(3)
            return rng.standard_normal((len(ids), 16)).astype(np.float32) * 0.02
    ids, a0, a1 = oracle.generate.prompt_ids(10)
    audio = np.random.default_rng(4).standard_normal((10, 16)).astype(np.float32) * 3.0
    x = oracle.generate.splice(Dec(), ids, a0, a1, audio, True)
    text = Dec().embed(ids)
    assert x.shape == (25, 16)
    assert np.array_equal(x[:9], text[:9]) and np.array_equal(x[19:], text[19:])
    assert abs(x[9:19].std() / text.std() - 1.0) < 1e-5       # std over all S rows, population variance
    y = oracle.generate.splice(Dec(), ids, a0, a1, audio, False)
    assert np.array_equal(y[9:19], audio)
    z = oracle.generate.splice(Dec(), ids, a0, a1, np.zeros_like(audio), True)
    assert np.array_equal(z[9:19], np.zeros_like(audio))      # audioStd == 0: no rescale


class _ScriptedDecoder:
    """Decoder stub that emits a scripted token sequence (one-hot logits)."""

    def __init__(self, script):
        self.script, self.i = list(script), 0

    def forward(self, x, cache):
        return np.zeros((x.shape[0], 4), np.float32), cache

    def embed(self, ids):
        return np.zeros((len(ids), 4), np.float32)

    def logits(self, h):
        out = np.zeros((h.shape[0], 152000), np.float32)
        out[:, self.script[min(self.i, len(self.script) - 1)]] = 1.0
        self.i += 1
        return out


def _run(script, **kw):
    return oracle.generate.greedy(_ScriptedDecoder(script), np.zeros((3, 4), np.float32), **kw)


def test_stop_rules():
    assert _run([5, 6, 151645, 9]) == [5, 6, 151645]                       # EOS is appended, then the loop stops
    assert _run([5, 151643, 9]) == [5, 151643]
    assert _run([151645]) == [151645]
    assert _run([5, 151644, 9]) == [5, 151644]                             # control token: appended, stop at once
    assert _run([5, 6, 7], max_tokens=2) == [5, 6]
    assert len(_run([7] * 500)) == 448                                     # no <asr_text>: repetition guards stay off
    # inside the ASR text region: 16-run once >= 32 tokens
    toks = _run([151704] + list(range(100, 120)) + [7] * 100)
    assert len(toks) == 37 and toks[-16:] == [7] * 16 and toks[-17] != 7
    assert len(_run([7] * 100, forced_language=True)) == 32
    # <= 4 distinct tokens in the last 64, runs shorter than 16
    toks = _run([151704] + [1, 2, 3, 4] * 100)
    assert len(toks) == 65                                                 # first length whose last 64 exclude the tag
    assert oracle.generate.post_process("language English<asr_text> hello <asr_text> world ") == "world"
    assert oracle.generate.post_process("no tag") == "no tag"


def test_c_restatement_of_the_direct_dft_agrees_with_the_numpy_one():
    """oracle/mel_direct.c (the reference's loops in plain C, timed as the reference frontend's own cost) against
    the numpy restatement's "literal" mode on the reference's test clip: two evaluations of the same fp32 arithmetic in
    different summation orders and with different libm cos / sin.  The literal twiddle angle -2 pi k n / 400 in fp32
    carries an error of 4e-3 rad at k n = 80 000, which the two libm then turn into table values that may differ in the
    last bits, so the bound is the one the GPU test uses for the literal form: 4e-3 at worst, 4e-5 on average."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    audio, rate = read_wav_pcm16(os.path.join(GOLDEN, "test_audio.wav"))
    a16 = oracle.mel.resample_linear(audio, rate, 16000)[80000:112000]      # 2 s of the spoken part
    got, seconds = oracle.mel.extract_features_c(a16)
    want = oracle.mel.extract_features(a16, "literal")
    assert got.shape == want.shape == (128, 200)
    d = np.abs(got - want)
    assert d.max() <= 4e-3 and d.mean() <= 4e-5
    assert np.abs(got - oracle.mel.extract_features(a16, "exact")).max() <= 5e-3
    assert seconds > 0
