"""GPU parity of the log-mel frontend against the CPU restatement.

Tolerance: 1e-4 absolute against the restatement evaluated with exact twiddles, float64
accumulation and the library's own fp32 filterbank table (itself checked against the restatement's
table to 4e-7: the smallest edge weights of the slaney table, e.g. 1.2e-4 in filter 53, move by
0.2 % with the libm used for exp/log, so a second table is a second, equally valid, rounding of the
reference formula).  Against the restatement with its own table and fp32 accumulation, whose
rounding noise alone is up to 8e-5 on white noise, the bound is 2.5e-4."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, read_wav_pcm16
from q3asr import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4
TOL_F32 = 2.5e-4


_FB = {}


def _check(got, audio, rate, model):
    if "fb" not in _FB:
        _FB["fb"] = model.feature_extractor.mel_filterbank()
    a16 = audio if rate == 16000 else oracle.mel.resample_linear(audio, rate, 16000)
    ref64 = oracle.mel.extract_features(a16, "exact", np.float64, fb=_FB["fb"])
    ref32 = oracle.mel.extract_features(a16, "exact", np.float32)
    assert got.shape == ref64.shape
    assert np.abs(got - ref64).max() <= TOL
    assert np.abs(got - ref32).max() <= TOL_F32
    return ref64


def test_filterbank_bit_close(tiny_model):
    fb = tiny_model[0].feature_extractor.mel_filterbank()
    ref = oracle.mel.mel_filterbank()
    assert fb.shape == (128, 201)
    assert np.abs(fb - ref).max() <= 4e-7      # libm logf/expf vs numpy fp32, values up to 0.025


@pytest.mark.parametrize("kind", ["speechlike", "noise"])
def test_mel_synthetic_30s(tiny_model, kind):
    audio = synth.synth_audio(480000, 11, kind)
    got = tiny_model[0].feature_extractor.process(audio, 16000)
    assert _check(got, audio, 16000, tiny_model[0]).shape == (128, 3000)


def test_mel_ragged_batch_and_edges(tiny_model):
    """Ragged lengths, a clip longer than 30 s (truncated to 3000 frames, max taken over the whole
    clip), clips shorter than one FFT window (reflect padding clamps)."""
    lens = [160, 333, 1000, 16000, 48017, 160 * 3000 + 5, 16000 * 33 + 77]
    audios = [synth.synth_audio(n, 100 + i) for i, n in enumerate(lens)]
    got = tiny_model[0].feature_extractor.process_batch(audios, 16000)
    for a, g in zip(audios, got):
        _check(g, a, 16000, tiny_model[0])
        assert g.shape[1] == min(len(a) // 160, 3000)


def test_mel_full_batch_1024_clips(tiny_model, monkeypatch):
    """The mel-only configuration at its full size, 1024 clips x 30 s in one call (1.97 GB in, 1.57 GB out).  The
    restatement cannot cover it in seconds, so three clips are checked against it and the rest through properties:
    a clip's features do not depend on its slot or on its neighbours (64 distinct clips repeated 16 times must give
    16 bit-identical copies), every clip spans at most 8 decades below its own maximum (range 2.0 after the
    (x + 4) / 4 scale, AudioPreprocessing.swift:269-288) and nothing is NaN."""
    m = tiny_model[0]
    base = [synth.synth_audio(480000, 500 + i, "noise" if i % 8 == 7 else "speechlike") for i in range(64)]
    clips = [base[i % 64] for i in range(1024)]
    got = m.feature_extractor.process_batch(clips, 16000)
    assert len(got) == 1024 and all(g.shape == (128, 3000) for g in got)
    for i in range(64):
        for r in range(1, 16):
            assert np.array_equal(got[i], got[i + 64 * r])
        assert np.isfinite(got[i]).all()
        assert got[i].max() - got[i].min() <= 2.0 + 1e-6
    for i in (0, 7, 63):
        _check(got[i], base[i], 16000, m)
    alone = m.feature_extractor.process(base[5], 16000)
    assert np.array_equal(alone, got[5])
    # The rows are stored scaled and only clamped afterwards where an utterance needs it; raw rows followed by a full
    # clamp-and-scale pass must give the same bits everywhere.
    monkeypatch.setenv("Q3ASR_MEL_TWO_PASS", "1")
    two = m.feature_extractor.process_batch(clips, 16000)
    for i in range(1024):
        assert np.array_equal(two[i], got[i]), i


def test_mel_test_wav_24k_resample(tiny_model):
    """The reference's own test clip (24 kHz, 20 s) through the linear resampler (AudioPreprocessing.swift:347-365)."""
    audio, rate = read_wav_pcm16(os.path.join(GOLDEN, "test_audio.wav"))
    assert rate == 24000 and len(audio) == 480000
    got = tiny_model[0].feature_extractor.process(audio, rate)
    assert _check(got, audio, rate, tiny_model[0]).shape == (128, 2000)


def _hip_runtime():
    """The HIP runtime the library itself is linked against (already mapped into this process)."""
    import ctypes
    with open("/proc/self/maps") as f:
        paths = {line.split()[-1] for line in f if "libamdhip64" in line and "/torch/" not in line}
    assert paths, "libq3asr is loaded, so its HIP runtime must be mapped"
    hip = ctypes.CDLL(sorted(paths)[0])
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    return hip


def test_mel_device_buffers_match_host_path(tiny_model):
    """The stage on buffers that already live in HBM (what a batching host hands over) writes the caller's device
    buffer directly; it must give exactly what the host-buffer call returns, ragged lengths included."""
    import ctypes
    m = tiny_model[0]
    hip = _hip_runtime()
    lens = [480000, 16000 * 7 + 13, 160 * 250, 333]
    audios = [synth.synth_audio(n, 300 + i) for i, n in enumerate(lens)]
    want = m.feature_extractor.process_batch(audios, 16000)
    cat = np.concatenate(audios)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    got = np.full((len(lens), 128, 3000), np.nan, dtype=np.float32)
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_in), cat.nbytes) == 0 and hip.hipMalloc(ctypes.byref(d_out), got.nbytes) == 0
    try:
        assert hip.hipMemcpy(d_in, cat.ctypes.data, cat.nbytes, 1) == 0          # host to device
        assert hip.hipMemcpy(d_out, got.ctypes.data, got.nbytes, 1) == 0
        frames = m.feature_extractor.process_device(d_in.value, off, d_out.value, 3000)
        assert hip.hipMemcpy(got.ctypes.data, d_out, got.nbytes, 2) == 0          # device to host
    finally:
        hip.hipFree(d_in)
        hip.hipFree(d_out)
    for i, w in enumerate(want):
        assert frames[i] == w.shape[1]
        assert np.array_equal(got[i, :, :frames[i]], w)


def test_mel_one_launch_clamp_path_with_silent_stretches(tiny_model, monkeypatch):
    """600 short utterances (0.2 - 2 s), most with a stretch of digital silence (ten decades under the peak: the max - 8
    clamp of AudioPreprocessing.swift:269-276 is active there), a few without: rows stored scaled and clamped only where
    an utterance needs it, against raw rows followed by a full clamp-and-scale pass, bit for bit, and five utterances
    against the restatement."""
    m = tiny_model[0]
    rng = np.random.default_rng(5)
    clips = []
    for i in range(600):
        n = int(rng.integers(3200, 32000))
        a = synth.synth_audio(n, 7000 + i, "noise" if i % 5 == 4 else "speechlike")
        if i % 4 != 3:
            z0 = int(rng.integers(0, n // 2))
            a[z0:z0 + n // 3] = 0.0
        clips.append(a)
    assert m.feature_extractor.process_batch(clips[:1], 16000)[0].shape[0] == 128
    got = m.feature_extractor.process_batch(clips, 16000)
    monkeypatch.setenv("Q3ASR_MEL_TWO_PASS", "1")
    two = m.feature_extractor.process_batch(clips, 16000)
    clamped = 0
    for i in range(600):
        assert np.array_equal(got[i], two[i]), i
        clamped += int((got[i] == got[i].min()).sum() > 128)       # a clamp floor shows as many equal minima
    assert clamped >= 300
    for i in (0, 3, 4, 299, 599):
        _check(got[i], clips[i], 16000, m)


def test_mel_row_stride_that_rules_out_16_byte_stores(tiny_model):
    """An output row stride that is not a multiple of four floats: the kernel then stores frame by frame and the clamp
    works element by element.  Same values as with the default stride, bit for bit, on a ragged batch with silent
    stretches (clamp active), tile tails and a clip past 30 s."""
    m = tiny_model[0]
    lens = [160, 333, 5000, 16000, 48017, 160 * 3000 + 5, 16000 * 31 + 77, 70000]
    audios = [synth.synth_audio(n, 300 + i) for i, n in enumerate(lens)]
    for a in audios[2:]:
        a[len(a) // 4: len(a) // 2] = 0.0
    want = m.feature_extractor.process_batch(audios, 16000)
    got = m.feature_extractor.process_batch(audios, 16000, t_stride=3001)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    _check(got[4], audios[4], 16000, m)
