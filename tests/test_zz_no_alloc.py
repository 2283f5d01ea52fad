"""The log-mel entry point that allocates nothing (fixed file-scope device scratch) against the allocating one, through
the C ABI on device buffers: two calls in a row on one model, the second with a smaller input, both bit-identical to
q3asr_mel; oversize requests are rejected."""
import ctypes

import numpy as np
import pytest

from q3asr import synth
from q3asr.lib import Q3Error

pytestmark = pytest.mark.gpu


def _hip():
    with open("/proc/self/maps") as f:
        paths = {line.split()[-1] for line in f if "libamdhip64" in line and "/torch/" not in line}
    assert paths, "libq3asr is loaded, so its HIP runtime must be mapped"
    hip = ctypes.CDLL(sorted(paths)[0])
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    return hip


def test_mel_noalloc_matches_the_allocating_entry_point(tiny_model):
    m = tiny_model[0]
    hip = _hip()
    T = 3000
    big = [480000, 16000 * 7 + 13, 160 * 250, 333, 16000 * 31 + 5, 160, 48017]      # ragged, > 30 s, < one FFT window
    small = [16000 * 3, 200, 160 * 3000 + 159]
    cap_in = sum(big) * 4
    cap_out = len(big) * 128 * T * 4
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_in), cap_in) == 0 and hip.hipMalloc(ctypes.byref(d_out), cap_out) == 0
    try:
        for call, lens in enumerate((big, small)):
            audios = [synth.synth_audio(n, 40 * call + i, "noise" if i % 3 == 2 else "speechlike") for i, n in enumerate(lens)]
            cat = np.concatenate(audios)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            assert hip.hipMemcpy(d_in, cat.ctypes.data, cat.nbytes, 1) == 0
            want = np.full((len(lens), 128, T), np.nan, dtype=np.float32)
            got = np.full_like(want, np.nan)
            # the allocating entry point on the same device buffers
            assert hip.hipMemset(d_out, 0xFF, want.nbytes) == 0
            fr_want = m.feature_extractor.process_device(d_in.value, off, d_out.value, T)
            assert hip.hipMemcpy(want.ctypes.data, d_out, want.nbytes, 2) == 0
            # the one that allocates nothing
            assert hip.hipMemset(d_out, 0xFF, want.nbytes) == 0
            fr_got = m.feature_extractor.process_device_noalloc(d_in.value, off, d_out.value, T)
            assert hip.hipMemcpy(got.ctypes.data, d_out, got.nbytes, 2) == 0
            assert np.array_equal(fr_got, fr_want) and fr_got.tolist() == [min(n // 160, 3000) for n in lens]
            for u in range(len(lens)):
                f = int(fr_want[u])
                assert np.isfinite(want[u, :, :f]).all()
                assert np.array_equal(got[u, :, :f], want[u, :, :f]), (call, u)
        # beyond the fixed scratch: rejected with a clear error, nothing launched
        off = np.arange(258, dtype=np.int64) * 1600
        with pytest.raises(Q3Error, match="256 utterances"):
            m.feature_extractor.process_device_noalloc(d_in.value, off, d_out.value, T)
        with pytest.raises(Q3Error, match="longer than 600 s"):
            m.feature_extractor.process_device_noalloc(d_in.value, np.array([0, 16000 * 601], np.int64), d_out.value, T)
        with pytest.raises(Q3Error, match="t_stride"):
            m.feature_extractor.process_device_noalloc(d_in.value, np.array([0, 480000], np.int64), d_out.value, 100)
        # and the model still works afterwards
        fr = m.feature_extractor.process_device_noalloc(d_in.value, np.array([0, 16000], np.int64), d_out.value, T)
        assert fr.tolist() == [100]
    finally:
        hip.hipFree(d_in)
        hip.hipFree(d_out)


def test_mel_noalloc_async_calls_on_two_streams_do_not_race(tiny_model):
    """Two calls issued back to back on two different streams, neither waited for: the second call's set-up rewrites
    the one per-GPU scratch, so the library must hold it back on the device until the first call's kernels are done.
    Both outputs equal q3asr_mel's, and so does a third call through a second model on the same GPU."""
    from q3asr import Qwen3ASRModel, make_config
    m = tiny_model[0]
    hip = _hip()
    T = 3000
    lens_a = [480000] * 24 + [16000 * 11 + 7]            # long enough that the first call is still running when the second arrives
    lens_b = [16000 * 5, 123456, 480000]
    lens_c = [99999, 160 * 777]
    sets = []
    bufs = []
    s1, s2 = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(s1)) == 0 and hip.hipStreamCreate(ctypes.byref(s2)) == 0
    other = Qwen3ASRModel(make_config("tiny"), device=0)          # the log-mel stage needs no weights
    try:
        for k, lens in enumerate((lens_a, lens_b, lens_c)):
            base = [synth.synth_audio(n, 900 + 10 * k + i) for i, n in enumerate(sorted(set(lens)))]
            by_len = dict(zip(sorted(set(lens)), base))
            cat = np.concatenate([by_len[n] for n in lens])
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            d_in, d_out, d_ref = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            nbytes = len(lens) * 128 * T * 4
            for d, n in ((d_in, cat.nbytes), (d_out, nbytes), (d_ref, nbytes)):
                assert hip.hipMalloc(ctypes.byref(d), n) == 0
                bufs.append(d)
            assert hip.hipMemcpy(d_in, cat.ctypes.data, cat.nbytes, 1) == 0
            assert hip.hipMemset(d_out, 0xFF, nbytes) == 0 and hip.hipMemset(d_ref, 0xFF, nbytes) == 0
            m.feature_extractor.process_device(d_in.value, off, d_ref.value, T)
            sets.append((lens, off, d_in, d_out, d_ref, nbytes))
        fr = []
        fr.append(m.feature_extractor.process_device_noalloc(sets[0][2].value, sets[0][1], sets[0][3].value, T, stream=s1))
        fr.append(m.feature_extractor.process_device_noalloc(sets[1][2].value, sets[1][1], sets[1][3].value, T, stream=s2))
        fr.append(other.feature_extractor.process_device_noalloc(sets[2][2].value, sets[2][1], sets[2][3].value, T, stream=s1))
        assert hip.hipStreamSynchronize(s2) == 0 and hip.hipStreamSynchronize(s1) == 0
        for (lens, off, d_in, d_out, d_ref, nbytes), f in zip(sets, fr):
            got = np.empty((len(lens), 128, T), dtype=np.float32)
            want = np.empty_like(got)
            assert hip.hipMemcpy(got.ctypes.data, d_out, nbytes, 2) == 0 and hip.hipMemcpy(want.ctypes.data, d_ref, nbytes, 2) == 0
            assert f.tolist() == [min(n // 160, 3000) for n in lens]
            for u in range(len(lens)):
                assert np.array_equal(got[u, :, :f[u]], want[u, :, :f[u]]), (len(lens), u)
    finally:
        other.close()
        hip.hipStreamDestroy(s1)
        hip.hipStreamDestroy(s2)
        for d in bufs:
            hip.hipFree(d)
