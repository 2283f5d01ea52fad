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
