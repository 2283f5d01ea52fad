"""Oracle: log-mel frontend (test infrastructure, numpy fp32).

Follows Sources/Qwen3ASR/AudioPreprocessing.swift of the reference.
"""
import numpy as np

SAMPLE_RATE = 16000   # AudioPreprocessing.swift:9
N_FFT = 400           # :10
HOP = 160             # :11
N_MELS = 128          # :12
CHUNK_SECONDS = 30    # :13
N_BINS = N_FFT // 2 + 1
MAX_FRAMES = CHUNK_SECONDS * SAMPLE_RATE // HOP  # 3000, :307

f32 = np.float32


def mel_filterbank():
    """Slaney filterbank [128, 201] fp32 -- AudioPreprocessing.swift:23-126."""
    f_max = f32(SAMPLE_RATE) / f32(2.0)
    min_log_hz = f32(1000.0)
    min_log_mel = f32(15.0)
    logstep_hz2mel = f32(27.0) / np.log(f32(6.4))       # :31
    logstep_mel2hz = np.log(f32(6.4)) / f32(27.0)       # :32

    def hz_to_mel(hz):                                  # :34-40
        hz = f32(hz)
        if hz < min_log_hz:
            return f32(3.0) * hz / f32(200.0)
        return f32(min_log_mel + np.log(hz / min_log_hz) * logstep_hz2mel)

    def mel_to_hz(mel):                                 # :42-48
        mel = f32(mel)
        if mel < min_log_mel:
            return f32(200.0) * mel / f32(3.0)
        return f32(min_log_hz * np.exp((mel - min_log_mel) * logstep_mel2hz))

    fft_freqs = (np.arange(N_BINS, dtype=f32) * f_max / f32(N_BINS - 1)).astype(f32)  # :54-57
    mel_min, mel_max = hz_to_mel(0.0), hz_to_mel(f_max)
    n_pts = N_MELS + 2
    mel_pts = (mel_min + np.arange(n_pts, dtype=f32) * (mel_max - mel_min) / f32(n_pts - 1)).astype(f32)  # :65-68
    filt = np.array([mel_to_hz(m) for m in mel_pts], dtype=f32)                       # :71
    diff = (filt[1:] - filt[:-1]).astype(f32)                                         # :74-77
    fb = np.zeros((N_BINS, N_MELS), dtype=f32)
    for m in range(N_MELS):                                                           # :84-107
        down = (fft_freqs - filt[m]) / diff[m]
        up = (filt[m + 2] - fft_freqs) / diff[m + 1]
        fb[:, m] = np.maximum(f32(0.0), np.minimum(down, up))
    enorm = (f32(2.0) / (filt[2:] - filt[:-2])).astype(f32)                           # :110-115
    fb = (fb * enorm[None, :]).astype(f32)
    return np.ascontiguousarray(fb.T)                                                 # [128, 201], :118-123


def resample_linear(audio, from_rate, to_rate):
    """Linear-interpolation resample, double index / float fraction --
    AudioPreprocessing.swift:347-365."""
    audio = np.asarray(audio, dtype=f32)
    ratio = float(to_rate) / float(from_rate)
    out_len = int(float(len(audio)) * ratio)
    if out_len <= 0:
        return np.zeros(0, dtype=f32)
    i = np.arange(out_len, dtype=np.float64)
    src = i / ratio
    fl = src.astype(np.int64)
    ce = np.minimum(fl + 1, len(audio) - 1)
    frac = (src - fl.astype(np.float64)).astype(f32)
    return (audio[fl] * (f32(1) - frac) + audio[ce] * frac).astype(f32)


def reflect_pad(audio):
    """AudioPreprocessing.swift:134-154 (edge-safe reflect pad of n_fft/2)."""
    n = len(audio)
    pad = N_FFT // 2
    i = np.arange(pad)
    left = audio[np.maximum(0, np.minimum(pad - i, n - 1))]
    right = audio[np.maximum(0, n - 2 - i)]
    return np.concatenate([left, audio, right]).astype(f32)


def hann_window():
    """Periodic Hann, fp32 -- AudioPreprocessing.swift:161-164."""
    i = np.arange(N_FFT, dtype=f32)
    return (f32(0.5) * (f32(1.0) - np.cos(f32(2.0) * f32(np.pi) * i / f32(N_FFT)))).astype(f32)


def twiddles(mode):
    """[201, 400] cos / sin tables.

    mode="literal": angle = -2*pi*k*n/400 evaluated in fp32 exactly as
    AudioPreprocessing.swift:170-176 does (k*n up to 80 000 loses ~1e-4 relative
    in the angle).  mode="exact": the angle is reduced (k*n mod 400) and
    evaluated in float64 before rounding to fp32 -- the mathematically intended
    table, which is what an FFT computes.
    """
    k = np.arange(N_BINS)[:, None]
    n = np.arange(N_FFT)[None, :]
    if mode == "literal":
        ang = (f32(-2.0) * f32(np.pi) * k.astype(f32) * n.astype(f32) / f32(N_FFT)).astype(f32)
        return np.cos(ang).astype(f32), np.sin(ang).astype(f32)
    ang = -2.0 * np.pi * ((k * n) % N_FFT).astype(np.float64) / N_FFT
    return np.cos(ang).astype(f32), np.sin(ang).astype(f32)


def n_frames_total(n_samples):
    """(padded - n_fft)/hop + 1 -- AudioPreprocessing.swift:157."""
    return (n_samples + N_FFT - N_FFT) // HOP + 1


def n_frames_out(n_samples):
    """Frames returned: drop last (:301-302), cap 3000 (:307-316)."""
    return min(max(n_frames_total(n_samples) - 1, 0), MAX_FRAMES)


def log_mel_raw(audio, mode="exact", accumulate=f32, fb=None):
    """log10(max(mel, 1e-10)) for ALL frames [nFrames, 128] and the global max
    -- AudioPreprocessing.swift:131-270 (before clamp / scale / trimming)."""
    audio = np.asarray(audio, dtype=f32)
    padded = reflect_pad(audio)
    nfr = (len(padded) - N_FFT) // HOP + 1
    idx = (np.arange(nfr) * HOP)[:, None] + np.arange(N_FFT)[None, :]
    frames = (padded[idx] * hann_window()[None, :]).astype(f32)          # :184-194
    tr, ti = twiddles(mode)
    re = (frames.astype(accumulate) @ tr.T.astype(accumulate))           # :197-205
    im = (frames.astype(accumulate) @ ti.T.astype(accumulate))
    power = (re * re + im * im).astype(f32)                              # :207
    fb = mel_filterbank() if fb is None else np.asarray(fb, dtype=f32)
    mel = (power.astype(accumulate) @ fb.T.astype(accumulate)).astype(f32)  # :232-240
    logmel = np.log10(np.maximum(mel, f32(1e-10))).astype(f32)           # :252-255
    return logmel, f32(logmel.max()) if logmel.size else f32(0)


def extract_features(audio, mode="exact", accumulate=f32, fb=None):
    """[128, T] fp32 log-mel -- AudioPreprocessing.swift:131-324.  fb: filterbank override (the
    [128, 201] table is ill-conditioned in its smallest edge weights: two correct fp32 evaluations of
    AudioPreprocessing.swift:98-102 with different libm exp/log differ by 2e-7 absolute, which is 0.2 %
    of the smallest weight)."""
    logmel, mx = log_mel_raw(audio, mode, accumulate, fb)
    x = np.maximum(logmel, f32(mx - f32(8.0)))                           # :269-276
    x = ((x + f32(4.0)) / f32(4.0)).astype(f32)                          # :286-288
    t = min(max(x.shape[0] - 1, 0), MAX_FRAMES)                          # :301-316
    return np.ascontiguousarray(x[:t].T)                                 # :322-323


def process(audio, sample_rate, mode="exact"):
    """WhisperFeatureExtractor.process -- AudioPreprocessing.swift:330-344."""
    audio = np.asarray(audio, dtype=f32)
    if sample_rate != SAMPLE_RATE:
        audio = resample_linear(audio, sample_rate, SAMPLE_RATE)
    return extract_features(audio, mode)


def extract_features_c(audio):
    """The same stage through oracle/mel_direct.c: the reference's loops (direct DFT per frame and bin, twiddle rows
    copied per dot product, scalar mel product) in plain C on one thread -- AudioPreprocessing.swift:131-324.
    Returns ([128, T] fp32, seconds spent inside the C function)."""
    import ctypes
    import os
    import time
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libq3oracle.so")
    if not os.path.exists(so):
        raise RuntimeError("oracle/libq3oracle.so is missing: run `make -C oracle` (build() does)")
    lib = ctypes.CDLL(so)
    lib.q3oracle_mel_direct.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int)]
    audio = np.ascontiguousarray(audio, dtype=f32)
    fb = np.ascontiguousarray(mel_filterbank(), dtype=f32)
    out = np.zeros((N_MELS, MAX_FRAMES), dtype=f32)
    t = ctypes.c_int(0)
    t0 = time.perf_counter()
    rc = lib.q3oracle_mel_direct(audio.ctypes.data, len(audio), fb.ctypes.data, out.ctypes.data, MAX_FRAMES, ctypes.byref(t))
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError("q3oracle_mel_direct failed")
    return np.ascontiguousarray(out[:, :t.value]), dt
