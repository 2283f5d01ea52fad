"""16-bit PCM WAV ingest for the harness, with the reference loader's validation
(AudioFileLoader.loadWAV, AudioFileLoader.swift:35-129): fixed-offset format fields, a chunk walk
from byte 36 that rejects sizes running past the file (:86-95, pinned by
SecurityRegressionTests.swift:59-82), first channel only, samples / 32768."""
import struct

import numpy as np


class AudioLoadError(ValueError):
    pass


def load_wav(path):
    """Returns (fp32 samples in [-1, 1), sample rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) <= 44 or data[0:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise AudioLoadError("Invalid WAV file")
    audio_format, channels = struct.unpack_from("<HH", data, 20)
    (sample_rate,) = struct.unpack_from("<I", data, 24)
    (bits,) = struct.unpack_from("<H", data, 34)
    if audio_format != 1:
        raise AudioLoadError("Unsupported format: Not PCM format")
    if channels == 0:
        raise AudioLoadError("Invalid WAV file")
    if bits != 16:
        raise AudioLoadError("Unsupported format: Not 16-bit")
    off, size = 36, None
    while off < len(data) - 8:
        cid = data[off:off + 4]
        (csize,) = struct.unpack_from("<I", data, off + 4)
        if cid == b"data":
            off, size = off + 8, csize
            break
        nxt = off + 8 + csize
        if nxt > len(data):
            raise AudioLoadError("Invalid WAV file")
        off = nxt
    if size is None or off + size > len(data):
        raise AudioLoadError("Invalid WAV file")
    frames = size // (2 * channels)
    pcm = np.frombuffer(data, dtype="<i2", count=frames * channels, offset=off).reshape(frames, channels)[:, 0]
    return pcm.astype(np.float32) / np.float32(32768.0), int(sample_rate)


def resample_linear(samples, from_rate, to_rate):
    """AudioFileLoader.resample (:132-150): same linear interpolation as the feature extractor's.
    Host-side only (the CLI's 24 kHz detour, main.swift:305); transcription resamples on the GPU."""
    samples = np.asarray(samples, dtype=np.float32)
    ratio = float(to_rate) / float(from_rate)
    n = int(float(len(samples)) * ratio)
    if n <= 0:
        return np.zeros(0, np.float32)
    src = np.arange(n, dtype=np.float64) / ratio
    fl = src.astype(np.int64)
    ce = np.minimum(fl + 1, len(samples) - 1)
    frac = (src - fl).astype(np.float32)
    return (samples[fl] * (np.float32(1) - frac) + samples[ce] * frac).astype(np.float32)
