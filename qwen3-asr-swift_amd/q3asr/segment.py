"""Long-audio policy in front of the transcription path.

The reference feeds one clip to the model and silently drops everything after 30 s: the feature
extractor truncates to 3000 frames (AudioPreprocessing.swift:307-316).  ``mode="truncate"`` keeps that
behaviour (the parity default).  ``mode="segment"`` cuts a long recording into pieces of at most 30 s,
transcribes them as independent utterances of one batch and joins the results; a cut is placed at the
quietest 20 ms frame of the last seconds of each window, using the RMS-energy rule of the reference's
SimpleVAD (TranscriptStabilizer.swift:152-251: a frame is silence when its RMS is below 0.01).
"""
import numpy as np

MAX_SECONDS = 30.0          # WhisperFeatureExtractor.chunk_length (AudioPreprocessing.swift:13)
FRAME_MS = 20
ENERGY_THRESHOLD = 0.01     # SimpleVAD.Config.energyThreshold (TranscriptStabilizer.swift:159)
MIN_PIECE_MS = 100          # no piece shorter than this: the library rejects utterances below one 10 ms frame


def frame_rms(audio, frame):
    """RMS of consecutive `frame`-sample blocks (SimpleVAD.calculateRMS, TranscriptStabilizer.swift:231-238)."""
    n = len(audio) // frame
    if n == 0:
        return np.zeros(0, dtype=np.float32)
    x = np.asarray(audio[:n * frame], dtype=np.float32).reshape(n, frame)
    return np.sqrt(np.mean(x * x, axis=1, dtype=np.float32))


def split_points(audio, sample_rate, max_seconds=MAX_SECONDS, search_seconds=5.0, energy=True):
    """[(start, end)] sample ranges covering `audio` exactly once, each at most `max_seconds` long.
    With energy=True a boundary moves back (by at most `search_seconds`) to the first sample of the quietest
    20 ms frame, preferring the latest frame that counts as silence; otherwise windows are cut at the limit."""
    n = len(audio)
    limit = int(max_seconds * sample_rate)
    frame = max(1, sample_rate * FRAME_MS // 1000)
    out, start = [], 0
    while n - start > limit:
        cut = start + limit
        if energy:
            lo = max(start + frame, cut - int(search_seconds * sample_rate))
            rms = frame_rms(audio[lo:cut], frame)
            if len(rms):
                silent = np.nonzero(rms < ENERGY_THRESHOLD)[0]
                k = int(silent[-1]) if len(silent) else int(np.argmin(rms))
                cut = lo + k * frame
        out.append((start, cut))
        start = cut
    out.append((start, n))
    # A recording a few ms over a multiple of the window leaves a sliver: move the last cut back so that the tail
    # piece reaches the minimum length (nothing is dropped, no piece grows past the window).
    min_len = max(1, sample_rate * MIN_PIECE_MS // 1000)
    if len(out) >= 2 and out[-1][1] - out[-1][0] < min_len:
        s0, _ = out[-2]
        cut = max(s0 + 1, n - min_len)
        out[-2], out[-1] = (s0, cut), (cut, n)
    return out


def segment_batch(audios, sample_rate, mode="truncate", **kw):
    """-> (pieces, owner): the arrays to transcribe and, for each, the index of the utterance it came from."""
    if mode not in ("truncate", "segment", "segment_fixed"):
        raise ValueError(f"unknown long-audio mode {mode!r}")
    pieces, owner = [], []
    for i, a in enumerate(audios):
        a = np.asarray(a, dtype=np.float32)
        if mode == "truncate":
            spans = [(0, len(a))]       # the library itself keeps the first 3000 frames
        else:
            spans = split_points(a, sample_rate, energy=(mode == "segment"), **kw)
        for s, e in spans:
            pieces.append(a[s:e])
            owner.append(i)
    return pieces, owner


def join_tokens(piece_tokens, owner, n_utt, eos_ids=(151645, 151643)):
    """Concatenates the pieces' token lists per utterance, dropping each piece's closing EOS
    (the reference appends EOS to the list before it stops, Qwen3ASR.swift:570-575)."""
    out = [[] for _ in range(n_utt)]
    for toks, i in zip(piece_tokens, owner):
        toks = list(toks)
        while toks and toks[-1] in eos_ids:
            toks.pop()
        out[i].extend(toks)
    return out
