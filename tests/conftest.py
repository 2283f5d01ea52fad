import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the library's decode loop reports its host-side progress on stderr: pytest shows it only for a test that fails or
# times out, which is where one wants to know at which step a call stood
os.environ.setdefault("Q3ASR_TRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """Every bound beside the value measured under it, and how many token comparisons were asserted."""
    import parity
    lines = parity.write_report(ROOT)
    if lines:
        terminalreporter.section("parity: measured against bounds")
        for ln in lines:
            terminalreporter.write_line(ln)


def bf16_bits(x):
    from q3asr.synth import to_bf16_bits
    return to_bf16_bits(np.asarray(x, dtype=np.float32))


def bf16_vals(bits):
    from q3asr.synth import from_bf16_bits
    return from_bf16_bits(bits)


def read_wav_pcm16(path):
    """16-bit PCM mono WAV -> (fp32 samples in [-1, 1), sample rate); walks chunks like
    AudioFileLoader.loadWAV (AudioFileLoader.swift:35-129)."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    pos, rate, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt, ch, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            assert fmt == 1 and ch == 1 and bits == 16
        elif cid == b"data":
            pcm = np.frombuffer(body, dtype="<i2")
        pos += 8 + size + (size & 1)
    return (pcm.astype(np.float32) / np.float32(32768.0)), rate


@pytest.fixture(scope="session")
def tiny_model():
    from q3asr import Qwen3ASRModel, make_config, synth
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=1)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck, device=0)
    yield m, cfg, ck
    m.close()


@pytest.fixture()
def pinned_model(monkeypatch):
    """A fresh tiny model whose decode projections are pinned to one kernel variant (64x64 MFMA tile, no split, at every group size).  The library
    otherwise measures the variants per group size, and two group sizes may then sum in different orders: fine within
    the logit tolerance, but a free-running greedy decode can take another branch at a near-tie.  Tests that demand
    token-for-token equality between differently shaped runs use this model."""
    from q3asr import Qwen3ASRModel, make_config, synth
    monkeypatch.setenv("Q3ASR_DECODE_TILE", "1")
    monkeypatch.setenv("Q3ASR_DECODE_SPLIT", "1")
    monkeypatch.setenv("Q3ASR_DECODE_NO_FUSE", "1")
    monkeypatch.setenv("Q3ASR_DECODE_GEMV", "0")       # groups of up to four rows would otherwise take the GEMV kernels
    cfg = make_config("tiny")
    ck = synth.make_checkpoint(cfg, seed=1)
    m = Qwen3ASRModel.from_checkpoint(cfg, ck, device=0)
    yield m, cfg, ck
    m.close()
