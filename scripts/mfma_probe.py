#!/usr/bin/env python3
"""Prints what the fp32 matrix pipe sustains on this GPU (register-only MFMA stream).  python scripts/mfma_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
from q3asr import Qwen3ASRModel, make_config, synth  # noqa: E402

cfg = make_config("tiny")
m = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=0))
for w in (1, 2):
    for it in (20000, 100000):
        print(f"{w} wave(s) per SIMD, {it} iterations: {m.op_mfma_f32_rate(w, it):.1f} TFLOP/s", flush=True)
m.close()
