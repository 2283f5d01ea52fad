#!/usr/bin/env python3
"""Runs the mel stage alone on N synthetic 30 s clips (for rocprofv3 passes).  python scripts/mel_one.py [N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
from q3asr import Qwen3ASRModel, make_config, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = make_config("tiny")
m = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=0))
base = [synth.synth_audio(480000, 900 + i) for i in range(16)]
clips = [base[i % 16] for i in range(n)]
for _ in range(3):
    t0 = time.perf_counter()
    out = m.feature_extractor.process_batch(clips, 16000)
    print(f"{n} clips: {time.perf_counter() - t0:.3f} s wall (host copies included)", flush=True)
m.close()
