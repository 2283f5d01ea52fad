#!/usr/bin/env python3
"""Times every (tile, split-K) variant of the decode projections at small row counts (0.6B and 1.7B shapes).

  python scripts/bench_gemm_small.py [rows ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))

from q3asr import Qwen3ASRModel, make_config, synth  # noqa: E402

rows = [int(x) for x in sys.argv[1:]] or [1, 16, 128, 256]
cfg = make_config("tiny")
m = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=0))
shapes = {"0.6b qkv": (4096, 1024), "0.6b o": (1024, 2048), "0.6b gate_up": (6144, 1024), "0.6b down": (1024, 3072),
          "1.7b qkv": (4096, 2048), "1.7b o": (2048, 2048), "1.7b gate_up": (12288, 2048), "1.7b down": (2048, 6144)}
for M in rows:
    for name, (N, K) in shapes.items():
        best = None
        line = []
        for tile in (1, 2, 0):
            for ks in (1, 2, 4, 8):
                if K // ks < 256:
                    continue
                kind = 1 if ks == 1 else 5      # fp32 rows, or split-K slabs
                us = m.op_gemm_time(M, N, K, tile=tile, ksplit=ks, out_kind=kind, iters=50) * 1e3
                line.append(f"t{tile}k{ks}:{us:.1f}")
                if best is None or us < best[0]:
                    best = (us, tile, ks)
        floor = N * K * 2 / 5.9e6   # us to stream the weights at the measured read ceiling
        print(f"M={M:4d} {name:13s} best {best[0]:5.1f} us (tile {best[1]} ks {best[2]}), weight-stream floor {floor:4.1f} us | " + " ".join(line),
              flush=True)
m.close()
