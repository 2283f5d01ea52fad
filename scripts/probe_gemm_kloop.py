#!/usr/bin/env python3
"""K-loop rate of the 256x256 kernel per busy CU: shapes with one tile per workgroup and a long K, operands small enough
to stay in the last-level cache, against a shape that streams its operands from HBM."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
import q3asr.lib  # noqa: E402
from q3asr import Qwen3ASRModel, make_config, synth  # noqa: E402
if os.environ.get("Q3LIB"):
    q3asr.lib.LIB_PATH = os.path.abspath(os.environ["Q3LIB"])

cfg = make_config("tiny")
m = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=0))
for M, N, K in ((2048, 4096, 8192), (2048, 8192, 8192), (4096, 4096, 8192), (4096, 4096, 16384), (1024, 2048, 16384), (207360, 4096, 2048), (65536, 4096, 8192)):
    tiles = (M // 256) * (N // 256)
    busy = min(tiles, 256)
    ms = min(m.op_gemm_time(M, N, K, tile=3, out_kind=0, iters=10) for _ in range(3))
    tf = 2.0 * M * N * K / ms / 1e9
    print(f"M={M:6d} N={N:5d} K={K:5d} tiles {tiles:5d}: {ms:8.3f} ms {tf:7.1f} TF/s = {tf / busy:5.2f} TF/s per busy CU, operands {(M + N) * K * 2 / 1e6:6.0f} MB", flush=True)
m.close()
