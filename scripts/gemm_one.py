#!/usr/bin/env python3
"""Runs one GEMM variant a few times (for rocprofv3 counter passes).

  python scripts/gemm_one.py M N K tile ksplit out_kind [iters]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))

from q3asr import Qwen3ASRModel, make_config, synth  # noqa: E402

M, N, K, tile, ks, kind = (int(x) for x in sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
cfg = make_config("tiny")
m = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=0))
ms = m.op_gemm_time(M, N, K, tile=tile, ksplit=ks, out_kind=kind, iters=iters)
print(f"M={M} N={N} K={K} tile={tile} ks={ks} kind={kind}: {ms * 1e3:.1f} us, {2.0 * M * N * K / ms / 1e9:.1f} TF/s")
m.close()
