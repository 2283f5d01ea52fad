#!/usr/bin/env python3
"""Times the GEMM kernel variants on the shapes the 1.7B transcription path launches (512 clips x 30 s).

  python scripts/bench_gemm.py [prefill|decode|all]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))

from q3asr import Qwen3ASRModel, make_config, synth  # noqa: E402

OUT = dict(bf16=0, f32=1, acc=2, swiglu=3, argmax=4, slab=5)
R = 512 * 405                 # packed prompt rows of 512 clips
E = 512 * 390                 # encoder tokens
PREFILL = [
    ("dec qkv", R, 4096, 2048, "bf16"), ("dec o", R, 2048, 2048, "acc"), ("dec gate_up", R, 12288, 2048, "swiglu"),
    ("dec down", R, 2048, 6144, "acc"), ("enc qkv", E, 3072, 1024, "bf16"), ("enc fc1", E, 4096, 1024, "bf16"),
    ("enc fc2", E, 1024, 4096, "acc"), ("lm head", 512, 151936, 2048, "argmax"),
]
DECODE = [("qkv", 4096, 2048), ("o", 2048, 2048), ("gate_up", 12288, 2048), ("down", 2048, 6144)]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    cfg = make_config("tiny")
    m = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=0))
    if which in ("prefill", "all"):
        for name, M, N, K, out in PREFILL:
            row = []
            for tile in (0, 3):
                ms = m.op_gemm_time(M, N, K, tile=tile, out_kind=OUT[out], iters=5)
                row.append(f"tile{tile} {ms:8.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF/s")
            print(f"{name:12s} M={M:6d} N={N:6d} K={K:5d} {out:7s} | " + " | ".join(row), flush=True)
    if which in ("decode", "all"):
        for rows in (256, 512, 1024):
            for name, N, K in DECODE:
                row = []
                for tile, splits in ((2, (1, 2, 4)), (0, (1, 2, 4)), (3, (1, 2, 4, 8))):
                    best = min((m.op_gemm_time(rows, N, K, tile=tile, ksplit=ks, out_kind=5, iters=50), ks) for ks in splits)
                    row.append(f"tile{tile} {best[0] * 1e3:6.1f} us (ks {best[1]}) {2.0 * rows * N * K / best[0] / 1e9:6.1f} TF/s")
                print(f"decode M={rows:4d} {name:8s} N={N:5d} K={K:4d} | " + " | ".join(row), flush=True)
    m.close()


if __name__ == "__main__":
    main()
