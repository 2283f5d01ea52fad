#!/usr/bin/env python3
"""Times the 256x256 kernel on the prefill / encoder shapes of the 1.7B path (512 clips).  Q3LIB=<path to a build of
libq3asr.so> times another build (kernel experiments are compared build against build in one GPU session).

  python scripts/bench_gemm256.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))

import q3asr.lib  # noqa: E402
from q3asr import Qwen3ASRModel, make_config, synth  # noqa: E402

if os.environ.get("Q3LIB"):
    q3asr.lib.LIB_PATH = os.path.abspath(os.environ["Q3LIB"])

OUT = dict(bf16=0, f32=1, acc=2, swiglu=3, argmax=4, slab=5)
R, E = 512 * 405, 512 * 390
SHAPES = [
    ("dec qkv", R, 4096, 2048, "bf16"), ("dec o", R, 2048, 2048, "acc"), ("dec gate_up", R, 12288, 2048, "swiglu"),
    ("dec down", R, 2048, 6144, "acc"), ("enc qkv", E, 3072, 1024, "bf16"), ("enc o", E, 1024, 1024, "acc"),
    ("enc fc1", E, 4096, 1024, "bf16"), ("enc fc2", E, 1024, 4096, "acc"), ("lm head", 2048, 151936, 2048, "argmax"),
]


def main():
    cfg = make_config("tiny")
    m = Qwen3ASRModel.from_checkpoint(cfg, synth.make_checkpoint(cfg, seed=0))
    tot = 0.0
    cells = []
    for name, M, N, K, out in SHAPES:
        ms = min(m.op_gemm_time(M, N, K, tile=3, out_kind=OUT[out], iters=5) for _ in range(3))
        tot += ms
        cells.append(f"{name} {2.0 * M * N * K / ms / 1e9:6.0f}")
    print(f"{os.path.basename(os.environ.get('Q3LIB', 'libq3asr.so')):22s} sum {tot:7.3f} ms | " + " | ".join(cells), flush=True)
    m.close()


if __name__ == "__main__":
    main()
