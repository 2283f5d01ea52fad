#!/usr/bin/env python3
"""Vendor GEMM library (through torch) on the prefill / encoder shapes of 512 clips: reference point only."""
import torch

R, E = 512 * 405, 512 * 390
shapes = [("dec qkv", R, 4096, 2048), ("dec o", R, 2048, 2048), ("dec gate_up", R, 12288, 2048), ("dec down", R, 2048, 6144),
          ("enc qkv", E, 3072, 1024), ("enc fc1", E, 4096, 1024), ("enc fc2", E, 1024, 4096)]
for name, m, n, k in shapes:
    x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02
    for _ in range(2):
        y = torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        y = torch.nn.functional.linear(x, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:12s} M={m:6d} N={n:6d} K={k:5d}: {ms:8.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s", flush=True)
    del x, w, y
