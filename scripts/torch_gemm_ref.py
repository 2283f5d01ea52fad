#!/usr/bin/env python3
"""What the vendor GEMM library (through torch.nn.functional.linear) reaches on the decode projections' shapes:
a reference point for the hand-written kernels, not part of the product."""
import torch

shapes = [("qkv", 4096, 2048), ("o", 2048, 2048), ("gate_up", 12288, 2048), ("down", 2048, 6144), ("lm_head", 151936, 2048)]
for rows in (256, 512, 1024):
    for name, n, k in shapes:
        x = torch.randn(rows, k, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02
        for _ in range(5):
            y = torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            y = torch.nn.functional.linear(x, w)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"rows {rows:5d} {name:8s} N={n:6d} K={k:5d}: {us:8.1f} us  {2.0 * rows * n * k / us / 1e6:7.1f} TF/s", flush=True)
