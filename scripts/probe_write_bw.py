#!/usr/bin/env python3
"""Write-only and read-only streaming rates of the GPU (torch fill / sum over 4 GiB): the practical ceilings next to
which the store-bound kernels (conv1, the log-mel output) are read."""
import torch

n = 1 << 30
x = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn, nbytes in (("fill (write only)", lambda: x.fill_(1.0), 4 * n), ("zero_ (write only)", lambda: x.zero_(), 4 * n),
                         ("sum (read only)", lambda: x.sum(), 4 * n), ("copy (read + write)", lambda: x[: n // 2].copy_(x[n // 2:]), 4 * n)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:22s} {nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12:.2f} TB/s", flush=True)
