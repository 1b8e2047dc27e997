#!/usr/bin/env python3
"""HBM bandwidth probe: pure write (fill), pure read (sum), copy -- the ceilings the byte-bound kernels are priced on."""
import torch

n = 1 << 28  # 1 GiB of fp32
x = torch.empty(n, device="cuda")
y = torch.empty(n, device="cuda")


def t(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


gb = n * 4 / 1e12
print(f"fill  (write only): {gb / t(lambda: x.fill_(1.0)):.2f} TB/s")
print(f"sum   (read only) : {gb / t(lambda: x.sum()):.2f} TB/s")
print(f"copy  (1R + 1W)   : {2 * gb / t(lambda: y.copy_(x)):.2f} TB/s")
print(f"add   (2R + 1W)   : {3 * gb / t(lambda: torch.add(x, y, out=y)):.2f} TB/s")
