#!/usr/bin/env python3
"""Micro-benchmark of the fused GroupNorm kernels (achieved HBM GB/s; also the target of rocprofv3 --pmc passes).

    python tools/gn_bench.py [--batch 1024] [--iters 20] [--bwd [--acc] [--add]]
--acc: dgamma / dbeta accumulated into given slots with atomics (the training step's form) instead of per-sample
partials + a row reduction; --add: with the bypass gradient added to dx.  At small batches the loop is host-bound: read
the kernel times from `rocprofv3 --kernel-trace --stats`.  Shapes = the GroupNorm sites of the B1/C network: (C, H)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--bwd", action="store_true")
ap.add_argument("--acc", action="store_true")
ap.add_argument("--add", action="store_true")
a = ap.parse_args()
dev = "cuda"
for (C, H) in [(128, 16), (256, 16), (384, 16), (256, 8), (512, 8), (256, 4), (512, 2), (1024, 2), (128, 8), (512, 4)]:
    B = a.batch
    x = torch.randn(B, C, H, H, device=dev)
    w, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    ss = torch.randn(B if a.bwd else 1, 2 * C, device=dev)
    dy = torch.randn_like(x)
    if a.bwd:
        _, mean, rstd = ops.groupnorm_fwd(x, w, b, 32, 1e-5, ss=ss, act=True, save_stats=True)
        dg, db = (torch.zeros(C, device=dev), torch.zeros(C, device=dev)) if a.acc else (None, None)
        ops.set_slots_zeroed(a.acc)
        addt = torch.randn_like(x) if a.add else None
        fn = lambda: ops.groupnorm_bwd(x, w, b, ss, mean, rstd, dy, 32, True, dgamma=dg, dbeta=db, add=addt)
        bytes_per_elem = 16 if a.add else 12
    else:
        fn = lambda: ops.groupnorm_fwd(x, w, b, 32, 1e-5, ss=ss, act=True, split_out=True)
        bytes_per_elem = 8
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    gb = x.numel() * bytes_per_elem / 1e9
    print(f"gn {'bwd' if a.bwd else 'fwd'} B{B} C{C} H{H}: {ms*1e3:8.1f} us  {gb/ms*1e3:7.0f} GB/s (algorithmic {bytes_per_elem} B/elem)",
          flush=True)
