#!/usr/bin/env python3
"""Micro-benchmark of the f16x2 inference attention kernel (also the target of rocprofv3 --pmc passes).
usage: attn_bench.py [B heads H [iters]]   -- qkv (B, 3*64*heads, H, H), T = H*H"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

B, heads, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (128, 2, 32)))
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
T, C = H * H, 64 * heads
qkv = torch.randn(B, 3 * C, H, H, device="cuda")
hi = qkv.half()
lo = (qkv - hi.float()).half()
pairs = ((hi.view(torch.int16).int() & 0xFFFF) | (lo.view(torch.int16).int() << 16)).view(torch.float32)
for presplit in (False, True):
    src = pairs if presplit else qkv
    for _ in range(3):
        out = ops.attention_fwd(src, heads, presplit=presplit)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = ops.attention_fwd(src, heads, presplit=presplit)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    flop = 4.0 * B * heads * T * T * 64
    print(f"attention presplit={presplit} B{B} heads{heads} T{T}: {us:.1f} us  {flop / us * 1e-6:.1f} TF algorithmic", flush=True)
