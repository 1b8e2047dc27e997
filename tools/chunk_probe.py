#!/usr/bin/env python3
"""Attention tail of a UNetBlock (norm2 -> qkv conv -> attention -> proj conv + residual) over the whole batch vs in
sample chunks whose intermediates stay in the 256 MB infinity cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import ops  # noqa: E402

imageni2r_amd.set_conv_math("f16x2")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for (C, H) in [(128, 16), (256, 8)]:
    x = torch.randn(B, C, H, H, device="cuda")
    gw, gb = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    w = torch.randn(3 * C, C, 1, 1, device="cuda") / C ** 0.5
    b = torch.randn(3 * C, device="cuda")
    wpr = torch.randn(C, C, 1, 1, device="cuda") / C ** 0.5
    bpr = torch.randn(C, device="cuda")
    wp, wpp = ops.conv2d_pack_h2(w), ops.conv2d_pack_h2(wpr)
    heads = C // 64
    out = torch.empty_like(x)

    def run(chunk):
        for i in range(0, B, chunk):
            xc = x[i:i + chunk]
            xn, _, _ = ops.groupnorm_fwd(xc, gw, gb, 32, 1e-5, split_out=True)
            qkv = ops.conv2d_fwd_h2(xn, tuple(w.shape), wp, b, None, True)
            a, _ = ops.attention_fwd(qkv, heads)
            out[i:i + chunk] = ops.conv2d_fwd_h2(a, tuple(wpr.shape), wpp, bpr, xc)

    for chunk in (B, 1024, 512, 256, 128):
        if chunk > B:
            continue
        for _ in range(3):
            run(chunk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            run(chunk)
        e1.record()
        torch.cuda.synchronize()
        print(f"B{B} C{C} H{H} chunk {chunk:5d}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
