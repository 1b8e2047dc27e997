#!/usr/bin/env python3
"""Does running qkv-conv -> attention in sample chunks (qkv buffer reused, staying in the 256 MB infinity cache) beat
one pass over the whole batch?  B=1024, C=128 @16x16 and C=256 @8x8."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import ops  # noqa: E402

imageni2r_amd.set_conv_math("f16x2")
for (C, H) in [(128, 16), (256, 8)]:
    B = 1024
    x = torch.randn(B, C, H, H, device="cuda")
    w = torch.randn(3 * C, C, 1, 1, device="cuda") / C ** 0.5
    b = torch.randn(3 * C, device="cuda")
    wp = ops.conv2d_pack_h2(w)
    heads = C // 64

    def run(chunk):
        outs = []
        for i in range(0, B, chunk):
            qkv = ops.conv2d_fwd_h2(x[i:i + chunk], tuple(w.shape), wp, b, None)
            a, _ = ops.attention_fwd(qkv, heads)
            outs.append(a)
            del qkv
        return outs

    for chunk in (1024, 512, 256, 128, 64):
        for _ in range(3):
            run(chunk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            run(chunk)
        e1.record()
        torch.cuda.synchronize()
        print(f"C{C} H{H} chunk {chunk:5d}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
