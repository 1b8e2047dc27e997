#!/usr/bin/env python3
"""Follow-up: which forms of attn_fwd_small_kernel give other results beside the f16x2 attention loop?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
imageni2r_amd.set_conv_math("f16x2")
g = torch.Generator(device=dev).manual_seed(3)
noise_qkv = torch.randn(2048, 384, 16, 16, device=dev, generator=g)
ns = torch.cuda.Stream()
for (H, W, heads, save_lse) in ((4, 4, 4, False), (4, 4, 4, True), (3, 4, 4, False), (3, 3, 4, False), (2, 4, 4, False), (2, 2, 8, False), (4, 8, 2, False), (1, 1, 8, False)):
    x = torch.randn(2048, 192 * heads, H, W, device=dev, generator=g)
    ref = ops.attention_fwd(x, heads, save_lse=save_lse)[0].clone()
    torch.cuda.synchronize()
    ns.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ns):
        for _ in range(60):
            ops.attention_fwd(noise_qkv, 2)
    outs = [ops.attention_fwd(x, heads, save_lse=save_lse)[0] for _ in range(8)]
    torch.cuda.synchronize()
    nbad = sum(0 if torch.equal(o, ref) else 1 for o in outs)
    worst = max(float((o - ref).abs().max()) for o in outs)
    print(f"T={H * W:2d} ({H}x{W}) heads {heads} lse {int(save_lse)}: results that differ beside the attention loop {nbad} of 8, largest difference {worst:.3e}", flush=True)
