#!/usr/bin/env python3
"""Follow-up of r3_det_victim.py: is the T=16 attention's INPUT changed by the attention loop on the other stream (an
out-of-bounds write of the non-presplit f16x2 attention launch), or only its result while the two run together?"""
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
x = torch.randn(2048, 768, 4, 4, device=dev, generator=g)
x_copy = x.clone()
noise_copy = noise_qkv.clone()
ns = torch.cuda.Stream()
ref = ops.attention_fwd(x, 4)[0].clone()
torch.cuda.synchronize()
for mode in ("noise = f16x2 attention T=256 (not presplit)", "noise = fp32 small attention T=16", "noise = torch matmul"):
    ns.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ns):
        for _ in range(60):
            if mode.startswith("noise = f16x2"):
                ops.attention_fwd(noise_qkv, 2)
            elif "small" in mode:
                ops.attention_fwd(x_copy, 4)
            else:
                torch.mm(noise_qkv.view(2048 * 384, 256)[:8192], noise_qkv.view(2048 * 384, 256)[:8192].t())
    outs = [ops.attention_fwd(x, 4)[0] for _ in range(8)]
    torch.cuda.synchronize()
    nbad = sum(0 if torch.equal(o, ref) else 1 for o in outs)
    after = ops.attention_fwd(x, 4)[0]
    torch.cuda.synchronize()
    print(f"{mode}: results that differ while running together {nbad} of 8; input unchanged {torch.equal(x, x_copy)}; "
          f"noise input unchanged {torch.equal(noise_qkv, noise_copy)}; result alone afterwards equal {torch.equal(after, ref)}", flush=True)
    if nbad:
        o = next(o for o in outs if not torch.equal(o, ref))
        d = (o - ref).abs().flatten(1).max(1).values
        bad = (d > 0).nonzero().flatten()
        print(f"   samples that differ {bad.numel()} of {x.shape[0]}: first {bad[:12].tolist()} last {bad[-6:].tolist()}  largest {float(d.max()):.3e}", flush=True)
