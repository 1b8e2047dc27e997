#!/usr/bin/env python3
"""The training kernels beside another stream's attention loop (the check that found the store-data hazard of the small
attention kernel, tools/r3_det_victim.py): every deterministic output must repeat to the bit."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(11)
R = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
noise = R(2048, 384, 16, 16)
ns = torch.cuda.Stream()
B = 32
ops.set_deterministic(True)
cands = {}


def conv_case(name, Cin, Cout, H, k):
    x, w, b = R(B, Cin, H, H), R(Cout, Cin, k, k) / np.sqrt(Cin * k * k), R(Cout)
    dy = R(B, Cout, H, H)
    wp = ops.conv2d_pack(w)
    cands[f"conv fwd f32 {name}"] = lambda: ops.conv2d_fwd(x, w, b, None, wp)
    cands[f"conv dgrad f32 {name}"] = lambda: ops.conv2d_dgrad(dy, w)
    cands[f"conv wgrad (det) {name}"] = lambda: ops.conv2d_wgrad(x, dy, k, True)[0]
    if H >= 8:   # the f16x2 data gradient of the high-resolution levels
        wpd = ops.conv2d_pack_h2_dgrad(w)
        cands[f"conv dgrad f16x2 {name}"] = lambda: ops.conv2d_dgrad_h2(dy, tuple(w.shape), wpd)


for (name, Cin, Cout, H, k) in (("128->128 3x3 @16", 128, 128, 16, 3), ("256->256 3x3 @8", 256, 256, 8, 3), ("512->512 3x3 @2", 512, 512, 2, 3),
                                 ("256->256 1x1 @8", 256, 256, 8, 1), ("128->384 1x1 @16", 128, 384, 16, 1)):
    try:
        conv_case(name, Cin, Cout, H, k)
    except Exception as e:  # an op whose python signature differs: skip, say so
        print("skipped", name, type(e).__name__, e)
for (C, H) in ((128, 16), (256, 8), (256, 4), (512, 2)):
    x, dy, add = R(B, C, H, H), R(B, C, H, H), R(B, C, H, H)
    w, b, ss = R(C), R(C), R(B, 2 * C) * 0.3
    _, mean, rstd = ops.groupnorm_fwd(x, w, b, 32, 1e-5, ss=ss, act=True, save_stats=True)
    cands[f"gn fwd train {C}@{H}"] = (lambda x=x, w=w, b=b, ss=ss: ops.groupnorm_fwd(x, w, b, 32, 1e-5, ss=ss, act=True, save_stats=True)[0])
    cands[f"gn bwd dx {C}@{H}"] = (lambda x=x, w=w, b=b, ss=ss, mean=mean, rstd=rstd, dy=dy, add=add: ops.groupnorm_bwd(x, w, b, ss, mean, rstd, dy, 32, True, add=add)[0])
for (heads, H) in ((2, 16), (4, 8), (4, 4), (8, 2)):
    qkv, da = R(B, 192 * heads, H, H), R(B, 64 * heads, H, H)
    a, lse = ops.attention_fwd(qkv, heads, save_lse=True)
    cands[f"attention fwd+lse h{heads}@{H}"] = (lambda qkv=qkv, heads=heads: ops.attention_fwd(qkv, heads, save_lse=True)[0])
    cands[f"attention bwd h{heads}@{H}"] = (lambda qkv=qkv, a=a, lse=lse, da=da, heads=heads: ops.attention_bwd(qkv, a, lse, da, heads))
xl, wl, bl = R(B, 512), R(2048, 512) / 22.0, R(2048)
cands["linear fwd 512->2048"] = lambda: ops.linear_fwd(xl, wl, bl, act=True)[0]
dyl = R(B, 2048)
cands["linear bwd dx 512->2048"] = lambda: ops.linear_bwd(xl, wl, dyl, need_dw=False)[0]

torch.cuda.synchronize()
for name, fn in cands.items():
    try:
        ref = fn().clone()
    except Exception as e:
        print(f"{name:36s} skipped ({type(e).__name__}: {str(e)[:60]})", flush=True)
        continue
    torch.cuda.synchronize()
    again = fn()
    torch.cuda.synchronize()
    alone_ok = torch.equal(again, ref)
    nbad, worst = 0, 0.0
    for rep in range(4):
        ns.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ns):
            for _ in range(40):
                ops.attention_fwd(noise, 2)
        outs = [fn() for _ in range(12)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                nbad += 1
                worst = max(worst, float((o - ref).abs().max()))
    print(f"{name:36s} repeats alone {alone_ok};  beside the attention loop: differing results {nbad:2d} of 48, largest difference {worst:.3e}", flush=True)
