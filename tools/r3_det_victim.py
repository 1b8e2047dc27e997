#!/usr/bin/env python3
"""Which kernel family changes its results when another stream runs the attention kernel beside it?  Every candidate
launch is repeated with the attention loop running on a second stream and compared, bit for bit, with its result alone."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
imageni2r_amd.set_conv_math("f16x2")
g = torch.Generator(device=dev).manual_seed(3)
R = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
B = 2048
noise_qkv = R(2048, 384, 16, 16)
ns = torch.cuda.Stream()


def gn_split(x):
    C = x.shape[1]
    return ops.groupnorm_fwd(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5, act=True, split_out=True)[0]


cands = {}
x128 = R(B, 128, 16, 16); xs128 = gn_split(x128)
x256 = R(B, 256, 8, 8); xs256 = gn_split(x256)
x256_4 = R(B, 256, 4, 4); xs256_4 = gn_split(x256_4)
x512_2 = R(B, 512, 2, 2); xs512_2 = gn_split(x512_2)
w = lambda co, ci, k: ops.conv2d_pack_h2(R(co, ci, k, k) / np.sqrt(ci * k * k))  # noqa: E731
w3_128, w3_256, w3_512 = w(128, 128, 3), w(256, 256, 3), w(512, 512, 3)
w1_qkv128, w1_qkv256, w1_p128, w1_p256, w1_qkv512, w1_p512 = w(384, 128, 1), w(768, 256, 1), w(128, 128, 1), w(256, 256, 1), w(1536, 512, 1), w(512, 512, 1)
ss128 = R(B, 256) * 0.3
cands["gn 128@16 split"] = lambda: gn_split(x128)
cands["gn 128@16 ss"] = lambda: ops.groupnorm_fwd(x128, torch.ones(128, device=dev), torch.zeros(128, device=dev), 32, 1e-5, ss=ss128, act=True, split_out=True)[0]
cands["gn 256@8"] = lambda: gn_split(x256)
cands["gn 256@4 (small)"] = lambda: gn_split(x256_4)
cands["gn 512@2 (small)"] = lambda: gn_split(x512_2)
cands["conv3 128@16 res (m16)"] = lambda: ops.conv2d_fwd_h2(xs128, (128, 128, 3, 3), w3_128, None, x128, x_split=True)
cands["conv3 256@8 res (m16)"] = lambda: ops.conv2d_fwd_h2(xs256, (256, 256, 3, 3), w3_256, None, x256, x_split=True)
cands["conv3 256@4 res (m16, 256 tiles)"] = lambda: ops.conv2d_fwd_h2(xs256_4, (256, 256, 3, 3), w3_256, None, x256_4, x_split=True)
cands["conv3 512@2 res (m16 rows of 2)"] = lambda: ops.conv2d_fwd_h2(xs512_2, (512, 512, 3, 3), w3_512, None, x512_2, x_split=True)
cands["conv1 qkv 128@16 (ws)"] = lambda: ops.conv2d_fwd_h2(xs128, (384, 128, 1, 1), w1_qkv128, None, None, x_split=True, qkv_out=True)
cands["conv1 qkv 256@8 (ws)"] = lambda: ops.conv2d_fwd_h2(xs256, (768, 256, 1, 1), w1_qkv256, None, None, x_split=True, qkv_out=True)
cands["conv1 qkv 256@4 (ws)"] = lambda: ops.conv2d_fwd_h2(xs256_4, (768, 256, 1, 1), w1_qkv256, None, None, x_split=True, qkv_out=True)
cands["conv1 proj 128@16 res (m16 k1)"] = lambda: ops.conv2d_fwd_h2(x128, (128, 128, 1, 1), w1_p128, None, x128)
cands["conv1 proj 256@8 res (m16 k1)"] = lambda: ops.conv2d_fwd_h2(x256, (256, 256, 1, 1), w1_p256, None, x256)
cands["conv1 qkv 512@2 (tiled/m16 k1)"] = lambda: ops.conv2d_fwd_h2(xs512_2, (1536, 512, 1, 1), w1_qkv512, None, None, x_split=True, qkv_out=True)
cands["conv1 proj 512@2 res (tiled)"] = lambda: ops.conv2d_fwd_h2(x512_2, (512, 512, 1, 1), w1_p512, None, x512_2)
qkv_pre = ops.conv2d_fwd_h2(xs128, (384, 128, 1, 1), w1_qkv128, None, None, x_split=True, qkv_out=True)
qkv_pre8 = ops.conv2d_fwd_h2(xs256, (768, 256, 1, 1), w1_qkv256, None, None, x_split=True, qkv_out=True)
cands["attention T=256 presplit"] = lambda: ops.attention_fwd(qkv_pre, 2, presplit=True)[0]
cands["attention T=64 presplit"] = lambda: ops.attention_fwd(qkv_pre8, 4, presplit=True)[0]
cands["attention T=16 f32"] = lambda: ops.attention_fwd(R(B, 768, 4, 4) if False else x768_4, 4)[0]
x768_4 = R(B, 768, 4, 4)
cands["attention T=4 f32"] = lambda: ops.attention_fwd(x1536_2, 8)[0]
x1536_2 = R(B, 1536, 2, 2)
cands["down2"] = lambda: ops.Down2Fn.apply(x128)
cands["pack h2 3x3"] = lambda: ops.conv2d_pack_h2(wraw)
wraw = R(256, 256, 3, 3)

torch.cuda.synchronize()
for name, fn in cands.items():
    ref = fn().clone()
    torch.cuda.synchronize()
    nbad, worst = 0, 0.0
    for rep in range(6):
        ns.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ns):
            for _ in range(60):
                ops.attention_fwd(noise_qkv, 2)
        outs = [fn() for _ in range(8)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o.view(torch.int32) if o.dtype == torch.float32 else o, ref.view(torch.int32) if ref.dtype == torch.float32 else ref):
                nbad += 1
                d = (o.float() - ref.float()).abs()
                worst = max(worst, float(d[torch.isfinite(d)].max()) if o.dtype == torch.float32 else 1.0)
    print(f"{name:40s} differing results {nbad:2d} of 48   largest difference {worst:.3e}", flush=True)
