#!/usr/bin/env python3
"""Stand-alone check of the f16x2 inference attention kernel on guard-padded buffers (run before pytest whenever the
kernel's addressing changes): qkv and out sit inside sentinel-filled allocations; guards and the input must come back
untouched and the result must match an fp64 softmax attention.  All three wave counts (T % 128, % 64, % 32), both
input forms (fp32, pre-split pairs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

dev, GUARD, SENT = "cuda", 1 << 16, 777.0


def padded(t):
    n = (t.numel() + 63) // 64 * 64
    big = torch.full((n + 2 * GUARD,), SENT, dtype=torch.float32, device=dev)
    v = big[GUARD: GUARD + t.numel()].view(t.shape)
    v.copy_(t)
    return v, big, n


def guards_ok(big, n):
    return bool((big[:GUARD] == SENT).all()) and bool((big[GUARD + n:] == SENT).all())


for (B, heads, H, W) in [(3, 2, 8, 8), (2, 2, 16, 8), (2, 4, 16, 16), (1, 2, 32, 32), (5, 1, 12, 8), (2, 8, 8, 8)]:
    T, C = H * W, 64 * heads
    gen = torch.Generator().manual_seed(T + heads)
    qkv = torch.randn(B, 3 * C, H, W, generator=gen)
    q, k, v = qkv.double().reshape(B * heads, 64, 3, T).unbind(2)
    w = torch.softmax(torch.einsum("ncq,nck->nqk", q, k / 8.0), dim=2)
    ref = torch.einsum("nqk,nck->ncq", w, v).reshape(B, C, H, W)
    for presplit in (False, True):
        src = qkv.to(dev)
        if presplit:
            sc = src.clone().view(B * heads, 64, 3, T)
            sc[:, :, 0] *= ops.qscale() if hasattr(ops, "qscale") else 0.125 * 1.4426950408889634
            sc = sc.view_as(src)
            hi = sc.half()
            lo = (sc - hi.float()).half()
            src = ((hi.view(torch.int16).int() & 0xFFFF) | (lo.view(torch.int16).int() << 16)).view(torch.float32)
        xin, xin_big, xn = padded(src)
        out, out_big, on = padded(torch.zeros(B, C, H, W))
        keep = xin.clone()
        torch.cuda.synchronize()
        ops._lib.call("i2r_attention_fwd_h2", ops._p(xin), ops._p(out), B, heads, T, int(presplit), ops._stream())
        torch.cuda.synchronize()
        ok = guards_ok(xin_big, xn) and guards_ok(out_big, on) and torch.equal(keep.view(torch.int32), xin.view(torch.int32))
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        print(f"B{B} heads{heads} T{T} presplit={presplit}: rel err {err:.2e}  guards/input intact {ok}", flush=True)
        assert ok and err < 5e-6, "FAILED"
print("probe ok")
