#!/usr/bin/env python3
"""Stand-alone check of attn_fwd_small_kernel (fp32 attention of the levels with T <= 32 tokens) on PADDED buffers:
guards untouched, result against the fp64 softmax attention of the same operands (2e-6), log-sum-exp rows likewise;
then timed.  usage: attn_small_probe.py [B heads]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
B, heads = (int(v) for v in (sys.argv[1:3] if len(sys.argv) >= 3 else (4096, 4)))
GUARD, SENT = 1 << 16, 4321.0
C = 64 * heads
for (H, W) in ((1, 1), (2, 2), (3, 3), (4, 4), (2, 4), (5, 5), (4, 8)):
    T = H * W
    gen = torch.Generator().manual_seed(T)
    qkv_h = torch.randn(B, 3 * C, H, W, generator=gen) * 1.5
    big = torch.full((qkv_h.numel() + 2 * GUARD,), SENT, device=dev)
    qkv = big[GUARD: GUARD + qkv_h.numel()].view(qkv_h.shape)
    qkv.copy_(qkv_h)
    a, lse = ops.attention_fwd(qkv, heads, save_lse=True)
    torch.cuda.synchronize()
    ok = bool((big[:GUARD] == SENT).all()) and bool((big[GUARD + qkv_h.numel():] == SENT).all()) and torch.equal(qkv.cpu(), qkv_h)
    r = qkv.double().view(B, heads, 64, 3, T)
    q, k, v = r[:, :, :, 0], r[:, :, :, 1], r[:, :, :, 2]
    s = torch.einsum("bhdq,bhdk->bhqk", q, k) * 0.125
    p = torch.softmax(s, dim=-1)
    ref = torch.einsum("bhqk,bhdk->bhdq", p, v).reshape(B, C, H, W)
    ref_lse = torch.logsumexp(s, dim=-1).reshape(B * heads, T)
    err = float((a.double() - ref).abs().max() / ref.abs().max())
    err_l = float((lse.double().view(B * heads, T) - ref_lse).abs().max())
    for _ in range(3):
        ops.attention_fwd(qkv, heads)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention_fwd(qkv, heads)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = 4.0 * B * C * T * 4
    print(f"T={T:3d} ({H}x{W}) B{B} heads{heads}: err {err:.2e}  lse err {err_l:.2e}  guards intact {ok}   {us:.1f} us  "
          f"{nbytes / us * 1e-6:.2f} TB/s", flush=True)
    assert ok and err < 2e-6 and err_l < 1e-5, "FAILED"
print("probe ok")
