#!/usr/bin/env python3
"""Stand-alone check of the single-term (plain f16) 3x3 conv on PADDED buffers: activations, packed weights, residual and
output all sit in the middle of larger guard-filled allocations; after each launch the guards must be untouched and the
result must equal the fp64 conv of the rounded operands.  Run this (after tools/h2_loader_check.py) before pytest
whenever the kernel's loader or epilogue indexing changes.  One launch at a time, each synchronised and reported."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
GUARD = 1 << 16  # floats either side
SENT = 12345.0


def padded(t):
    """Copy `t` into the middle of a sentinel-filled buffer; returns (view like t, whole buffer)."""
    n = (t.numel() + 63) // 64 * 64
    big = torch.full((n + 2 * GUARD,), SENT, dtype=torch.float32, device=dev)
    view = big[GUARD: GUARD + t.numel()].view(t.shape)
    view.copy_(t)
    return view, big, t.numel()


def guards_ok(big, n):
    return bool((big[:GUARD] == SENT).all()) and bool((big[GUARD + (n + 63) // 64 * 64:] == SENT).all())


for mode in ("f16", "f16x2"):
    imageni2r_amd.set_conv_math(mode)
    for (B, Cin, Cout, H) in [(2, 128, 128, 16), (3, 96, 160, 4), (2, 1024, 512, 2), (2, 128, 12, 16), (260, 128, 256, 16),
                              (2, 64, 128, 32), (1, 48, 40, 8)]:
        gen = torch.Generator().manual_seed(B + Cin + Cout + H)
        x = torch.randn(B, Cin, H, H, generator=gen) * 2.0 + 0.5
        w = torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9)
        b = torch.randn(Cout, generator=gen)
        res = torch.randn(B, Cout, H, H, generator=gen)
        G = min(32, Cin // 4)
        gw, gb = torch.ones(Cin), torch.zeros(Cin)
        ysp, _, _ = ops.groupnorm_fwd(x.to(dev), gw.to(dev), gb.to(dev), G, 1e-5, act=True, split_out=True)
        wp = ops.conv2d_pack_h2(w.to(dev))
        xs, xs_big, xs_n = padded(ysp)
        GB = 1 << 18  # guard bytes either side of the packed weights
        wp_big = torch.full((wp.numel() + 2 * GB,), 0xAB, dtype=torch.uint8, device=dev)
        wps = wp_big[GB: GB + wp.numel()]
        wps.copy_(wp)
        rs, rs_big, rs_n = padded(res.to(dev))
        out, out_big, out_n = padded(torch.zeros(B, Cout, H, H))
        torch.cuda.synchronize()
        y = ops.conv2d_fwd_h2(xs, tuple(w.shape), wps, b.to(dev), rs, x_split=True, out=out)
        torch.cuda.synchronize()
        ok = all(guards_ok(bg, n) for bg, n in ((xs_big, xs_n), (rs_big, rs_n), (out_big, out_n)))
        ok = ok and bool((wp_big[:GB] == 0xAB).all()) and bool((wp_big[GB + wp.numel():] == 0xAB).all())
        ok = ok and torch.equal(wps, wp) and torch.equal(xs.view(torch.int32), ysp.view(torch.int32))  # inputs unmodified
        words = ysp.view(torch.int32).cpu()
        x_hi = (words & 0xFFFF).to(torch.int16).view(torch.float16).double()
        x_lo = ((words >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double()
        if mode == "f16":
            ref = torch.nn.functional.conv2d(x_hi, (w * 64.0).half().double() / 64.0, b.double(), padding=1) + res.double()
        else:
            ref = torch.nn.functional.conv2d(x_hi + x_lo, w.double(), b.double(), padding=1) + res.double()
        err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
        print(f"{mode} B{B} {Cin}->{Cout} @{H}: rel err vs rounded-operand conv {err:.2e}  guards intact {ok}", flush=True)
        assert ok and err < 5e-6, "FAILED"
imageni2r_amd.set_conv_math("f16x2")
print("probe ok")
