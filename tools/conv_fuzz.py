#!/usr/bin/env python3
"""Random-shape check of the f16x2 inference conv (every dispatch branch: tiled 128 / 64 / 32-row tiles, DPP and gather
3x3, split-K, the persistent 1x1 loop, pair-format input, residual, upsampled input / residual, qkv output) against
torch in fp64, with the output placed inside a guard band so that a write outside it shows.

    python tools/conv_fuzz.py [cases=300] [seed=0]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"
GUARD = 4096
worst, bad = 0.0, 0
for it in range(N):
    k = int(rng.choice([1, 3]))
    big = rng.random() < 0.25
    H = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 32]))
    W = H if rng.random() < 0.7 else int(rng.choice([2, 4, 5, 8, 10, 16, 32]))
    B = int(rng.integers(1, 40)) if not big else int(rng.integers(200, 1200))
    if big and H * W > 256:
        B = max(1, B // 4)
    Cin = int(rng.choice([3, 12, 32, 64, 96, 100, 128, 192, 256, 320]))
    Cout = int(rng.choice([3, 5, 12, 28, 33, 64, 96, 128, 130, 192, 256, 384]))
    sp = Cin % 32 == 0 and rng.random() < 0.5 and B * H * W > 1   # (torch's group_norm reference refuses one value per channel)
    rs = rng.random() < 0.4
    qo = Cout % 3 == 0 and not rs and rng.random() < 0.3
    up_in = k == 3 and sp and rng.random() < 0.15 and W in (2, 4, 8, 16) and H == W
    up_res = rs and not up_in and H % 2 == 0 and W % 2 == 0 and rng.random() < 0.2
    g = torch.Generator(device=dev).manual_seed(1000 + it)
    Hi, Wi = (H, W)
    if up_in:
        H, W = 2 * H, 2 * W
    x0 = torch.randn(B, Cin, Hi, Wi, device=dev, generator=g)
    x, xr = x0, x0.double()
    if sp:
        x = ops.groupnorm_fwd(x0, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-5, split_out=True)[0]
        xr = torch.nn.functional.group_norm(x0.double(), 32, eps=1e-5)
    if up_in:
        xr = xr.repeat_interleave(2, 2).repeat_interleave(2, 3)
    w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (k * k * Cin) ** 0.5
    b = torch.randn(Cout, device=dev, generator=g) if rng.random() < 0.8 else None
    res = None
    if rs:
        res = torch.randn(B, Cout, H // 2 if up_res else H, W // 2 if up_res else W, device=dev, generator=g)
    wph = ops.conv2d_pack_h2(w)
    n_out = B * Cout * H * W
    band = torch.full((n_out + 2 * GUARD,), 777.0, device=dev)
    out = band[GUARD:GUARD + n_out].view(B, Cout, H, W)
    ops.conv2d_fwd_h2(x, w.shape, wph, b, res, bool(sp), out, bool(qo), bool(up_in), bool(up_res))
    ok_guard = bool((band[:GUARD] == 777.0).all()) and bool((band[GUARD + n_out:] == 777.0).all())
    ref = torch.nn.functional.conv2d(xr, w.double(), None if b is None else b.double(), padding=k // 2)
    if rs:
        r = res.double()
        ref = ref + (r.repeat_interleave(2, 2).repeat_interleave(2, 3) if up_res else r)
    got = out.double()
    if qo:  # split pairs, q rows pre-scaled: undo both
        bits = out.view(torch.int32)
        hi = (bits & 0xFFFF).to(torch.int16).view(torch.float16).double()
        lo = ((bits >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double()
        got = hi + lo
        qs = torch.ones(Cout, device=dev, dtype=torch.float64)
        qs[0::3] = 0.125 * 1.4426950408889634
        ref = ref * qs.view(1, -1, 1, 1)
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    worst = max(worst, err)
    flag = "" if (err < 2e-5 and ok_guard) else "   <-- FAIL"
    bad += bool(flag)
    if flag or it % 25 == 0:
        print(f"{it:4d} k{k} B{B} {Cin}->{Cout} {H}x{W} split={int(sp)} res={int(rs)} qkv={int(qo)} up_in={int(up_in)} "
              f"up_res={int(up_res)} bias={int(b is not None)}: err {err:.2e} guard {'ok' if ok_guard else 'WRITTEN'}{flag}",
              flush=True)
print(f"{N} cases, worst error {worst:.2e}, failures {bad}")
sys.exit(1 if bad else 0)
