#!/usr/bin/env python3
"""A/B of the short Cout tiles (64 / 32 rows) of the 3x3 f16x2 kernel against the 128-row tile on the same operands
(I2R_H2_BM128=1 forces the latter): equal bits, then the time of each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
# B, Cin, Cout, H, W, split_in, res
CASES = [(4096, 128, 12, 16, 16, 1, 0), (1024, 128, 7, 32, 32, 1, 0), (4096, 64, 64, 8, 8, 1, 1),
         (4096, 128, 64, 8, 8, 1, 0), (512, 128, 28, 32, 32, 0, 0), (4096, 128, 40, 16, 16, 0, 1),
         (300, 100, 33, 12, 12, 0, 0), (64, 64, 64, 8, 8, 1, 1), (32, 128, 5, 8, 8, 1, 0), (2048, 14, 64, 8, 8, 0, 0),
         (200, 64, 20, 6, 10, 0, 1)]
bad = 0
for (B, Cin, Cout, H, W, sp, rs) in CASES:
    g = torch.Generator(device=dev).manual_seed(B + Cout)
    x = torch.randn(B, Cin, H, W, device=dev, generator=g)
    if sp:
        grp = 32 if Cin % 32 == 0 else 2
        x = ops.groupnorm_fwd(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), grp, 1e-5, split_out=True)[0]
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device=dev, generator=g)
    res = torch.randn(B, Cout, H, W, device=dev, generator=g) if rs else None
    wph = ops.conv2d_pack_h2(w)
    fn = lambda: ops.conv2d_fwd_h2(x, w.shape, wph, b, res, bool(sp))
    out, ms = {}, {}
    for mode in ("1", "0"):
        if mode == "1":
            os.environ["I2R_H2_BM128"] = "1"
        else:
            os.environ.pop("I2R_H2_BM128", None)
        out[mode] = fn().clone()
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms[mode] = e0.elapsed_time(e1) / 30
    same = torch.equal(out["0"].view(torch.int32), out["1"].view(torch.int32))
    bad += not same
    print(f"B{B} {Cin}->{Cout} {H}x{W} split_in={sp} res={rs}: bits {'EQUAL' if same else 'DIFFER'}  "
          f"128-row {ms['1']*1e3:7.1f} us  short {ms['0']*1e3:7.1f} us  ({ms['1']/ms['0']:.2f}x)", flush=True)
sys.exit(1 if bad else 0)
