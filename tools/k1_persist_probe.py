#!/usr/bin/env python3
"""A/B of the weight-stationary 1x1 kernel (conv1x1_h2_persist_kernel) against the tiled kernel on the same operands:
bits must be equal (same fragments, same MFMA order), then the time of each.

    python tools/k1_persist_probe.py [--iters 50]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = "cuda"
# B, Cin, Cout, H, W, split_in, qkv_out, res
CASES = [(512, 128, 384, 16, 16, 1, 1, 0), (512, 128, 128, 16, 16, 1, 0, 1), (4096, 128, 384, 16, 16, 1, 1, 0),
         (4096, 128, 128, 16, 16, 1, 0, 1), (256, 128, 384, 32, 32, 1, 1, 0), (256, 128, 128, 32, 32, 1, 0, 1),
         (120, 128, 384, 12, 12, 1, 1, 0), (130, 100, 200, 16, 16, 0, 0, 1), (300, 64, 96, 8, 8, 0, 0, 0),
         (67, 128, 130, 16, 16, 0, 0, 0), (512, 128, 128, 16, 16, 0, 0, 0), (130, 100, 200, 16, 16, 0, 0, 0),
         (1024, 128, 384, 16, 16, 1, 1, 0), (128, 128, 384, 32, 32, 1, 1, 0), (65, 128, 384, 16, 16, 1, 1, 0), (256, 128, 384, 16, 16, 1, 1, 0),
         (4096, 128, 256, 8, 8, 0, 0, 0), (1100, 128, 256, 16, 16, 0, 0, 0), (2048, 128, 384, 16, 16, 1, 1, 0)]
bad = 0
for (B, Cin, Cout, H, W, sp, qo, rs) in CASES:
    g = torch.Generator(device=dev).manual_seed(B + Cin)
    x = torch.randn(B, Cin, H, W, device=dev, generator=g)
    if sp:
        x = ops.groupnorm_fwd(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-5,
                              split_out=True)[0]  # the (hi, lo) pair format the GroupNorm feeds these convs
    w = torch.randn(Cout, Cin, 1, 1, device=dev, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, device=dev, generator=g)
    res = torch.randn(B, Cout, H, W, device=dev, generator=g) if rs else None
    wph = ops.conv2d_pack_h2(w)
    fn = lambda: ops.conv2d_fwd_h2(x, w.shape, wph, b, res, bool(sp), None, bool(qo))
    out, ms = {}, {}
    for mode in ("0", "1"):
        os.environ["I2R_H2_PERSIST"] = mode
        out[mode] = fn().clone()
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms[mode] = e0.elapsed_time(e1) / a.iters
    same = torch.equal(out["0"].view(torch.int32), out["1"].view(torch.int32))
    bad += not same
    print(f"B{B} {Cin}->{Cout} {H}x{W} split_in={sp} qkv_out={qo} res={rs}: bits {'EQUAL' if same else 'DIFFER'}  "
          f"tiled {ms['0']*1e3:7.1f} us  persist {ms['1']*1e3:7.1f} us  ({ms['0']/ms['1']:.2f}x)", flush=True)
sys.exit(1 if bad else 0)
