#!/usr/bin/env python3
"""Times the 3x3 weight gradient at the training shapes of B1 / D (batch 32) with the halo-tile kernel on and off
(I2R_WGRAD_TILE, read per call): microseconds and algorithmic TFLOP/s per shape.

    python tools/wgrad_probe.py [batch=32] [ksize=3] [small]
small: the low-resolution layers that stay on the gather kernel (compare I2R_WGRAD_PLAN=old, a per-process switch).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
KS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SHAPES = [  # Cin, Cout, side
    (128, 128, 32), (256, 128, 32), (384, 128, 32),
    (128, 256, 16), (256, 256, 16), (384, 256, 16), (512, 256, 16), (128, 128, 16), (256, 128, 16),
    (256, 256, 8), (512, 256, 8), (128, 256, 8), (384, 256, 8),
]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


if KS == 1:
    SHAPES = [(128, 384, 32), (128, 128, 32), (256, 128, 32), (384, 128, 32), (256, 768, 16), (256, 256, 16),
              (128, 384, 16), (128, 128, 16), (512, 256, 16), (256, 768, 8), (256, 256, 8), (512, 256, 8),
              (512, 1536, 4), (512, 512, 4)]
if len(sys.argv) > 3 and sys.argv[3] == "small":
    SHAPES = ([(256, 256, 4), (512, 256, 4), (512, 512, 4), (512, 512, 2), (1024, 512, 2), (768, 256, 4)] if KS == 3 else
              [(512, 1536, 2), (512, 512, 2), (256, 768, 4), (256, 256, 4), (512, 1536, 4), (512, 256, 4)])
print("%-22s %10s %8s %10s %8s %7s" % ("shape", "gather us", "TF", "tile us", "TF", "x"), flush=True)
for Cin, Cout, S in SHAPES:
    x = torch.randn(B, Cin, S, S, device="cuda")
    dy = torch.randn(B, Cout, S, S, device="cuda")
    dw = torch.zeros(Cout, Cin, KS, KS, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    flop = 2.0 * B * S * S * Cin * Cout * KS * KS
    t = {}
    for sw in ("0", "1"):
        os.environ["I2R_WGRAD_TILE"] = sw
        t[sw] = timed(lambda: ops.conv2d_wgrad(x, dy, KS, dw=dw, db=db, accumulate=True))
    print("%-22s %10.1f %8.1f %10.1f %8.1f %7.2f" % (f"{Cin}->{Cout} {S}x{S}", t["0"], flop / t["0"] * 1e-6, t["1"],
                                                     flop / t["1"] * 1e-6, t["0"] / t["1"]), flush=True)
