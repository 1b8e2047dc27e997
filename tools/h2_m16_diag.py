#!/usr/bin/env python3
"""Which tap lands where: conv3x3_h2_m16_kernel on a one-hot input and weights that encode (tap, input channel), the
output printed beside the expected one for a few rows.  Debug aid for changes to the kernel's position mapping."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["I2R_H2_M16_MINTILES"] = "1"
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
B, Cin, Cout, H = 64, 32, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.zeros(B, Cin, H, H)
for (c, yy, xx) in ((0, 5 % H, 6 % H), (9, 2 % H, 3 % H)):
    x[1, c, yy, xx] = 1.0
w = torch.zeros(Cout, Cin, 3, 3)
for r in range(3):
    for s_ in range(3):
        w[0, :, r, s_] = 1 + r * 3 + s_           # channel 0: tap id 1..9
        w[1, 9, r, s_] = 10 * (1 + r * 3 + s_)    # channel 1: only input channel 9
wp = ops.conv2d_pack_h2(w.to(dev))
xd = x.to(dev)
os.environ["I2R_H2_M16"] = "1"
y = ops.conv2d_fwd_h2(xd, tuple(w.shape), wp)
torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(x, w, padding=1)
torch.set_printoptions(linewidth=200, precision=0, sci_mode=False)
for co in (0, 1):
    print("channel", co, "kernel:")
    print(y[1, co].cpu())
    print("expected:")
    print(ref[1, co])
bad = (y.cpu() - ref).abs() > 1e-3
print("wrong elements:", int(bad.sum()), "of", bad.numel(), "; samples with errors:", bad.flatten(1).any(1).nonzero().flatten().tolist()[:20])
