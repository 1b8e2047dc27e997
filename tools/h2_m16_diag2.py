#!/usr/bin/env python3
"""Error map of conv3x3_h2_m16_kernel against the 32x32x16 kernel on random data: which samples / channels / positions
differ.  Debug aid."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
B, Cin, Cout, H = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (512, 128, 128, 16)))
gen = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, H, H, generator=gen).to(dev)
w = (torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9)).to(dev)
wp = ops.conv2d_pack_h2(w)
RES = float(os.environ.get("DIAG_RES", "0"))
res = torch.full((B, Cout, H, H), RES, device=dev) if RES else None
os.environ["I2R_H2_M16"] = "1"
y = ops.conv2d_fwd_h2(x, tuple(w.shape), wp, None, res)
os.environ["I2R_H2_M16"] = "0"
y0 = ops.conv2d_fwd_h2(x, tuple(w.shape), wp, None, res)
torch.cuda.synchronize()
bad = (y - y0).abs() > 1e-4 * y0.abs().max()
print("nan/inf in new:", int((~torch.isfinite(y)).sum()))
bs = bad.flatten(1).any(1).nonzero().flatten().tolist()
print("wrong samples:", len(bs), bs[:40])
if bs:
    b0 = bs[0]
    print("sample", b0, "wrong channels:", bad[b0].flatten(1).any(1).nonzero().flatten().tolist())
    print("wrong positions (any channel):")
    print(bad[b0].any(0).int().cpu().numpy())
    c0 = bad[b0].flatten(1).any(1).nonzero().flatten()[0]
    m = bad[b0, c0]
    print("channel", int(c0), "new (wrong elements):", y[b0, c0][m].cpu().numpy().round(2))
    print("old (same elements):", y0[b0, c0][m].cpu().numpy().round(2))
    # is it another tile's / chunk's data?  compare with the other samples' correct outputs
    d = (y[b0].unsqueeze(0) - y0).abs().flatten(1).max(1).values
    print("closest correct sample:", int(d.argmin()), float(d.min()))
