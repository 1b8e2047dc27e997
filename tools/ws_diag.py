#!/usr/bin/env python3
"""Error map of conv1x1_h2_ws_kernel (I2R_H2_WS=1) against the other 1x1 kernels on random data with a residual."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
B, Cin, Cout, H = 1024, 256, 256, 8
gen = torch.Generator().manual_seed(1)
MODE = sys.argv[1] if len(sys.argv) > 1 else "plain"
x = torch.randn(B, Cin, H, H, generator=gen).to(dev)
SP = "split" in MODE
if SP:
    x = ops.groupnorm_fwd(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-5, act=False, split_out=True)[0]
w = (torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)).to(dev)
res = torch.full((B, Cout, H, H), 1000.0, device=dev)
if "rres" in MODE:
    res = res + torch.randn(B, Cout, H, H, generator=gen).to(dev)
bias = torch.randn(Cout, generator=gen).to(dev) if "bias" in MODE else None
wp = ops.conv2d_pack_h2(w)
for rep in range(3):
    os.environ["I2R_H2_WS"] = "1"
    y = ops.conv2d_fwd_h2(x, tuple(w.shape), wp, bias, res, x_split=SP)
    os.environ["I2R_H2_WS"] = "0"
    y0 = ops.conv2d_fwd_h2(x, tuple(w.shape), wp, bias, res, x_split=SP)
    torch.cuda.synchronize()
    bad = (y - y0).abs() > 1e-2
    bs = bad.flatten(1).any(1).nonzero().flatten().tolist()
    print("rep", rep, "wrong samples:", len(bs), bs[:16], "wrong elements", int(bad.sum()))
    if bs:
        b0 = bs[0]
        chans = bad[b0].flatten(1).any(1).nonzero().flatten().tolist()
        print("  sample", b0, "wrong channels:", chans[:40])
        print("  wrong positions of its first wrong channel:", bad[b0, chans[0]].flatten().nonzero().flatten().tolist())
        m = bad[b0, chans[0]]
        print("  new:", y[b0, chans[0]][m].cpu().numpy().round(2)[:8], "old:", y0[b0, chans[0]][m].cpu().numpy().round(2)[:8])
