#!/usr/bin/env python3
"""List every conv / GroupNorm / attention call of one denoiser evaluation with its shapes and HBM-byte floor."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import imageni2r_amd  # noqa: E402
from imageni2r_amd import networks  # noqa: E402
from bench import build_model  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "B1_weather96"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
imageni2r_amd.set_conv_math("f16x2")
model, args = build_model(wl, torch.device("cuda"), seed=0)
net = model.net.model if hasattr(model.net, "model") else model.net
rows = []


def hook(name):
    def f(mod, inp, out):
        x = inp[0]
        if isinstance(x, tuple):
            cin = sum(t.shape[1] for t in x)
            x = x[0]
        else:
            cin = x.shape[1]
        k = mod.weight.shape[-1] if mod.weight is not None and mod.weight.dim() == 4 else 0
        rows.append((name, cin, out.shape[1], out.shape[2], k, len(inp) > 1 and inp[1] is not None))
    return f


for n, m in net.named_modules():
    if isinstance(m, networks.Conv2d):
        m.register_forward_hook(hook(n))
S = args.img_resolution if hasattr(args, "img_resolution") else 16
from imageni2r_amd.workloads import image_side  # noqa: E402
S = image_side(args)
x = torch.randn(B, args.input_channels, S, S, device="cuda")
sig = torch.ones(B, device="cuda")
with torch.no_grad():
    model.net(x, sig, None)
tot = {1: 0.0, 3: 0.0}
for (n, ci, co, h, k, res) in rows:
    byts = (ci + co * (2 if res else 1)) * h * h * 4
    tot[k] = tot.get(k, 0) + byts
    print(f"{n:40s} Cin{ci:5d} Cout{co:5d} H{h:3d} k{k} res{int(res)}  {byts/1e3:8.1f} KB/sample  {2*ci*co*k*k*h*h/1e6:8.1f} MFLOP/sample")
print({k: v / 1e6 for k, v in tot.items()}, "MB per sample per eval")
