#!/usr/bin/env python3
"""Sweep tile variant x split-K of the fp32 conv kernel over the layer shapes of a workload at a given batch
(tuning aid for conv_plan in csrc/conv_fwd.hip).  Prints, per shape, the default plan's time and the best override."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
B = a.batch
# (Cin, Cout, H, k) of the B1/C network (fwd); dgrad = swapped channels
SH = [(128, 128, 16, 3), (256, 128, 16, 3), (384, 128, 16, 3), (128, 256, 8, 3), (256, 256, 8, 3), (512, 256, 8, 3),
      (384, 256, 8, 3), (256, 256, 4, 3), (512, 256, 4, 3), (768, 256, 4, 3), (256, 512, 2, 3), (512, 512, 2, 3),
      (1024, 512, 2, 3), (768, 512, 2, 3), (128, 384, 16, 1), (128, 128, 16, 1), (256, 768, 8, 1), (256, 256, 8, 1),
      (256, 768, 4, 1), (512, 1536, 2, 1), (512, 512, 2, 1), (256, 128, 16, 1), (512, 256, 8, 1), (1024, 512, 2, 1)]
SH = SH + [(co, ci, h, k) for (ci, co, h, k) in SH if ci != co]
dev = "cuda"


def timeit(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


tot_def = tot_best = 0.0
for (Cin, Cout, H, k) in SH:
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    wp = ops.conv2d_pack(w)
    bias = torch.randn(Cout, device=dev)
    os.environ.pop("I2R_CONV_VARIANT", None)
    os.environ.pop("I2R_CONV_NSPLIT", None)
    base = timeit(lambda: ops.conv2d_fwd(x, w, bias, None, wp))
    best = (base, "default")
    res = []
    for v in (0, 1, 4, 5):
        for ns in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            os.environ["I2R_CONV_VARIANT"] = str(v)
            os.environ["I2R_CONV_NSPLIT"] = str(ns)
            # workspace sized for the override
            t = timeit(lambda: ops.conv2d_fwd(x, w, bias, None, wp))
            res.append((t, v, ns))
            if t < best[0]:
                best = (t, f"v{v} ns{ns}")
    res.sort()
    fl = 2.0 * B * H * H * Cout * Cin * k * k
    tot_def += base
    tot_best += best[0]
    print(f"Cin{Cin:5d} Cout{Cout:5d} H{H:3d} k{k}: default {base:7.1f}us ({fl/base*1e-6:6.1f} TF)  best {best[0]:7.1f}us "
          f"{best[1]:10s} ({fl/best[0]*1e-6:6.1f} TF)  top3 {[(round(t,1),v,ns) for t,v,ns in res[:3]]}", flush=True)
print(f"sum default {tot_def:.0f} us, sum best {tot_best:.0f} us")
