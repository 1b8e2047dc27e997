#!/usr/bin/env python3
"""Micro-benchmark of i2r_conv2d_fwd_f32 / wgrad / dgrad on chosen shapes (for rocprofv3 --pmc passes and A/B tuning).

    python tools/conv_bench.py --shape 256,128,128,16,3 --iters 20 [--op fwd|dgrad|wgrad]
shape = B,Cin,Cout,H,k
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

SHAPES_B1_1024 = [f"1024,{ci},{co},{h},{k}" for (ci, co, h, k) in
                  [(128, 128, 16, 3), (256, 128, 16, 3), (384, 128, 16, 3), (128, 256, 8, 3), (256, 256, 8, 3),
                   (512, 256, 8, 3), (256, 256, 4, 3), (512, 256, 4, 3), (768, 256, 4, 3), (512, 512, 2, 3),
                   (1024, 512, 2, 3), (128, 384, 16, 1), (128, 128, 16, 1), (256, 768, 8, 1), (256, 256, 8, 1),
                   (256, 768, 4, 1), (512, 1536, 2, 1)]]
SHAPES_B1_256 = ["256,128,128,16,3", "256,256,256,8,3", "256,512,256,8,3", "256,256,256,4,3", "256,512,256,4,3",
                 "256,512,512,2,3", "256,1024,512,2,3", "256,256,768,8,1", "256,128,384,16,1"]

ap = argparse.ArgumentParser()
ap.add_argument("--shape", action="append")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--op", default="fwd")
ap.add_argument("--zeros", action="store_true", help="all-zero operands (DVFS probe)")
ap.add_argument("--set", default="", help="named shape set: b1_1024")
ap.add_argument("--split-in", action="store_true", help="h2: feed pre-split activations")
ap.add_argument("--qkv-out", action="store_true", help="h2: store split pairs with q pre-scaled (the qkv conv's mode)")
ap.add_argument("--res", action="store_true", help="h2: add a residual in the epilogue")
a = ap.parse_args()
shapes = a.shape or (SHAPES_B1_1024 if a.set == "b1_1024" else SHAPES_B1_256)
dev = "cuda"
for sh in shapes:
    B, Cin, Cout, H, k = map(int, sh.split(","))
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, device=dev)
    dy = torch.randn(B, Cout, H, H, device=dev)
    if a.zeros:
        x.zero_(); w.zero_(); dy.zero_()
    if a.op == "fwd":
        wp = ops.conv2d_pack(w)
        fn = lambda: ops.conv2d_fwd(x, w, b, None, wp)
    elif a.op == "h2":
        wph = ops.conv2d_pack_h2(w)
        res = torch.randn(B, Cout, H, H, device=dev) if a.res else None
        fn = lambda: ops.conv2d_fwd_h2(x, w.shape, wph, b, res, a.split_in, None, a.qkv_out)
    elif a.op == "dgrad":
        fn = lambda: ops.conv2d_dgrad(dy, w)
    else:
        fn = lambda: ops.conv2d_wgrad(x, dy, k)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = 2.0 * B * H * H * Cout * Cin * k * k
    nbytes = 4.0 * B * H * H * (Cin + Cout * (2 if (a.op == "h2" and a.res) else 1))
    print(f"{a.op} B{B} Cin{Cin} Cout{Cout} H{H} k{k}: {ms*1e3:8.1f} us  {fl/ms*1e-9:7.1f} TFLOP/s  "
          f"{nbytes/ms*1e-6:7.0f} GB/s", flush=True)
