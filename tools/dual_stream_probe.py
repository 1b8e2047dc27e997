#!/usr/bin/env python3
"""Two half-batch sampler graphs replayed concurrently on two streams vs one full-batch graph: does overlapping the
HBM-bound kernels of one half with the MFMA-bound kernels of the other raise series/s?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import DiffusionProcess  # noqa: E402
from imageni2r_amd.workloads import image_side  # noqa: E402
from bench import build_model  # noqa: E402

dev = torch.device("cuda")
imageni2r_amd.set_conv_math("f16x2")
model, args = build_model("B1_weather96", dev, seed=0)
S = image_side(args)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def timeit(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


full = DiffusionProcess(args, model.net, (args.input_channels, S, S))
x_full = torch.randn(N, args.input_channels, S, S, device=dev)
t_full = timeit(lambda: full.sample(x_full))
print(f"one graph, {N} series: {N / t_full:.1f} series/s", flush=True)

halves = [DiffusionProcess(args, model.net, (args.input_channels, S, S)) for _ in range(NS)]
xs = [torch.randn(N // NS, args.input_channels, S, S, device=dev) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
for h, x in zip(halves, xs):  # capture each on the default stream first
    h.sample(x)
torch.cuda.synchronize()


def dual():
    cur = torch.cuda.current_stream()
    for st, h, x in zip(streams, halves, xs):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            h.sample(x)
    for st in streams:
        cur.wait_stream(st)


t_dual = timeit(dual)
print(f"{NS} graphs x {N // NS} on {NS} streams: {N / t_dual:.1f} series/s", flush=True)
