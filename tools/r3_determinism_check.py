#!/usr/bin/env python3
"""Is the sampler bit-reproducible at the bench batch?  Samples the same latents several times (same process, replayed
graphs) and in a second DiffusionProcess, and reports the largest difference between any two results.
usage: r3_determinism_check.py [workload [batch [repeats]]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import DiffusionProcess  # noqa: E402
from imageni2r_amd.workloads import image_side  # noqa: E402
from bench import build_model  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "B1_weather96"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
imageni2r_amd.set_conv_math("f16x2")
model, args = build_model(wl, dev, seed=0)
S = image_side(args)
gen = torch.Generator(device=dev).manual_seed(5)
xT = torch.randn(batch, args.input_channels, S, S, device=dev, generator=gen)
proc = DiffusionProcess(args, model.net, (args.input_channels, S, S))
noise = os.environ.get("DET_NOISE") == "1"   # an unrelated stream kept busy beside the sampler (matmuls + copies)
ns = torch.cuda.Stream(device=dev)
na, nb = torch.randn(4096, 4096, device=dev), torch.randn(1 << 26, device=dev)


def sample_once():
    if noise:
        ns.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ns):
            for _ in range(1500):
                torch.mm(na, na)
                nb.mul_(1.0001)
    o = proc.sample(xT).clone()
    torch.cuda.synchronize()
    return o


outs = [sample_once() for _ in range(reps)]
proc2 = DiffusionProcess(args, model.net, (args.input_channels, S, S))
outs.append(proc2.sample(xT).clone())
torch.cuda.synchronize()
ref = outs[0]
scale = float(ref.abs().max())
for i, o in enumerate(outs[1:], 1):
    d = (o - ref).abs()
    nbad = int((d > 0).flatten(1).any(1).sum())
    print(f"run {i} vs run 0: equal bits {torch.equal(o, ref)}, max diff / max {float(d.max()) / scale:.3e}, series that differ {nbad} of {batch}",
          flush=True)
    if nbad:
        bad = (d > 0).flatten(1).any(1).nonzero().flatten().tolist()
        half = batch // 2
        runs, start, prev = [], bad[0], bad[0]
        for b_ in bad[1:]:
            if b_ != prev + 1:
                runs.append((start, prev - start + 1))
                start = b_
            prev = b_
        runs.append((start, prev - start + 1))
        print(f"   first half {sum(1 for b_ in bad if b_ < half)}, second half {sum(1 for b_ in bad if b_ >= half)}; "
              f"runs of consecutive series (start, length): {runs[:16]}{' ...' if len(runs) > 16 else ''}", flush=True)
