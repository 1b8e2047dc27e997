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
own = os.environ.get("DET_OWN", "")          # ... or with ONE family of this library's kernels: gn | conv3 | conv1 | attn
from imageni2r_amd import ops  # noqa: E402
if own:
    gq = torch.Generator(device=dev).manual_seed(9)
    ox = torch.randn(2048, 128, 16, 16, device=dev, generator=gq)
    og, ob = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    oxs, _, _ = ops.groupnorm_fwd(ox, og, ob, 32, 1e-5, act=True, split_out=True)
    ow3 = ops.conv2d_pack_h2(torch.randn(128, 128, 3, 3, device=dev, generator=gq) * 0.03)
    ow1 = ops.conv2d_pack_h2(torch.randn(384, 128, 1, 1, device=dev, generator=gq) * 0.09)
    oqkv = torch.randn(2048, 384, 16, 16, device=dev, generator=gq)
ns = torch.cuda.Stream(device=dev)
na, nb = torch.randn(4096, 4096, device=dev), torch.randn(1 << 26, device=dev)


eager2 = os.environ.get("DET_EAGER2") == "1"   # the two halves EAGERLY (no graphs) on two streams
if eager2:
    pe = DiffusionProcess(args, model.net, (args.input_channels, S, S), use_graph=False)
    pe.range_check = False
    s2 = [torch.cuda.Stream(device=dev) for _ in range(2)]


def sample_once():
    if eager2:
        halves = xT.chunk(2)
        res = []
        cur = torch.cuda.current_stream()
        for st, h in zip(s2, halves):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                res.append(pe.sample(h))
        for st in s2:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        return torch.cat(res).clone()
    if own:
        ns.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ns):
            for _ in range({"gn": 9000, "conv3": 1600, "conv1": 5000, "attn": 4000}[own]):
                if own == "gn":
                    ops.groupnorm_fwd(ox, og, ob, 32, 1e-5, act=True, split_out=True)
                elif own == "conv3":
                    ops.conv2d_fwd_h2(oxs, (128, 128, 3, 3), ow3, None, ox, x_split=True)
                elif own == "conv1":
                    ops.conv2d_fwd_h2(oxs, (384, 128, 1, 1), ow1, None, None, x_split=True, qkv_out=True)
                else:
                    ops.attention_fwd(oqkv, 2)
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
