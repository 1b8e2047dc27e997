#!/usr/bin/env python3
"""Diagnostic for the deterministic mode: which tensors differ (bitwise) between two eager steps from the same state,
and between an eager step and a graph-replayed one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import FusedAdamW, GraphedTrainStep, TS2img_Karras, ops  # noqa: E402
from oracle import i2r_oracle as O  # noqa: E402  (diagnostic tool, not the product)
from tests.helpers import tiny_args  # noqa: E402

DEV = "cuda"
ops.set_deterministic(True)


def model():
    torch.manual_seed(0)
    m = TS2img_Karras(tiny_args(device=DEV), DEV)
    O.perturb_zero_init(m.net.state_dict(), seed=1234)
    return m.to(DEV).train()


def batch(m, seed=11):
    gen = torch.Generator().manual_seed(seed)
    ts = torch.rand(4, 24, 3, generator=gen)
    ts[0, [2, 3], :] = float("nan")
    _, mask = m.ts_to_img_with_mask(ts.to(DEV))
    return m.ts_to_img(torch.rand(4, 24, 3, generator=gen).to(DEV)), mask


def run(use_graph, steps=1):
    m = model()
    opt = FusedAdamW(m.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
    step = GraphedTrainStep(m, opt, use_graph=use_graph)
    x, k = batch(m)
    losses = [step(x, k).item() for _ in range(steps)]
    grads = {n: p.grad.detach().clone() for n, p in m.net.named_parameters()}
    weights = {n: p.detach().clone() for n, p in m.net.named_parameters()}
    return losses, grads, weights, opt.scalars.clone()


def diff(a, b, what):
    bad = [(n, int((a[n] != b[n]).sum()), a[n].numel()) for n in a if not torch.equal(a[n], b[n])]
    print(f"{what}: {len(bad)} of {len(a)} tensors differ")
    for n, c, t in bad[:40]:
        print(f"   {n}: {c}/{t} elements, max abs diff {(a[n] - b[n]).abs().max().item():.3e}")


e1, e2, g1 = run(False), run(False), run(True)
print("losses", e1[0], e2[0], g1[0])
print("scalars eager", e1[3].tolist(), "\nscalars graph", g1[3].tolist())
diff(e1[1], e2[1], "eager vs eager gradients")
diff(e1[1], g1[1], "eager vs graph gradients")
diff(e1[2], g1[2], "eager vs graph weights after 1 step")
