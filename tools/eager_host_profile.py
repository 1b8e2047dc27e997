#!/usr/bin/env python3
"""Where the HOST time of the reference's own (eager) training loop goes on this path: cProfile over a few steps of
loss_fn_irregular -> backward -> clip -> AdamW -> EMA at the config batch, top functions by own time.

    python tools/eager_host_profile.py [workload=B1_weather96] [steps=10]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, synthetic_series  # noqa: E402
from imageni2r_amd.workloads import WORKLOADS  # noqa: E402

WL = sys.argv[1] if len(sys.argv) > 1 else "B1_weather96"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
model, args = build_model(WL, dev, seed=0)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
B = WORKLOADS[WL]["batch_size"]
x_ts, x_tilde = synthetic_series(args, B, dev, seed=1)


def step():
    x_img = model.ts_to_img(x_ts)
    mask = torch.isnan(x_img).float() * -1 + 1
    xt = model.ts_to_img(x_tilde)
    loss, _ = model.loss_fn_irregular(xt, mask)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    model.on_train_batch_end()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
print(f"{WL}: {(time.perf_counter() - t0) / STEPS * 1e3:.2f} ms per eager step (batch {B})", flush=True)
# host enqueue time per phase (no synchronisation inside a step: what the Python thread spends before it can move on)
ph = {"embed": 0.0, "forward": 0.0, "backward": 0.0, "clip": 0.0, "adamw": 0.0, "ema": 0.0}
for _ in range(STEPS):
    t = [time.perf_counter()]
    x_img = model.ts_to_img(x_ts); mask = torch.isnan(x_img).float() * -1 + 1; xt = model.ts_to_img(x_tilde)
    t.append(time.perf_counter())
    loss, _ = model.loss_fn_irregular(xt, mask)
    t.append(time.perf_counter())
    opt.zero_grad(); loss.backward()
    t.append(time.perf_counter())
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    t.append(time.perf_counter())
    opt.step()
    t.append(time.perf_counter())
    model.on_train_batch_end()
    t.append(time.perf_counter())
    for k, a, b in zip(ph, t[:-1], t[1:]):
        ph[k] += (b - a) / STEPS * 1e3
    torch.cuda.synchronize()
print("host ms per phase:", {k: round(v, 2) for k, v in ph.items()}, "sum", round(sum(ph.values()), 2), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
