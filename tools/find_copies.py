#!/usr/bin/env python3
"""Which host call sites launch ATen kernels (copies, fills, adds ...) inside one eager train step / one eager denoiser
evaluation: aten ops with device time, grouped by the innermost repo frame."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, synthetic_series  # noqa: E402
from imageni2r_amd import FusedAdamW, GraphedTrainStep  # noqa: E402
from imageni2r_amd.workloads import image_side  # noqa: E402

dev = torch.device("cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
model, args = build_model("B1_weather96", dev, seed=0)
S = image_side(args)
if mode == "train":
    model.train()
    opt = FusedAdamW(model.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(model.model_ema, model.net))
    step = GraphedTrainStep(model, opt, use_graph=False)
    x_ts, x_tilde = synthetic_series(args, 32, dev, seed=1)
    _, mask = model.ts_to_img_with_mask(x_ts)
    xt = model.ts_to_img(x_tilde)
    fn = lambda: step(xt, mask)
else:
    x = torch.randn(4096, args.input_channels, S, S, device=dev)
    sig = torch.tensor([1.5], device=dev)
    def fn():
        with torch.no_grad():
            model.net(x, sig)
for _ in range(3):
    fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fn()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    t_dev = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if not ev.name.startswith("aten::") or t_dev <= 0:
        continue
    frame = next((f for f in ev.stack if "/repo/" in f or "imageni2r_amd" in f), ev.stack[0] if ev.stack else "?")
    a = agg[(ev.name, frame.strip()[-110:])]
    a[0] += 1
    a[1] += t_dev
cp = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::zero_", "aten::fill_"):
        frames = [f.strip()[-100:] for f in ev.stack if "/repo/" in f or "imageni2r_amd" in f][:2]
        cp[(ev.name, " <- ".join(frames) if frames else (ev.stack[0].strip()[-100:] if ev.stack else "?"))] += 1
print("copy_/zero_/fill_ call sites:")
for (name, frame), n in cp.most_common(25):
    print(f"   n={n:4d}  {name:14s} {frame}")
print("aten ops with device time:", sum(v[0] for v in agg.values()))
for (name, frame), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t:9.1f} us  n={n:4d}  {name:28s} {frame}")
