#!/usr/bin/env python3
"""Which Python lines of the train step issue the torch-side device work that is NOT one of the library's kernels:
device-to-device copies (aten::copy_ / clone), fills (zero_ / zeros), adds / muls by autograd.  Runs the step of
GraphedTrainStep eagerly under torch.profiler with stacks and groups those ops by the innermost frame of this package.

    python tools/step_op_sources.py [workload=B1_weather96]
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, synthetic_series  # noqa: E402
from imageni2r_amd import FusedAdamW, GraphedTrainStep  # noqa: E402
from imageni2r_amd.workloads import WORKLOADS  # noqa: E402

WL = sys.argv[1] if len(sys.argv) > 1 else "B1_weather96"
dev = torch.device("cuda")
model, args = build_model(WL, dev, seed=0)
model.train()
opt = FusedAdamW(model.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(model.model_ema, model.net))
gstep = GraphedTrainStep(model, opt, use_graph=False)
B = WORKLOADS[WL]["batch_size"]
x_ts, x_tilde = synthetic_series(args, B, dev, seed=1)
x_img, mask = model.ts_to_img_with_mask(x_ts)
xt = model.ts_to_img(x_tilde)
for _ in range(3):
    gstep(xt, mask)
torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::clone", "aten::zero_", "aten::zeros", "aten::fill_", "aten::add", "aten::add_",
         "aten::mul", "aten::zeros_like", "aten::contiguous", "aten::sum", "aten::cat", "aten::empty_like")
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            with_stack=True) as prof:
    gstep(xt, mask)
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    site = "(no package frame: autograd engine / torch internals)"
    for fr in ev.stack or []:
        if "imageni2r_amd" in fr or "bench.py" in fr:
            site = fr.strip()
            break
    by[(ev.name, site)] += 1
for (name, site), n in by.most_common(40):
    print("%5d  %-18s %s" % (n, name, site[-110:]), flush=True)
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kern[ev.name[:70]] += 1
print("--- device activities")
for k, n in kern.most_common(14):
    print("%5d  %s" % (n, k), flush=True)
