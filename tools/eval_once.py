#!/usr/bin/env python3
"""One eager, single-stream denoiser evaluation of a workload at the given batch: the dispatch list the rocprofv3
passes behind bench.py's per-kernel figures are taken over (--kernel-trace --stats for durations; --pmc FETCH_SIZE and
--pmc WRITE_SIZE, each in a run of its own, for roofline.traffic -- tools/traffic_summary.py reads their CSVs).

    eval_once.py [batch=4096] [workload=B1_weather96] [evals=1]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd.workloads import image_side  # noqa: E402
from bench import build_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
WL = sys.argv[2] if len(sys.argv) > 2 else "B1_weather96"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1
imageni2r_amd.set_conv_math("f16x2")
model, args = build_model(WL, torch.device("cuda"), seed=0)
S = image_side(args)
x = torch.randn(B, args.input_channels, S, S, device="cuda")
sig = torch.tensor([1.5], device="cuda")
with torch.no_grad():
    for _ in range(N):
        model.net(x, sig)
torch.cuda.synchronize()
print("done", WL, B, N)
