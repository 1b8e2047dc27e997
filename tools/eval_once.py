#!/usr/bin/env python3
"""One eager denoiser evaluation of workload B1 at the given batch (default 1024): the dispatch list the rocprofv3
--pmc passes for the roofline's `traffic` figure are taken over (tools/traffic_summary.py reads their CSVs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd.workloads import image_side  # noqa: E402
from bench import build_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
imageni2r_amd.set_conv_math("f16x2")
model, args = build_model("B1_weather96", torch.device("cuda"), seed=0)
S = image_side(args)
x = torch.randn(B, args.input_channels, S, S, device="cuda")
sig = torch.tensor([1.5], device="cuda")
with torch.no_grad():
    model.net(x, sig)
torch.cuda.synchronize()
print("done")
