#!/usr/bin/env python3
"""Algorithmic GFLOP of one denoiser evaluation per series (convs + attention, as bench.py's launch table counts
them): the `fwd_gflop` figures of imageni2r_amd/workloads.py.

    python tools/fwd_gflop.py [workload ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from bench import build_model, kernel_breakdown  # noqa: E402
from imageni2r_amd.workloads import WORKLOADS  # noqa: E402

imageni2r_amd.set_conv_math("f16x2")
B = 8
for wl in (sys.argv[1:] or list(WORKLOADS)):
    model, args = build_model(wl, torch.device("cuda"), seed=0)
    agg, _ = kernel_breakdown(model, args, B, torch.device("cuda"), None)
    gf = sum(a[1] for a in agg.values()) / B * 1e-9
    print(f"{wl}: fwd_gflop = {gf:.3f}  (table holds {WORKLOADS[wl]['fwd_gflop']})", flush=True)
    del model
    torch.cuda.empty_cache()
