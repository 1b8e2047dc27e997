#!/usr/bin/env python3
"""Discriminative score on the GPU: the eager 2000-step loop against the captured-graph loop (same seeds)."""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import metrics as M  # noqa: E402
from tests.test_metrics_cpu import metric_inputs  # noqa: E402

for n, L, C in ((160, 24, 6), (2000, 24, 6), (1000, 96, 12)):
    real, gen = metric_inputs(seed=0, n=n, L=L, C=C)
    a, b = torch.from_numpy(real).cuda(), torch.from_numpy(gen).cuda()
    args = SimpleNamespace(input_size=C, device="cuda")
    for mode in ("0", "1", "1"):
        os.environ["I2R_METRICS_GRAPH"] = mode
        vals = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for seed in (0, 1, 2):
            np.random.seed(seed); torch.manual_seed(seed)
            vals.append(float(M.discriminative_score_metrics(a, b, args)))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"n={n} L={L} C={C} graph={mode}: {dt:6.2f} s per score; scores {['%.4f' % v for v in vals]}", flush=True)
