#!/usr/bin/env python3
"""Wall time of the evaluation stage's scores with the series resident on the GPU, beside the same
code on the host cores: discriminative (2000 GRU steps), predictive, Context-FID (TS2Vec fit), correlation.

    python tools/metrics_bench.py [n_series=2000] [seq_len=24] [channels=6] [out.json]
"""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import context_fid as CF  # noqa: E402
from imageni2r_amd import metrics as M  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 24
C = int(sys.argv[3]) if len(sys.argv) > 3 else 6
OUT = sys.argv[4] if len(sys.argv) > 4 else None
rng = np.random.default_rng(0)
t = np.arange(L)[None, :, None]


def sines(noise):
    f = rng.uniform(0.05, 0.3, (N, 1, C)); ph = rng.uniform(0, 2 * np.pi, (N, 1, C))
    return (0.5 + 0.4 * np.sin(2 * np.pi * f * t + ph) + noise * rng.standard_normal((N, L, C))).astype(np.float32)


real, gen = sines(0.02), sines(0.05)
rec = {"n_series": N, "seq_len": L, "channels": C, "host_threads": torch.get_num_threads(), "scores": {}, "seconds": {}}
for dev in ("cuda", "cpu"):
    a, b = (torch.from_numpy(real).to(dev), torch.from_numpy(gen).to(dev))
    args = SimpleNamespace(input_size=C, device=dev)
    jobs = {"discriminative": lambda: M.discriminative_score_metrics(a, b, args),
            "predictive": lambda: M.predictive_score_metrics(a, b, device=dev),
            "context_fid": lambda: CF.context_fid(a, b, device=dev),
            "correlation": lambda: M.calculate_pearson_correlation(a, b, device=dev)[0]}
    for name, fn in jobs.items():
        np.random.seed(0); torch.manual_seed(0)
        if dev == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = float(fn())
        if dev == "cuda":
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec["scores"][f"{name}_{dev}"] = v
        rec["seconds"][f"{name}_{dev}"] = round(dt, 3)
        print(f"{dev:5s} {name:15s} {dt:8.2f} s  score {v:.4f}", flush=True)
if OUT:
    with open(OUT, "w") as f:
        json.dump(rec, f, indent=1)
