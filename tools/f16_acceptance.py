#!/usr/bin/env python3
"""Acceptance check BASELINE.json attaches to the optional reduced-precision inference path (set_conv_math("f16")):
"match to 1e-2, with discriminative and predictive scores within the reference's reported std".

The reference reports no scores in its repository, so the band is measured here: a shape-A model
(configs/seq_len_24/sine.yaml: 57.6 M parameters) is trained on the reference's synthetic sine data
(utils/utils_data.py:78-122 restated: per feature sin(freq * t + phase), freq, phase ~ U(0, 0.1), scaled to (0, 1))
with the graph-replayed train step, then the SAME latents are sampled with the default f16x2-split kernels and with the
single-term f16 kernels, and both sets are scored R times with the on-device metrics (imageni2r_amd/metrics.py, the
reference's algorithms).  Reported: max relative deviation of the generated series, and per metric mean +- std over
the R repetitions for both paths.  Pass = series within 1e-2 and |mean_f16 - mean_f16x2| <= the default path's own std
(floored at 0.01: the scores are rounded to 4 digits and the discriminator's test split has a few hundred series).

    python tools/f16_acceptance.py [--steps 1500] [--series 1024] [--reps 3] [--out profiles/r02_f16_acceptance.json]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import DiffusionProcess, FusedAdamW, GraphedTrainStep, TS2img_Karras, metrics  # noqa: E402
from imageni2r_amd.workloads import image_side, workload_args  # noqa: E402


def sine_data(no, seq_len, dim, seed):
    rng = np.random.default_rng(seed)
    freq = rng.uniform(0, 0.1, (no, 1, dim))
    phase = rng.uniform(0, 0.1, (no, 1, dim))
    t = np.arange(seq_len)[None, :, None]
    return ((np.sin(freq * t + phase) + 1) * 0.5).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--series", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda")
    args = workload_args("A_sine24", device="cuda")
    S = image_side(args)
    torch.manual_seed(0)
    np.random.seed(0)
    model = TS2img_Karras(args, dev).to(dev).train()
    real = sine_data(4096, args.seq_len, args.input_channels, seed=1)
    real_dev = torch.from_numpy(real).to(dev)
    opt = FusedAdamW(model.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0,
                     ema=(model.model_ema, model.net))
    step = GraphedTrainStep(model, opt)
    mask = torch.ones(args.batch_size, args.input_channels, S, S, device=dev)
    t0 = time.perf_counter()
    losses = []
    for it in range(a.steps):
        idx = torch.randint(0, real_dev.shape[0], (args.batch_size,), device=dev)
        loss = step(model.ts_to_img(real_dev[idx]), mask)
        if it % 100 == 0 or it == a.steps - 1:
            losses.append(round(float(loss), 5))
            print(f"train step {it}: loss {losses[-1]}", flush=True)
    train_s = time.perf_counter() - t0
    model.eval()
    xT = torch.randn(a.series, args.input_channels, S, S, generator=torch.Generator().manual_seed(7))
    gen = {}
    with torch.no_grad(), model.ema_scope():
        for math in ("f16x2", "f16"):
            imageni2r_amd.set_conv_math(math)
            proc = DiffusionProcess(args, model.net, (args.input_channels, S, S))
            gen[math] = model.img_to_ts(proc.sample(xT)).clamp(0, 1)
        imageni2r_amd.set_conv_math("f16x2")
    dev_rel = float((gen["f16"] - gen["f16x2"]).abs().max() / gen["f16x2"].abs().max())
    print(f"generated series, f16 vs f16x2: max rel deviation {dev_rel:.3e}", flush=True)
    margs = SimpleNamespace(input_size=args.input_channels, device="cuda")
    real_eval = real_dev[: a.series]
    out = {"workload": "A_sine24", "train_steps": a.steps, "train_seconds": train_s, "losses": losses,
           "series": a.series, "reps": a.reps, "series_max_rel_deviation_f16_vs_f16x2": dev_rel, "scores": {}}
    for math in ("f16x2", "f16"):
        disc, pred = [], []
        for r in range(a.reps):  # same metric seeds for both paths: the generated set is the only difference
            np.random.seed(100 + r); torch.manual_seed(100 + r)
            disc.append(float(metrics.discriminative_score_metrics(real_eval, gen[math], margs)))
            np.random.seed(100 + r); torch.manual_seed(100 + r)
            pred.append(float(metrics.predictive_score_metrics(real_eval, gen[math], device="cuda")))
            print(f"{math} rep {r}: disc {disc[-1]:.4f} pred {pred[-1]:.4f}", flush=True)
        out["scores"][math] = {"disc": disc, "disc_mean": float(np.mean(disc)), "disc_std": float(np.std(disc)),
                               "pred": pred, "pred_mean": float(np.mean(pred)), "pred_std": float(np.std(pred))}
    s0, s1 = out["scores"]["f16x2"], out["scores"]["f16"]
    out["disc_within_std"] = abs(s1["disc_mean"] - s0["disc_mean"]) <= max(s0["disc_std"], 0.01)
    out["pred_within_std"] = abs(s1["pred_mean"] - s0["pred_mean"]) <= max(s0["pred_std"], 0.01 * s0["pred_mean"])
    out["accepted"] = bool(dev_rel < 1e-2 and out["disc_within_std"] and out["pred_within_std"])
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)
    sys.exit(0 if out["accepted"] else 1)


if __name__ == "__main__":
    main()
