#!/usr/bin/env python3
"""the EDM step beside a TST-shaped completion stage, per training iteration (ms), on one GPU.

    python tools/tst_boundary_bench.py [--workload B1_weather96] [--iters 30]

Three loops over the same synthetic batch (run_irregular.py:209-244; the completion network is the builder-written
stand-in of imageni2r_amd/tst_boundary.py -- the reference's TST is not shipped or imported):
  reference-shaped : eager diffusion step (torch AdamW, clip, EMA), TST forward twice, empty_cache() every iteration
  boundary         : GraphedTrainStep + detached image + TST forward once + no empty_cache()
  boundary, 2 fwd  : same but keeping the second TST forward (bit-for-bit the reference's order of random draws)
and the stages alone (completion fwd+bwd+step, diffusion step) so the split is visible.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model, synthetic_series  # noqa: E402
from imageni2r_amd import FusedAdamW, GraphedTrainStep  # noqa: E402
from imageni2r_amd.tst_boundary import CompletionStandIn, TrainIteration, recovery_loss  # noqa: E402


def timed(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="B1_weather96")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda")
    out = {"workload": a.workload}

    def fresh():
        model, args = build_model(a.workload, dev, seed=0)
        model.train()
        torch.manual_seed(1)
        comp = CompletionStandIn(args.input_channels, args.seq_len).to(dev).train()
        return model, args, comp, torch.optim.Adam(comp.parameters())

    model, args, comp, opt_er = fresh()
    x_ts, _ = synthetic_series(args, args.batch_size, dev, seed=100)
    out["batch"] = args.batch_size
    pm = ~torch.isnan(x_ts).any(dim=-1)

    def completion_only():
        loss = recovery_loss(comp(x_ts, pm), x_ts)
        opt_er.zero_grad()
        loss.backward()
        opt_er.step()

    out["completion_fwd_bwd_step_ms"] = timed(completion_only, a.iters)
    with torch.no_grad():
        out["completion_fwd_ms"] = timed(lambda: comp(x_ts, pm), a.iters)

    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    ref_loop = TrainIteration(model, comp, opt_er, diffusion_optimizer=opt, reuse_completion=False, empty_cache=True)
    out["reference_shaped_iteration_ms"] = timed(lambda: ref_loop(x_ts), a.iters)
    del model, comp, opt, opt_er, ref_loop
    torch.cuda.empty_cache()

    for key, reuse in (("boundary_iteration_ms", True), ("boundary_two_forwards_iteration_ms", False)):
        model, args, comp, opt_er = fresh()
        fopt = FusedAdamW(model.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0,
                          ema=(model.model_ema, model.net))
        gstep = GraphedTrainStep(model, fopt)
        loop = TrainIteration(model, comp, opt_er, diffusion_step=gstep, reuse_completion=reuse)
        out[key] = timed(lambda: loop(x_ts), a.iters)
        if reuse:
            x_img, mask = model.ts_to_img_with_mask(x_ts)
            xt = model.ts_to_img(torch.rand_like(torch.nan_to_num(x_ts)))
            out["diffusion_step_graph_ms"] = timed(lambda: gstep(xt, mask), a.iters)
        del model, comp, opt_er, fopt, gstep, loop
        torch.cuda.empty_cache()
    out["speedup_vs_reference_shaped"] = out["reference_shaped_iteration_ms"] / out["boundary_iteration_ms"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
