#!/usr/bin/env python3
"""time the IMPORTED REFERENCE (never copied; /root/reference must exist -- build container
only) on the host CPU cores, and the oracle (oracle/i2r_oracle.py, what bench.py's cpu_baseline times on the GPU box)
beside it on the same inputs, so the "port" baseline can be read as the reference's.

    python tools/time_reference_cpu.py [--threads 8] [--out profiles/r02_reference_cpu_timing.json]

Protocol: torch.set_num_threads(threads), 1 warm-up + 3 timed iterations, median (run it on an otherwise idle box).  Shape A (sine-24) at the config
batch 128: train step (ts_to_img x2 -> loss_fn_irregular -> backward; and the full step with clip + AdamW + EMA) and
sampling(128) + img_to_ts; B1 / D at batch 8 for scaling context.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, ROOT)
torch.Tensor.cuda = lambda self, *a, **k: self  # img_transformations.py:170 ends in .cuda(); CPU-only process

from models.our import TS2img_Karras as RefModel  # noqa: E402  (reference)
from models.sampler import DiffusionProcess as RefProcess  # noqa: E402  (reference)

from imageni2r_amd.workloads import image_side, workload_args  # noqa: E402
from oracle import i2r_oracle as O  # noqa: E402


def median_time(fn, warmup=2, iters=5):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def synthetic(args, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, args.seq_len, args.input_channels, generator=g)
    for b in range(B):
        x[b, torch.randperm(args.seq_len, generator=g)[: int(0.3 * args.seq_len)], :] = float("nan")
    return x, torch.rand(B, args.seq_len, args.input_channels, generator=g)


def run(name, B, do_sampler, iters):
    args = workload_args(name, device="cpu")
    S = image_side(args)
    torch.manual_seed(0)
    m = RefModel(args, "cpu")
    O.perturb_zero_init(m.net.state_dict(), seed=1234)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    x_ts, x_tilde = synthetic(args, B)
    res = {"workload": name, "batch": B}

    def fwd_bwd():
        m.train()
        x_img = m.ts_to_img(x_ts)
        mask = torch.isnan(x_img).float() * -1 + 1
        loss, _ = m.loss_fn_irregular(m.ts_to_img(x_tilde), mask)
        opt.zero_grad()
        loss.backward()

    def full_step():
        fwd_bwd()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        m.on_train_batch_end()

    t, all_t = median_time(fwd_bwd, 1, iters)
    res["reference_train_fwd_bwd_samples_per_s"] = B / t
    t, all_t = median_time(full_step, 1, iters)
    res["reference_train_full_step_samples_per_s"] = B / t
    res["reference_train_full_step_s"] = all_t
    if do_sampler:
        m.eval()
        proc = RefProcess(args, m.net, (args.input_channels, S, S))
        xT = torch.randn(B, args.input_channels, S, S, generator=torch.Generator().manual_seed(1))
        m.ts_to_img(x_tilde)  # caches img_shape

        def sample():
            with torch.no_grad():
                return m.img_to_ts(proc.sampling(B, xT=xT))

        t, all_t = median_time(sample, 1, iters)
        res["reference_sampler_series_per_s"] = B / t
        res["reference_sampler_s"] = all_t
        sd = {k: v.detach() for k, v in m.net.state_dict().items()}
        spec = O.unet_spec(args.img_resolution, args.input_channels, args.unet_channels, args.ch_mult,
                           args.attn_resolution)

        def port():
            with torch.no_grad():
                out = O.heun_sample(lambda xx, s: O.edm_precond(sd, spec, xx, s), xT, args.diffusion_steps)
            return O.img_to_ts(out.numpy(), args.seq_len, args.delay, args.embedding)

        t, all_t = median_time(port, 1, iters)
        res["oracle_sampler_s"] = all_t
        res["oracle_sampler_series_per_s"] = B / t
        res["oracle_over_reference"] = res["oracle_sampler_series_per_s"] / res["reference_sampler_series_per_s"]
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    out = {"torch": torch.__version__, "threads": a.threads, "cores": os.cpu_count(),
           "protocol": "median of 3 timed iterations after 1 warm-up, synthetic seeded inputs, dropout live in training",
           "runs": [run("A_sine24", 128, True, 3), run("B1_weather96", 8, True, 3), run("D_etth1_768", 8, True, 3)]}
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)
