"""DiffusionProcess: the 18-step Heun (EDM) sampler of models/sampler.py:4-71 with the whole loop -- 35 denoiser
evaluations plus the fp64 state updates -- captured once into a HIP graph and replayed.

Same constructor and ``sampling`` / ``sample`` signatures as the reference.  Differences that do not change results:
the per-step tensor->bool host syncs (:50) are gone (the schedule lives on the host), and the dead
``randn_like * 0`` draw (:52, S_churn = 0) is skipped -- it only advances the device RNG stream.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops


def karras_schedule(num_steps, sigma_min, sigma_max, rho):
    """fp64 t_steps with t_N = 0 appended (:39-42), evaluated on the host with torch's own fp64 ops."""
    step = torch.arange(num_steps, dtype=torch.float64)
    t = (sigma_max ** (1 / rho) + step / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros_like(t[:1])])


class DiffusionProcess:
    def __init__(self, args, diffusion_fn, shape, use_graph=True):
        self.args = args
        self.device = args.device
        self.shape = tuple(shape)
        # DDPM leftovers of the reference ctor (:16-20); unused by `sample`, kept as host attributes
        self.betas = torch.linspace(start=args.beta1, end=args.betaT, steps=args.diffusion_steps)
        self.alphas = 1 - self.betas
        self.alpha_bars = torch.cumprod(self.alphas, dim=0)
        self.alpha_prev_bars = torch.cat([torch.ones(1), self.alpha_bars[:-1]])
        self.deterministic = args.deterministic
        self.net = diffusion_fn.to(device=self.device)
        self.sigma_data = 0.5
        self.sigma_min = 0.002
        self.sigma_max = 80
        self.rho = 7
        self.S_churn = 0
        self.S_min = 0
        self.S_max = float('inf')
        self.S_noise = 1
        self.num_steps = args.diffusion_steps
        self.use_graph = use_graph
        # captured loops live on the denoiser, not on this object: run_irregular.py:253 builds a new DiffusionProcess
        # at every evaluation, and with the pointer-swapped ema_scope the weights it sees sit at the same addresses
        # each time, so the second evaluation replays the first one's graphs
        self._graphs = self.net.__dict__.setdefault("_i2r_sampler_graphs", {})
        # batches at least this big (or of as many positions as that many 16x16 images) are sampled as two concurrent
        # half-batch graphs on two streams (0: off).  Replays are bit-reproducible in both forms
        # (tests/test_gpu_bench_batch.py; tools/r3_determinism_check.py -- which is how a store-data hazard in the small
        # attention kernel was found: it only showed beside another stream's kernels).
        self.dual_stream_min = int(os.environ.get("I2R_DUAL_STREAM_MIN", 2048))
        self._streams = None
        # fp16 range guard of the f16x2 kernels (DESIGN.md "range contract"): series that come back non-finite are
        # sampled again on the exact fp32 kernels.  One isfinite reduction + host sync per sample() call; 0 = off.
        self.range_check = os.environ.get("I2R_RANGE_CHECK", "1") != "0"
        # series per captured loop: bigger requests are sampled in pieces (every activation of a piece has to stay
        # under the conv kernels' 2 GiB buffer-addressing limit).  None = 4096 at 16x16 images, scaled with the area.
        self.max_graph_batch = int(os.environ["I2R_MAX_GRAPH_BATCH"]) if "I2R_MAX_GRAPH_BATCH" in os.environ else None

    # ------------------------------------------------------------------------------------------------------
    def _t_steps(self):
        sigma_min = max(self.sigma_min, self.net.sigma_min)
        sigma_max = min(self.sigma_max, self.net.sigma_max)
        return karras_schedule(self.num_steps, sigma_min, sigma_max, self.rho)

    def _loop(self, latents, t_host, sig_dev, steps=None):
        """The Heun loop (:45-63) on device buffers; t_host: python floats, sig_dev: fp32 device copy of t_steps.
        `steps`: only the first so many steps (the warm-up before a capture)."""
        n = self.num_steps if steps is None else min(steps, self.num_steps)
        x_cur = ops.heun_init(latents, t_host[0])
        x_next = torch.empty_like(x_cur)
        d_cur = torch.empty_like(x_cur)
        x_out = torch.empty_like(x_cur)
        for i in range(n):
            t_cur, t_next = t_host[i], t_host[i + 1]
            D = self.net(x_cur, sig_dev[i:i + 1])
            ops.heun_euler(x_cur, D, d_cur, x_next, t_cur, t_next)
            if i < n - 1:
                Dp = self.net(x_next, sig_dev[i + 1:i + 2])
                ops.heun_correct(x_cur, x_next, Dp, d_cur, x_out, t_cur, t_next)
                x_cur, x_out = x_out, x_cur
            else:
                x_cur, x_next = x_next, x_cur
        return x_cur

    def _graph_entry(self, latents, t_steps, t_host, slot=0):
        """(graph, static_in, static_out, sigma table) for this batch shape, captured on first use."""
        key = self._graph_key(latents) + (slot, tuple(t_host))
        entry = self._graphs.get(key)
        if entry is None:
            sig_dev = t_steps.to(device=latents.device, dtype=torch.float32)
            static_in = latents.clone()
            # two eager Heun steps (three denoiser evaluations: both of a full step's, and a last step's) on a side
            # stream bring up everything that is set up lazily -- packed weights, affine plans, kernel attributes --
            # before the capture; the whole 18-step loop is not needed for that
            side = torch.cuda.Stream(device=latents.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._loop(static_in, t_host, sig_dev, steps=2)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                ops.new_pack_epoch()  # weights are re-packed INSIDE the graph: replays always see live weights
                static_out = self._loop(static_in, t_host, sig_dev)
            ops.new_pack_epoch()      # and eager code never reuses the graph-owned packed buffers
            entry = (graph, static_in, static_out, sig_dev)
            self._graphs[key] = entry
        return entry

    def _graph_key(self, latents):
        # the kernel-selection switches are part of the key: a graph replays the kernels it was captured with
        return (tuple(latents.shape), latents.dtype, latents.device.index,
                hash(tuple(p.data_ptr() for p in self.net.parameters())), ops.kernel_mode_key())

    @torch.no_grad()
    def sample(self, latents, class_labels=None):
        if self.S_churn != 0:
            raise NotImplementedError("S_churn != 0 is never used by ImagenI2R (sampler.py:26)")
        # fp32 latents (what sampling() draws) or fp64 ones, kept as they are: the reference multiplies
        # latents.to(float64) by t_0 (:45), so a caller's fp64 xT must reach the state unrounded
        keep = latents.dtype if latents.dtype in (torch.float32, torch.float64) else torch.float32
        latents = latents.to(device=self.device, dtype=keep).contiguous()
        if not latents.is_cuda:
            raise ops._lib.I2RError("DiffusionProcess: device must be a HIP ('cuda') device; no CPU fallback")
        t_steps = self._t_steps()
        t_host = [float(t) for t in t_steps]
        was_training = self.net.training
        self.net.eval()
        try:
            if not self.use_graph:
                sig_dev = t_steps.to(device=latents.device, dtype=torch.float32)
                return self._loop(latents, t_host, sig_dev)
            cap = self.max_graph_batch or max(1, 4096 * 256 // (latents.shape[-1] * latents.shape[-2]))
            out = self._sample_any(latents, t_steps, t_host, cap)
            if self.range_check and ops.h2_math():
                out = self._resample_overflowed(latents, out, t_steps, t_host)
            return out
        finally:
            self.net.train(was_training)

    def _resample_overflowed(self, latents, out, t_steps, t_host):
        """The f16x2 kernels hold activations as f16 hi + lo pairs: a magnitude >= 65504 anywhere in the network
        becomes inf there and reaches the output as inf / NaN -- never as a finite wrong number.  The reference's fp32
        convs (models/networks.py:87) have no such limit, so series that came back non-finite are sampled again on
        the exact fp32 kernels; if they are non-finite there too, that is the reference's own result."""
        ok = torch.isfinite(out).flatten(1).all(dim=1)
        if bool(ok.all()):
            return out
        import warnings
        bad = (~ok).nonzero().flatten()
        warnings.warn(f"DiffusionProcess.sample: {bad.numel()} of {out.shape[0]} series left the fp16 range of the "
                      "f16x2 kernels (|activation| >= 65504); they are sampled again on the exact fp32 kernels",
                      RuntimeWarning, stacklevel=3)
        prev = ops.conv_math()
        ops.set_conv_math("f32")
        ops.new_pack_epoch()
        try:
            sig_dev = t_steps.to(device=latents.device, dtype=torch.float32)
            out = out.clone()
            out[bad] = self._loop(latents[bad].contiguous(), t_host, sig_dev)
        finally:
            ops.set_conv_math(prev)
            ops.new_pack_epoch()
        return out

    def _dual(self, m):
        """Two concurrent half-batch graphs from `dual_stream_min` series on -- or from as many POSITIONS as that many
        16x16 images have (1024 series of 32x32: shape D at its bench batch, +1.4 %, tools/r3_dual_d.sh)."""
        if not self.dual_stream_min or m % 2:
            return False
        hw = int(self.shape[-1]) * int(self.shape[-2])
        return m >= self.dual_stream_min or m * hw >= self.dual_stream_min * 512

    def _sample_any(self, latents, t_steps, t_host, cap):
        """Series per captured loop never exceed `cap` (every activation of a piece has to stay under the conv kernels'
        2 GiB buffer-addressing limit).  A request above what one replay takes -- 2*cap as two concurrent half-batch
        graphs, else cap -- is bulk generation (BASELINE config E): it is cut into pieces sampled one after the other;
        an odd piece above cap, which cannot run as two halves, is cut again at cap."""
        n = latents.shape[0]
        if n <= cap or (self._dual(n) and n <= 2 * cap):
            return self._sample_piece(latents, t_steps, t_host)
        piece = 2 * cap if self._dual(2 * cap) else cap
        if n <= piece:
            piece = cap
        return torch.cat([self._sample_any(latents[i:i + piece], t_steps, t_host, cap) for i in range(0, n, piece)])

    def _sample_piece(self, latents, t_steps, t_host):
        n = latents.shape[0]
        if self._dual(n):
            # Two half-batch graphs replayed side by side on two streams: the HBM-bound kernels of one half
            # (GroupNorm, 1x1 convs) run beside the matrix-core-bound 3x3 convs of the other (+2-3 % series/s).
            parts = latents.chunk(2)
            entries = [self._graph_entry(p, t_steps, t_host, slot=i) for i, p in enumerate(parts)]
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=latents.device) for _ in range(2)]
            cur = torch.cuda.current_stream()
            for st, (graph, static_in, _, _), p in zip(self._streams, entries, parts):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    static_in.copy_(p)
                    graph.replay()
            for st in self._streams:
                cur.wait_stream(st)
            return torch.cat([e[2] for e in entries])
        graph, static_in, static_out, _ = self._graph_entry(latents, t_steps, t_host)
        static_in.copy_(latents)
        graph.replay()
        return static_out.clone()

    @torch.no_grad()
    def sampling(self, sampling_number=16, impute=False, xT=None):
        if xT is None:
            xT = torch.randn([sampling_number, *self.shape]).to(device=self.device)  # CPU draw like :70
        return self.sample(xT)
