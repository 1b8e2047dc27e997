"""GraphedTrainStep -- the diffusion train step of run_irregular.py:215-230 as ONE replayed HIP graph.

    x_img, mask = model.ts_to_img_with_mask(x_ts)          # :215-216
    x_tilde_img = model.ts_to_img(x_recon)                 # :219   (TST output; stays on PyTorch-ROCm, outside)
    loss = step(x_tilde_img, mask)                         # :220-230 = forward_irregular + masked loss + backward
                                                           #            + clip + AdamW + EMA

Everything between the two inputs and the updated weights is captured once (torch.cuda.graph drives the stream capture;
the library never allocates or synchronises) and replayed: ~2.5k kernel launches per step cost one graph launch on the
host.  What makes that legal: the per-step random streams (sigma / noise draws, dropout masks) read a device-resident
step counter that the graph itself increments, the conv weight packing is part of the graph (replays always read the
live weights), and the optimiser tail is `FusedAdamW` whose step count / bias corrections / EMA decay live on the device.
The only thing left out is the reference's per-step ``loss.item()`` host sync (our.py:67): the loss comes back as a
device scalar; call ``.item()`` when you log.

Data parallel: with a `GradBucketReducer` the capture is split after backward -- graph 1 (forward + backward), the
bucketed RCCL all-reduce launched eagerly, graph 2 (optimiser tail) -- no collective is captured, and none is ISSUED
while capturing either (the warm-up steps run rank-local): every rank issues exactly one round of bucket all-reduces per
call, whether that call captures, replays or runs eagerly, so ranks that first meet a new input shape on different
steps (a short last batch on one rank only) stay paired.
"""
from __future__ import annotations

import torch

from . import _lib, ops
from .optim import FusedAdamW


class GraphedTrainStep:
    def __init__(self, model, optimizer: FusedAdamW, reducer=None, use_graph=True, apply_update=True):
        """``apply_update=False`` leaves the optimiser tail (clip + AdamW + EMA) out: the call is then
        run_irregular.py:215-227 only -- gradients are left in the parameters' slots for the caller (gradient
        accumulation, a foreign optimiser, the "forward + backward only" measurement of bench.py)."""
        if not isinstance(optimizer, FusedAdamW):
            raise TypeError("GraphedTrainStep needs a FusedAdamW optimiser (device-resident step state)")
        self.model = model
        self.opt = optimizer
        self.reducer = reducer
        self.use_graph = use_graph
        self.apply_update = bool(apply_update)
        self.range_check_every = 64   # calls between looks at the optimiser's skipped-step count (one host sync each)
        self.max_skipped_exact = 8    # skipped steps per look tolerated once the forward is already exact fp32
        self._calls = 0
        self._skipped_seen = 0
        self._entries = {}  # input shape -> captured step (the short last batch of an epoch gets its own, once)
        dev = next(model.net.parameters()).device
        if dev.type != "cuda":
            raise _lib.I2RError("GraphedTrainStep: the model must live on a HIP ('cuda') device; no CPU fallback")
        self.rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)

    # ------------------------------------------------------------------------------------------------------
    def _fwd_bwd(self, x_tilde_img, mask):
        m = self.model
        ops.counter_add(self.rng_counter, 1)
        zeroed = self.reducer.zero_grad() if self.reducer is not None else self.opt.zero_grad()
        m.net.model.pack_all_weights(dgrad=True)  # one launch instead of two pack kernels per conv
        ops.set_slots_zeroed(bool(zeroed))        # weight-gradient kernels add into the zeroed slots: no memsets
        try:
            output, weight = m.forward_irregular(x_tilde_img, mask)
            loss = ops.MaskedLossFn.apply(output, x_tilde_img, mask, weight.reshape(-1).contiguous())
            loss.backward()
        finally:
            ops.set_slots_zeroed(False)
        return loss.detach()

    def _eager(self, x_tilde_img, mask, local_only=False):
        """One uncaptured step.  ``local_only``: no collective at all (the throw-away warm-up steps of a capture --
        ranks capture when THEY first see a shape, so a collective here could pair with another rank's real step)."""
        if self.opt._table_key is None:
            self.opt.prepare()  # gradient slots exist from the FIRST step on: eager and replayed steps run the same kernels
        prev = ops.rng_step()
        ops.set_rng_step(self.rng_counter)
        red = self.reducer
        was_deferred = red.defer if red is not None else False
        try:
            if red is not None and local_only:
                red.defer = True   # hooks adopt and count, nothing is launched
            loss = self._fwd_bwd(x_tilde_img, mask)
            if red is not None:
                if local_only:
                    red.reset()
                else:
                    red.finish()
            if self.apply_update:
                self.opt.step()
        finally:
            ops.set_rng_step(prev)
            if red is not None:
                red.defer = was_deferred
        return loss

    def _generation(self):
        return (self.opt.state_generation, self.opt.table_generation)

    def _training_state(self):
        """Every tensor a train step writes: weights, Adam moments, the shared step counter, the optimiser's scalars,
        EMA shadow weights + update counter, and the RNG step counter."""
        ps = self.opt._params()
        ts = list(ps)
        for p in ps:
            st = self.opt.state[p]
            ts += [st["exp_avg"], st["exp_avg_sq"]]
        ts += [self.opt.state[ps[0]]["step"], self.opt.scalars, self.rng_counter]
        if getattr(self.model, "use_ema", False):
            ts += [b for b in self.model.model_ema.buffers()]
        return [t.detach() for t in ts]

    def _capture(self, x_tilde_img, mask):
        self.opt.prepare()
        static_x = x_tilde_img.clone()
        static_m = mask.clone()
        side = torch.cuda.Stream(device=static_x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            # Warm-up (allocator, lazy state, autograd graph shape) runs two REAL steps -- and must leave no trace:
            # the batch that triggers a capture gets exactly one update like every other batch (the reference's
            # loop, run_irregular.py:220-230, has no such thing as a warm-up).  Snapshot, step twice, put back.
            live = self._training_state()
            saved = [t.clone() for t in live]
            with torch.enable_grad():
                for _ in range(2):
                    self._eager(static_x, static_m, local_only=True)  # rank-local: gradients are not averaged
            torch._foreach_copy_(live, saved)
            del saved
            if getattr(self.model, "use_ema", False):
                self.model.model_ema._n = None  # host mirror of num_updates: re-read from the restored buffer
        torch.cuda.current_stream().wait_stream(side)
        prev = ops.rng_step()
        ops.set_rng_step(self.rng_counter)
        if self.reducer is not None:
            self.reducer.defer = True  # no collective inside the capture; launch_all() after each replay
        try:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                ops.new_pack_epoch()  # weights are packed inside the graph: replays read the live weights
                loss = self._fwd_bwd(static_x, static_m)
                if self.reducer is None and self.apply_update:
                    self.opt.step()
            g2 = None
            if self.reducer is not None and self.apply_update:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g1.pool()):
                    self.opt.step()
            ops.new_pack_epoch()
        finally:
            ops.set_rng_step(prev)
            if self.reducer is not None:
                self.reducer.defer = False
                self.reducer.reset()  # the arrivals the capture-time hooks counted; NO collective (see _eager)
        entry = (g1, g2, static_x, static_m, loss, self._generation())
        self._entries[tuple(static_x.shape)] = entry
        return entry

    def check_range(self) -> int:
        """fp16 range guard for the replayed step (no per-step host sync): the fused optimiser SKIPS a step whose
        gradient norm is not finite and counts it.  If steps were skipped since the last look and training forwards
        run on the f16x2 kernels (operands must stay below 65504 there; the reference's fp32 convs have no such
        limit), the forward is switched to the exact fp32 kernels and the step is captured again.  Returns the number
        of newly skipped steps.  Called automatically every ``range_check_every`` steps; call it when logging."""
        n = self.opt.skipped_steps()
        new = n - self._skipped_seen
        self._skipped_seen = n
        if new > 0:
            import warnings
            if ops.train_fwd_h2_enabled():
                warnings.warn(f"GraphedTrainStep: {new} step(s) had a non-finite gradient norm and were skipped; the "
                              "training forward leaves the f16x2 kernels (fp16 range) for the exact fp32 kernels",
                              RuntimeWarning, stacklevel=2)
                ops.set_train_fwd_h2("0")
                ops.new_pack_epoch()
                self._entries.clear()
            else:
                # nothing left to fall back to: the reference would have written NaN into every weight at the first
                # such batch (clip_grad_norm_ + AdamW, run_irregular.py:228-229); here the updates were skipped
                if new > self.max_skipped_exact:
                    raise FloatingPointError(
                        f"GraphedTrainStep: {new} of the last {self.range_check_every} steps had a non-finite "
                        "gradient norm although the forward runs on the exact fp32 kernels: the loss itself is not "
                        "finite (NaN / inf in the data?).  The updates of those steps were skipped.")
                warnings.warn(f"GraphedTrainStep: {new} step(s) had a non-finite gradient norm and were skipped "
                              "(the forward is already exact fp32: the loss itself is not finite)",
                              RuntimeWarning, stacklevel=2)
        return new

    def last_step_skipped(self):
        """Device scalar (1.0 = the optimiser skipped the last step: non-finite gradient norm); read it with the
        loss when logging -- no extra sync is made here."""
        return self.opt.last_step_skipped()

    def __call__(self, x_tilde_img, mask):
        """One full train step; returns the loss as a 0-dim device tensor (valid until the next call)."""
        self._calls += 1
        if self.range_check_every and self._calls % self.range_check_every == 0:
            self.check_range()
        x_tilde_img = x_tilde_img.detach().to(torch.float32).contiguous()
        mask = mask.to(device=x_tilde_img.device, dtype=torch.float32).contiguous()
        if not self.model.training:
            raise RuntimeError("GraphedTrainStep: call model.train() first (dropout is part of the captured step)")
        if not self.use_graph:
            return self._eager(x_tilde_img, mask)
        entry = self._entries.get(tuple(x_tilde_img.shape))
        if entry is not None and entry[5] != self._generation():
            # the optimiser's device pointer tables were rebuilt since (load_state_dict() swapped the moment tensors,
            # something re-bound a gradient): every captured optimiser tail points at freed tables -- capture again
            self._entries.clear()
            entry = None
        if entry is None:
            entry = self._capture(x_tilde_img, mask)
        g1, g2, static_x, static_m, loss, _ = entry
        static_x.copy_(x_tilde_img)
        static_m.copy_(mask)
        g1.replay()
        if self.reducer is not None:
            self.reducer.launch_all()
            self.reducer.finish()
        if g2 is not None:
            g2.replay()
        return loss
