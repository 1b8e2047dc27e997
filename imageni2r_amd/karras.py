"""TS2img_Karras -- host-side mirror of models/our.py:8-120 on the HIP kernels.

Same constructor (``args`` namespace, ``device``), attributes and methods: ``net`` (EDMPrecond), ``ts_to_img``,
``img_to_ts``, ``loss_fn_irregular``, ``forward_irregular``, ``unpad``, ``ema_scope``, ``on_train_batch_end``,
``model_ema`` / ``use_ema``.  state_dict keys equal the reference's (net.* + model_ema.*).
"""
from __future__ import annotations

import itertools
import math
import os
from contextlib import contextmanager

import torch
import torch.nn as nn

from . import ops
from .delay import DelayEmbedder
from .ema import LitEma
from .networks import EDMPrecond

_noise_counter = itertools.count(1)


class TS2img_Karras(nn.Module):
    def __init__(self, args, device):
        super().__init__()
        self.P_mean = -1.2
        self.P_std = 1.2
        self.sigma_data = 0.5
        self.sigma_min = 0.002
        self.sigma_max = 80
        self.rho = 7
        self.T = args.diffusion_steps
        self.device = device
        self.net = EDMPrecond(args.img_resolution, args.input_channels, channel_mult=args.ch_mult,
                              model_channels=args.unet_channels, attn_resolutions=args.attn_resolution)
        self.delay = args.delay
        self.embedding = args.embedding
        self.seq_len = args.seq_len
        self.batch_size = args.batch_size
        self.num_features = args.input_channels
        self.ts_img = DelayEmbedder(self.device, args.seq_len, args.delay, args.embedding, self.batch_size,
                                    self.num_features)
        if args.ema:
            self.use_ema = True
            self.model_ema = LitEma(self.net, decay=0.9999, use_num_upates=True, warmup=args.ema_warmup)
        else:
            self.use_ema = False
        self._injected = None  # (rnd_normal, noise) for parity tests; None = draw in-kernel

    # -- transforms -------------------------------------------------------------------------------------------
    def ts_to_img(self, signal):
        return self.ts_img.ts_to_img(signal)

    def ts_to_img_with_mask(self, signal):
        return self.ts_img.ts_to_img_with_mask(signal)

    def img_to_ts(self, img):
        return self.ts_img.img_to_ts(img)

    # -- loss ---------------------------------------------------------------------------------------------------
    def inject_randomness(self, rnd_normal=None, noise=None):
        """Test hook: use these instead of the in-kernel Philox draws for the next forward_irregular call(s)."""
        self._injected = None if rnd_normal is None and noise is None else (rnd_normal, noise)

    def loss_fn_irregular(self, x, mask=None):
        """(loss, {'karras loss': float}) exactly like our.py:53-68."""
        to_log = {}
        x = x.to(torch.float32).contiguous()
        if mask is None:
            x, mask = ops.nan_mask(x)  # :60-62
        else:
            mask = mask.to(device=x.device, dtype=torch.float32).contiguous()
        output, weight = self.forward_irregular(x, mask)
        # unpad(.., x.shape) of :64-65 are no-ops; loss = mean(weight * (out*mask - x*mask)^2) (:66)
        loss = ops.MaskedLossFn.apply(output, x.detach(), mask, weight.reshape(-1).contiguous())
        value = loss.detach().item()
        if not math.isfinite(value) and torch.is_grad_enabled() and ops.train_fwd_h2_enabled():
            # fp16 range guard: a training forward on the f16x2 kernels met |activation| >= 65504 (inf there, finite
            # in the reference's fp32 convs).  From here on every training forward uses the exact fp32 kernels.
            import warnings
            warnings.warn("loss_fn_irregular: non-finite loss with f16x2 training forwards (|activation| >= 65504?); "
                          "switching the training forward to the exact fp32 kernels and recomputing this step",
                          RuntimeWarning, stacklevel=2)
            ops.set_train_fwd_h2("0")
            ops.new_pack_epoch()
            output, weight = self.forward_irregular(x, mask, _reuse_offset=True)  # the SAME sigma / noise draw
            loss = ops.MaskedLossFn.apply(output, x.detach(), mask, weight.reshape(-1).contiguous())
            value = loss.detach().item()
        to_log['karras loss'] = value
        return loss, to_log

    def forward_irregular(self, x, mask, labels=None, augment_pipe=None, _reuse_offset=False):
        """sigma ~ exp(N(P_mean, P_std)); noise only on observed pixels; returns (D(x + n*mask, sigma), weight)
        (our.py:70-78).  ``_reuse_offset``: draw from the Philox offset of the previous call again (the fp32
        recomputation of a step whose f16x2 forward overflowed must see the same sigma and noise)."""
        if augment_pipe is not None or labels is not None:
            raise NotImplementedError("augment_pipe / labels are never passed by ImagenI2R")
        rnd, noise = self._injected if self._injected is not None else (None, None)
        seed = ops.rng_seed()
        step = ops.rng_step()
        if step is not None:
            offset = 0
        elif _reuse_offset and getattr(self, "_last_noise_offset", None) is not None:
            offset = self._last_noise_offset
        else:
            offset = self._last_noise_offset = next(_noise_counter)
        xn, sigma, weight = ops.edm_noise(x.detach().contiguous(), mask, self.P_mean, self.P_std, self.sigma_data,
                                          rnd=rnd, noise=noise, seed=seed, offset=offset, step=step)
        D_yn = self.net(xn, sigma)
        return D_yn, weight.reshape(-1, 1, 1, 1)

    def unpad(self, x, original_shape):
        _, _, original_cols, original_rows = original_shape
        return x[:, :, :original_cols, :original_rows]

    # -- EMA ----------------------------------------------------------------------------------------------------
    @contextmanager
    def ema_scope(self, context=None):
        """Run the body with the EMA weights in ``self.net`` (our.py:87-108).  The reference stores a copy of the
        training weights, copies the shadow weights in and copies the stored ones back; here the two sets exchange
        storage (LitEma.swap) -- same weights inside and after the scope, no copies.  ``I2R_EMA_SCOPE=copy`` keeps the
        reference's sequence (needed only by a caller that reads ``model_ema``'s buffers *inside* the scope)."""
        swap = self.use_ema and os.environ.get("I2R_EMA_SCOPE", "swap") != "copy"
        if self.use_ema:
            if swap:
                self.model_ema.swap(self.net)
            else:
                self.model_ema.store(self.net.parameters())
                self.model_ema.copy_to(self.net)
            if context is not None:
                print(f"{context}: Switched to EMA weights")
        try:
            yield None
        finally:
            if self.use_ema:
                if swap:
                    self.model_ema.swap(self.net)
                else:
                    self.model_ema.restore(self.net.parameters())
                if context is not None:
                    print(f"{context}: Restored training weights")

    def on_train_batch_end(self, *args):
        if self.use_ema:
            self.model_ema(self.net)
