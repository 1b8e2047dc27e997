"""LitEma -- same buffers, names and update rule as models/ema.py:5-79 (kept API-compatible; SURVEY.md §8(f) row 1
lists the fused multi-tensor optimiser tail as the next component, so this is host plumbing for now)."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn


class LitEma(nn.Module):
    def __init__(self, model, decay=0.9999, use_num_upates=True, warmup=0):
        super().__init__()
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.m_name2s_name = {}
        self.register_buffer('decay', torch.tensor(decay, dtype=torch.float32))
        self.register_buffer('num_updates', torch.tensor(0 if use_num_upates else -1, dtype=torch.int))
        for name, p in model.named_parameters():
            if p.requires_grad:
                s_name = name.replace('.', '')  # '.' is not allowed in buffer names (:19)
                self.m_name2s_name[name] = s_name
                self.register_buffer(s_name, p.clone().detach().data)
        self.collected_params = []
        self.warmup = warmup
        self._n = None  # host copy of num_updates (read once from the buffer, then tracked without device syncs)
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_n", None))

    def _pairs(self, model):
        shadow = dict(self.named_buffers())
        return [(shadow[self.m_name2s_name[k]], p) for k, p in model.named_parameters() if p.requires_grad]

    def forward(self, model):
        if getattr(self, "fused_into_optimizer", False):
            return  # FusedAdamW.step() already advanced the shadow weights in its own kernel
        return self._update(model)

    def _update(self, model):
        """One EMA step (:26-47): copy while num_updates <= warmup, else s -= (1 - decay) * (s - p) with
        decay = min(0.9999, (1 + n) / (10 + n)) evaluated in fp32 like the reference's 0-dim tensors."""
        if self._n is None:
            self._n = int(self.num_updates)
            self._decay0 = np.float32(float(self.decay))
        decay = self._decay0
        if self._n >= 0:
            self._n += 1
            self.num_updates += 1
            decay = min(decay, np.float32(1 + self._n) / np.float32(10 + self._n))
        one_minus_decay = float(np.float32(1.0) - decay)
        with torch.no_grad():
            pairs = self._pairs(model)
            shadows = [s for s, _ in pairs]
            params = [p.detach() for _, p in pairs]
            if self._n > self.warmup:
                diffs = torch._foreach_sub(shadows, params)
                torch._foreach_mul_(diffs, one_minus_decay)
                torch._foreach_sub_(shadows, diffs)
            else:
                torch._foreach_copy_(shadows, params)

    def copy_to(self, model):
        from . import ops
        with torch.no_grad():
            for s, p in self._pairs(model):
                p.data.copy_(s.data)
        ops.new_pack_epoch()  # .data writes do not bump _version: drop every cached packed conv weight

    def swap(self, model):
        """Exchange the storage of every parameter with its shadow buffer (no copy, no extra memory): the module then
        computes with the EMA weights and the shadow buffers hold the training weights, until the next ``swap``.
        What ``ema_scope`` uses instead of the reference's store + copy_to + restore (models/our.py:87-108: three
        whole-model copies per evaluation).  Addresses come back exactly after the second swap, so pointer tables
        built outside the scope (FusedAdamW, captured graphs) stay valid."""
        from . import ops
        with torch.no_grad():
            for s, p in self._pairs(model):
                if s.shape != p.shape or s.dtype != p.dtype or s.device != p.device:
                    raise RuntimeError("LitEma.swap: shadow buffer and parameter differ in shape/dtype/device")
                held = p.data
                p.data = s.data
                s.data = held
        ops.new_pack_epoch()

    def store(self, parameters):
        self.collected_params = [param.clone() for param in parameters]

    def restore(self, parameters):
        from . import ops
        for c_param, param in zip(self.collected_params, parameters):
            param.data.copy_(c_param.data)
        ops.new_pack_epoch()
