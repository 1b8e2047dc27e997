"""FusedAdamW -- clip_grad_norm_ + AdamW + LitEma of the diffusion train step (run_irregular.py:228-230,
models/ema.py:26-47) as ONE multi-tensor HIP step (csrc/optim.hip), SURVEY.md §8(f) row 1.

Drop-in for ``torch.optim.AdamW(model.parameters(), lr, weight_decay)`` as the reference constructs it
(run_irregular.py:117): same hyper-parameters, same update rule, same per-parameter state names (``step``, ``exp_avg``,
``exp_avg_sq``).  Optional extras: ``max_grad_norm`` folds ``torch.nn.utils.clip_grad_norm_(params, max_grad_norm)``
into the step, ``ema=(LitEma, net)`` folds ``model.on_train_batch_end()`` into it (the LitEma module is flagged so its
own forward becomes a no-op).  Every per-step scalar lives on the device, so ``step()`` is HIP-graph capturable.

Gradients must keep their addresses between steps (the pointer table is built once): ``zero_grad()`` therefore zeroes
one flat gradient buffer in place instead of dropping the tensors; if something else re-binds ``p.grad`` the table is
rebuilt on the next step.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib, ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 ema=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdamW: one parameter group (as run_irregular.py:117 builds it)")
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self._ema, self._ema_net = (ema if ema is not None else (None, None))
        if self._ema is not None:
            self._ema.fused_into_optimizer = True
        self._table_key = None
        self.state_generation = 0  # bumped when load_state_dict swaps the state tensors
        self.table_generation = 0  # bumped whenever the device pointer tables are rebuilt (captured graphs go stale)
        self._flat_grad = None
        self.scalars = None  # device float[8]: [0] total grad norm of the last step, [1] its clip coefficient,
                             # [6] steps skipped for a non-finite norm, [7] 1 if the last step was skipped

    # ------------------------------------------------------------------------------------------------------
    def _params(self):
        return [p for p in self.param_groups[0]["params"] if p.requires_grad]

    def _ensure_state(self, ps):
        dev = ps[0].device
        if self.scalars is None:
            self.scalars = torch.zeros(8, dtype=torch.float32, device=dev)
        first = self.state.get(ps[0], {})
        step = first.get("step")
        if step is None or not torch.is_tensor(step) or step.device != dev or step.dtype != torch.float32:
            step = torch.tensor(float(step) if step is not None else 0.0, dtype=torch.float32, device=dev)
        for p in ps:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise _lib.I2RError("FusedAdamW: parameters must be contiguous fp32 HIP tensors (no CPU fallback)")
            st = self.state[p]
            st["step"] = step  # one shared device counter
            for k in ("exp_avg", "exp_avg_sq"):
                if k not in st:
                    st[k] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif st[k].device != dev:
                    st[k] = st[k].to(dev)
        return step

    def bind_grads(self):
        """Give every parameter that has neither a gradient nor a gradient slot a slot in one flat buffer.  A slot
        (``p._grad_slot``) is where the backward kernels write that parameter's gradient (ops.grad_slot); autograd
        then adopts it as ``p.grad`` without a copy, so gradient addresses are the same on every step."""
        ps = self._params()
        missing = [p for p in ps if p.grad is None and getattr(p, "_grad_slot", None) is None]
        if missing:
            n = sum((p.numel() + 3) // 4 * 4 for p in missing)
            flat = torch.zeros(n, dtype=torch.float32, device=missing[0].device)
            off = 0
            for p in missing:
                p._grad_slot = flat[off: off + p.numel()].view_as(p)
                off += (p.numel() + 3) // 4 * 4
            self._flat_grad = flat if len(missing) == len(ps) else None
        for p in ps:
            if p.grad is None:
                p.grad = p._grad_slot.detach()
        return ps

    def zero_grad(self, set_to_none: bool = True):
        """Parameters with a gradient slot: drop ``.grad`` (backward re-creates it at the slot's address) and zero the
        flat slot buffer with one memset; others: zero in place (their addresses are in the kernel's table)."""
        ps = self._params()
        if self._flat_grad is not None and all(getattr(p, "_grad_slot", None) is not None for p in ps):
            self._flat_grad.zero_()
            for p in ps:
                p.grad = None
            return True
        for p in ps:
            if getattr(p, "_grad_slot", None) is not None:
                p.grad = None
        grads = [p.grad for p in ps if p.grad is not None]
        if grads:
            torch._foreach_zero_(grads)
        return False

    def _build_table(self, ps):
        shadows = None
        if self._ema is not None:
            bufs = dict(self._ema.named_buffers())
            by_param = {id(p): bufs[self._ema.m_name2s_name[k]] for k, p in self._ema_net.named_parameters()
                        if p.requires_grad}
            shadows = [by_param.get(id(p)) for p in ps]
        chunk = _lib.query("i2r_opt_chunk_elems")
        rows, chunks = [], []
        for t, p in enumerate(ps):
            st = self.state[p]
            sh = shadows[t] if shadows is not None else None
            rows.append([p.data_ptr(), self._gptr(p), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                         sh.data_ptr() if sh is not None else 0, p.numel()])
            chunks.extend([t, c] for c in range((p.numel() + chunk - 1) // chunk))
        dev = ps[0].device
        self._table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self._chunks = torch.tensor(chunks, dtype=torch.int32).to(dev)
        self._partials = torch.empty(len(chunks), dtype=torch.float32, device=dev)
        self._nchunks = len(chunks)
        self.table_generation += 1  # graphs captured against the previous tables are stale from here on
        self._ema_decay = float(self._ema.decay) if self._ema is not None else 0.0  # one host read, at table build

    @staticmethod
    def _gptr(p):
        if p.grad is not None:
            return p.grad.data_ptr()
        return p._grad_slot.data_ptr()

    def _key(self, ps):
        """Everything the device pointer table holds: parameter, gradient, both moments (load_state_dict replaces those
        tensors) and, through the generation, the EMA shadows."""
        return (self.state_generation,) + tuple(
            (p.data_ptr(), self._gptr(p), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
            for p in ps)

    def load_state_dict(self, state_dict):
        """torch's loader swaps in new ``exp_avg`` / ``exp_avg_sq`` / ``step`` tensors: the pointer table is rebuilt on
        the next step and GraphedTrainStep re-captures (``state_generation``)."""
        super().load_state_dict(state_dict)
        self._table_key = None
        self.state_generation += 1

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FusedAdamW.step: closures are not supported")
        ps = self._params()
        if any(p.grad is None and getattr(p, "_grad_slot", None) is None for p in ps):
            ps = self.bind_grads()
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:  # the table cannot change inside a capture; it was validated when the capture was made
            step = self._ensure_state(ps)
            key = self._key(ps)
            if key != self._table_key:
                self._build_table(ps)
                self._table_key = key
        else:
            step = self.state[ps[0]]["step"]
            if self._table_key is None:
                raise _lib.I2RError("FusedAdamW: run one eager step (or prepare()) before capturing a graph")
        g = self.param_groups[0]
        ema = self._ema
        vp = ctypes.c_void_p
        _lib.call("i2r_fused_adamw_ema_step", vp(self._table.data_ptr()), vp(self._chunks.data_ptr()), self._nchunks,
                  vp(self._partials.data_ptr()), vp(self.scalars.data_ptr()), vp(step.data_ptr()),
                  vp(ema.num_updates.data_ptr()) if ema is not None else None, self.max_grad_norm, float(g["lr"]),
                  float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                  self._ema_decay,
                  int(ema.warmup) if ema is not None else 0, ops._stream())
        if ema is not None:
            ema._n = None  # host mirror of num_updates is stale now
        ops.new_pack_epoch()  # raw-pointer writes do not bump Parameter._version: drop cached packed conv weights
        return None

    def prepare(self):
        """Build state + pointer table without stepping (call before capturing ``step()`` into a graph).  The device
        tables are rebuilt ONLY when an address they hold changed: a captured graph keeps raw pointers to them, and
        replacing the tensors would leave an earlier capture replaying against freed memory (``table_generation``
        tells GraphedTrainStep when that happened and its graphs must be captured again)."""
        ps = self.bind_grads()
        self._ensure_state(ps)
        key = self._key(ps)
        if key != self._table_key:
            self._build_table(ps)
            self._table_key = key
        return self

    def skipped_steps(self) -> int:
        """Steps the kernel skipped because the total gradient norm was inf / NaN (host sync; only counted when
        ``max_grad_norm`` is set).  Nothing -- weights, moments, step count, EMA -- is touched by a skipped step."""
        return 0 if self.scalars is None else int(self.scalars[6].item())

    def last_step_skipped(self):
        """Device scalar, 1.0 if the last step was skipped for a non-finite gradient norm (no host sync: read it
        together with the loss when logging -- ``bool(opt.last_step_skipped())`` syncs)."""
        return self.scalars[7]

    def grad_norm(self):
        """Total gradient norm measured by the last step (device scalar, as clip_grad_norm_ would have returned)."""
        return self.scalars[0]
