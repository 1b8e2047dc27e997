"""Multi-GPU partitioning of the hot path over one 8 x MI355X node (one process per GPU, torch.distributed; the "nccl"
backend is RCCL over xGMI on ROCm, "gloo" is used by the CPU tests).

The reference has no distributed code at all; the path shards naturally in two places (§8(e)):

* training: samples are independent and the loss is a mean over all elements, so with equal per-rank batches the
  global gradient is the mean of the rank gradients -> ONE averaged all-reduce of the net gradients per step, done in
  size-bounded buckets that are launched from autograd hooks as soon as a bucket's gradients exist, so the collectives
  overlap the rest of backward (`GradBucketReducer`);
* sampling: every series depends only on its own latent -> shard by noise seed, no collective except a final gather of
  the generated series (`shard_range`, `gather_series`).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of the 413 MB of fp32 gradients of shape C
moves 2*(7/8)*413 MB per GPU; buckets default to 32 MiB so RCCL can pipeline them across links while backward runs.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `total` items owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_series(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather per-rank (n_r, L, C) series into (total, L, C) in rank order (ranks may own n or n+1 items)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    counts = [shard_range(total, r, world) for r in range(world)]
    nmax = max(e - b for b, e in counts)
    padded = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], dim=0)
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded.contiguous(), group=group)
    return torch.cat([o[: e - b] for o, (b, e) in zip(out, counts)], dim=0)


class GradBucketReducer:
    """Bucketed, backward-overlapped gradient averaging for data-parallel training.

    Parameters are walked in REVERSE registration order (the order backward produces their gradients) and packed into
    flat buckets of at most `bucket_bytes`.  Each parameter's ``.grad`` is (re)bound to a view of its bucket, so the
    all-reduce needs no flatten/unflatten copies; a post-accumulate-grad hook counts arrivals and launches the bucket's
    asynchronous all-reduce when it is complete.  `finish()` waits for the outstanding collectives and divides by the
    world size; call it after ``loss.backward()`` and before ``clip_grad_norm_`` / ``optimizer.step()`` (the clip
    norm is then identical on every rank without another collective).
    Works unchanged if the caller uses ``optimizer.zero_grad()`` (grads set to None): the hook re-binds the view.

    Unequal per-rank batches (a loader without ``drop_last``, as the reference builds it, utils/utils_data.py:243, split
    unevenly): the loss is a mean over the LOCAL elements, so the global gradient is the element-count-weighted mean
    sum_r n_r g_r / sum_r n_r, not the plain mean.  Call ``set_local_count(n_local)`` before backward on every rank
    (one 1-element all-reduce, no host sync; or pass ``n_total`` when the caller knows it -- no collective at all):
    each bucket is then scaled by n_r / N before it is summed and ``finish()`` does not divide.  The weighting stays in
    force until ``set_local_count(None)``.

    ``host_stage=True`` is the REHEARSAL form for boxes with fewer GPUs than ranks (ranks share a device, the group is
    gloo): every bucket crosses to a host tensor, is summed there and copied back in ``finish()`` -- the same hooks,
    counters and call sequence as on RCCL, no device tensor is ever handed to gloo.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20, group=None,
                 host_stage: bool = False):
        self.group = group
        self.host_stage = bool(host_stage)
        self._weight = None      # device scalar n_local / N_total, or None = equal batches (plain mean)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.buckets: List[torch.Tensor] = []
        self._slot = {}          # param -> (bucket index, offset)
        self._need: List[int] = []
        self._arrived: List[int] = []
        self._work: List[Optional[object]] = []
        self.defer = False       # True: hooks only adopt/count; collectives start at launch_all() (graph-replayed backward)
        self._next = 0           # first bucket the hooks have not started yet (buckets start in index order)
        cur: List[torch.nn.Parameter] = []
        cur_bytes = 0

        def close():
            nonlocal cur, cur_bytes
            if not cur:
                return
            pad4 = lambda k: (k + 3) // 4 * 4  # keep every view 16-byte aligned (vector loads in the optimiser kernel)
            n = sum(pad4(p.numel()) for p in cur)
            flat = torch.zeros(n, dtype=cur[0].dtype, device=cur[0].device)
            off = 0
            for p in cur:
                self._slot[p] = (len(self.buckets), off)
                off += pad4(p.numel())
            self.buckets.append(flat)
            self._need.append(len(cur))
            self._arrived.append(0)
            self._work.append(None)
            cur, cur_bytes = [], 0

        for p in reversed(self.params):
            if cur and (cur_bytes + p.numel() * p.element_size() > bucket_bytes
                        or p.dtype != cur[0].dtype or p.device != cur[0].device):
                close()
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
        close()
        for p in self.params:  # backward kernels of imageni2r_amd write gradients straight into the bucket (ops.grad_slot)
            p._grad_slot = self._view(p)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _view(self, p):
        b, off = self._slot[p]
        return self.buckets[b][off: off + p.numel()].view_as(p)

    def set_local_count(self, n_local, n_total=None):
        """Weight this rank's gradients by ``n_local / n_total`` (elements or samples -- any unit, as long as every
        rank uses the same one).  ``n_total`` None: summed over the ranks here (a 1-element all-reduce on the
        gradients' device, nothing is read back).  ``n_local`` None: back to the plain mean."""
        if n_local is None:
            self._weight = None
            return
        ref = self.buckets[0]
        w = torch.tensor([float(n_local)], dtype=torch.float32, device=ref.device)
        if n_total is None:
            tot = w.clone()
            if self.world > 1:
                dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
            w = w / tot
        else:
            w = w / float(n_total)
        self._weight = w

    def _start(self, b):
        flat = self.buckets[b]
        if self._weight is not None:
            flat.mul_(self._weight)
        if self.world > 1 and self.host_stage:
            host = flat.cpu()  # (synchronises the stream: rehearsal only)
            self._work[b] = (dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group, async_op=True), host)
        elif self.world > 1:
            self._work[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work[b] = True  # started (world 1: the weighting alone)

    def _on_grad(self, p):
        b, _ = self._slot[p]
        view = self._view(p)
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)      # first step / after zero_grad(set_to_none=True): adopt the bucket storage
            p.grad = view
        self._arrived[b] += 1
        if not self.defer and (self.world > 1 or self._weight is not None):
            # Buckets start strictly in index order on every rank, whatever order their gradients arrive in: a
            # bucket that completes early (a layer's weight before an earlier bucket's last bias) waits for its
            # predecessors.  launch_all() / finish() use the same order, so a rank whose backward was a replayed
            # graph pairs with a rank whose backward fired these hooks.
            while self._next < len(self.buckets) and self._arrived[self._next] >= self._need[self._next]:
                if self._work[self._next] is None:
                    self._start(self._next)
                self._next += 1

    def launch_all(self):
        """Start every bucket's all-reduce now (used when backward ran inside a replayed HIP graph, where the hooks
        do not fire)."""
        if self.world > 1 or self._weight is not None:
            for b in range(len(self.buckets)):
                if self._work[b] is None:
                    self._start(b)

    def finish(self):
        """Wait for every bucket and turn sums into means.  Buckets whose parameters got no gradient this step are
        reduced here (as zeros for the missing ones) so all ranks issue the same collectives."""
        for b, flat in enumerate(self.buckets):
            if self.world > 1 or self._weight is not None:
                if self._work[b] is None:
                    self._start(b)
                if isinstance(self._work[b], tuple):   # host-staged rehearsal: sum arrived on the host, copy it back
                    work, host = self._work[b]
                    work.wait()
                    flat.copy_(host)
                elif self._work[b] is not True:
                    self._work[b].wait()
                if self._weight is None:
                    flat.div_(self.world)  # equal batches: plain mean; weighted: the n_r / N factors already sum to 1
            self._work[b] = None
            self._arrived[b] = 0
        self._next = 0

    def reset(self):
        """Forget the arrivals the hooks counted WITHOUT any collective: for backwards whose gradients are thrown away
        on this rank alone (the warm-up steps before a graph capture -- another rank may not be capturing at all, so a
        collective here would pair with one of its real steps).  Must not be called with a collective in flight."""
        for b in range(len(self.buckets)):
            if self._work[b] is not None and self._work[b] is not True:
                raise RuntimeError("GradBucketReducer.reset: a bucket's all-reduce is in flight; call finish()")
            self._work[b] = None
            self._arrived[b] = 0
        self._next = 0

    def zero_grad(self):
        """Zero the buckets in place (keeps the views bound; cheaper than set_to_none + re-adoption)."""
        for flat in self.buckets:
            flat.zero_()
        for p in self.params:
            p.grad = None  # re-created at the slot's address by the next backward (no accumulate kernels)
        return True

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            if hasattr(p, "_grad_slot"):
                del p._grad_slot
