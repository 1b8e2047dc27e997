"""world_size-2 gloo tests (CPU) of the multi-GPU partitioning: bucketed gradient averaging == full-batch gradient,
seed-sharded sampling union == single-rank output."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imageni2r_amd.parallel import GradBucketReducer, gather_series, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.SiLU(), torch.nn.Linear(16, 16), torch.nn.SiLU(),
                               torch.nn.Linear(16, 3))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        x = torch.randn(8, 6)
        y = torch.randn(8, 3)
        m = _model()
        red = GradBucketReducer(m.parameters(), bucket_bytes=600)  # tiny buckets -> several collectives
        assert len(red.buckets) >= 3
        b, e = shard_range(8, rank, world)
        for step in range(2):  # second step exercises zero_grad(set_to_none=True) re-adoption
            m.zero_grad()
            loss = ((m(x[b:e]) - y[b:e]) ** 2).mean()
            loss.backward()
            red.finish()
        grads = [p.grad.clone() for p in m.parameters()]
        # seed-sharded "sampling": each rank produces its slice of a deterministic function of the item index
        total = 7
        sb, se = shard_range(total, rank, world)
        local = torch.stack([torch.full((4, 2), float(i)) for i in range(sb, se)]) if se > sb else torch.empty(0, 4, 2)
        allser = gather_series(local, total)
        # plain numpy (pickled by value): torch tensors in an mp.Queue are shared-memory handles that can die with
        # the producer process
        q.put((rank, [g.numpy() for g in grads], allser.numpy()))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_sharded_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # reference: full-batch gradient on one process
    torch.manual_seed(0)
    x = torch.randn(8, 6)
    y = torch.randn(8, 3)
    m = _model()
    ((m(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in m.parameters()]
    for rank, grads, allser in res:
        for g, r in zip(grads, ref):
            assert torch.allclose(torch.from_numpy(g), r, rtol=1e-5, atol=1e-7), rank
        assert allser.shape == (7, 4, 2)
        assert (allser[:, 0, 0] == list(range(7))).all()


def test_shard_range_partitions():
    for total in (0, 1, 7, 8, 100000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_reducer_single_process_is_identity():
    m = _model()
    red = GradBucketReducer(m.parameters())
    x = torch.randn(4, 6)
    m(x).sum().backward()
    ref = [p.grad.clone() for p in m.parameters()]
    red.finish()
    assert all(torch.equal(a, p.grad) for a, p in zip(ref, m.parameters()))
    red.zero_grad()  # buckets zeroed in place, .grad dropped: the next backward lands at the slot address again
    assert all(p.grad is None for p in m.parameters())
    assert all((b == 0).all() for b in red.buckets)
    m(x).sum().backward()
    assert all(p.grad.data_ptr() == p._grad_slot.data_ptr() for p in m.parameters())
    assert all(torch.equal(a, p.grad) for a, p in zip(ref, m.parameters()))


# ------------------------------------------------------------------------------------------------------------------
# The real gradient path of imageni2r_amd: backward "kernels" WRITE the gradient into the parameter's slot (a view of
# the reducer's bucket) and return an alias of it, which autograd adopts as .grad without a copy.  No HIP on the CPU,
# so a stand-in autograd.Function does exactly what ops.Conv2dFn / LinearFn / GroupNormFn do with ops.param_slot().
class _SlotLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        from imageni2r_amd import ops
        ctx.save_for_backward(x, w)
        ctx.slots = (ops.param_slot(w), ops.param_slot(b))
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        sw, sb = ctx.slots
        dw, db = dy.t() @ x, dy.sum(0)
        if sw is not None and sb is not None:
            sw.detach().copy_(dw)  # "the kernel writes the slot"
            sb.detach().copy_(db)
            dw, db = sw.detach(), sb.detach()
        return dy @ w, dw, db


class _SlotNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(11)
        self.w1 = torch.nn.Parameter(torch.randn(16, 6) * 0.3)
        self.b1 = torch.nn.Parameter(torch.randn(16) * 0.1)
        self.w2 = torch.nn.Parameter(torch.randn(3, 16) * 0.3)
        self.b2 = torch.nn.Parameter(torch.randn(3) * 0.1)

    def forward(self, x):
        return _SlotLinear.apply(torch.tanh(_SlotLinear.apply(x, self.w1, self.b1)), self.w2, self.b2)


def _worker_unequal(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imageni2r_amd import ops
        torch.manual_seed(0)
        x = torch.randn(8, 6)
        y = torch.randn(8, 3)
        spans = [(0, 5), (5, 8)]  # the short last batch of an epoch, split unevenly: 5 + 3 samples
        b, e = spans[rank]
        out = {}
        for deferred in (False, True):
            m = _SlotNet()
            red = GradBucketReducer(m.parameters(), bucket_bytes=300)
            assert len(red.buckets) >= 2
            for step in range(2):
                red.zero_grad()
                red.set_local_count(e - b)           # elements are (e - b) * 3 on every rank: samples are the unit
                red.defer = deferred
                loss = ((m(x[b:e]) - y[b:e]) ** 2).mean()
                loss.backward()
                assert all(p.grad.data_ptr() == p._grad_slot.data_ptr() for p in m.parameters())  # written in place
                if deferred:
                    red.launch_all()                 # what GraphedTrainStep does after replaying its backward graph
                red.finish()
            out[deferred] = [p.grad.clone().numpy() for p in m.parameters()]
            red.remove()
        # plain mean (no weighting) with equal halves still works through the slot path
        m = _SlotNet()
        red = GradBucketReducer(m.parameters(), bucket_bytes=300)
        hb, he = shard_range(8, rank, world)
        ((m(x[hb:he]) - y[hb:he]) ** 2).mean().backward()
        red.finish()
        out["equal"] = [p.grad.clone().numpy() for p in m.parameters()]
        # training RNG streams: same torch seed on every rank (identical initial weights), different Philox keys
        out["seed"] = ops.rng_seed()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_unequal_batches_slot_path_and_rank_rng_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_unequal, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    x = torch.randn(8, 6)
    y = torch.randn(8, 3)
    m = _SlotNet()
    ((m(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in m.parameters()]
    for rank in range(world):
        for key in (False, True, "equal"):
            for g, r in zip(res[rank][key], ref):
                assert torch.allclose(torch.from_numpy(g), r, rtol=1e-5, atol=1e-7), (rank, key)
    assert res[0]["seed"] != res[1]["seed"]
    assert res[0]["seed"] == torch.initial_seed() & ((1 << 63) - 1)  # rank 0 keeps the single-process streams


def _worker_asym_capture(rank, world, port, q):
    """What GraphedTrainStep does when only ONE rank meets a new input shape (a short last batch on that rank alone):
    that rank runs two throw-away warm-up steps (`_eager(local_only=True)`: hooks deferred, `reset()` instead of
    `finish()`) and a capture pass before its real step; the other rank goes straight to its real step.  Both must
    issue the same collectives -- a warm-up that all-reduced would pair with the other rank's REAL buckets."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        torch.manual_seed(0)
        x = torch.randn(8, 6)
        y = torch.randn(8, 3)
        b, e = shard_range(8, rank, world)
        m = _SlotNet()
        red = GradBucketReducer(m.parameters(), bucket_bytes=300)
        calls = []
        orig_start = red._start
        red._start = lambda bi: (calls.append(bi), orig_start(bi))[1]
        out = {}
        for step in range(2):
            capturing = rank == step  # rank 0 "captures" at step 0, rank 1 at step 1
            if capturing:
                for _ in range(2):    # warm-up: rank-local, other data, gradients discarded
                    red.zero_grad()
                    red.defer = True
                    ((m(x[b:e] * 3.0 + 1.0) - y[b:e]) ** 2).mean().backward()
                    red.reset()
                    red.defer = False
                red.zero_grad()       # the capture pass itself: hooks deferred, counters reset, nothing launched
                red.defer = True
                ((m(x[b:e]) - y[b:e]) ** 2).mean().backward()
                red.defer = False
                red.reset()
                assert not calls, "a capture must not issue a collective"
                red.launch_all()      # the replay that follows the capture
            else:
                red.zero_grad()
                ((m(x[b:e]) - y[b:e]) ** 2).mean().backward()
            red.finish()
            out[step] = ([p.grad.clone().numpy() for p in m.parameters()], len(calls))
            calls.clear()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_capture_on_one_rank_only_issues_no_extra_collective_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_asym_capture, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    x = torch.randn(8, 6)
    y = torch.randn(8, 3)
    m = _SlotNet()
    ((m(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in m.parameters()]
    for rank in range(world):
        for step in range(2):
            grads, ncalls = res[rank][step]
            assert ncalls == res[1 - rank][step][1] > 0, (rank, step)  # same number of collectives on both ranks
            for g, r in zip(grads, ref):
                assert torch.allclose(torch.from_numpy(g), r, rtol=1e-5, atol=1e-7), (rank, step)


def test_reducer_reset_refuses_with_a_collective_in_flight():
    m = _model()
    red = GradBucketReducer(m.parameters())
    red._work[0] = object()
    with pytest.raises(RuntimeError):
        red.reset()


def test_gradient_accumulation_over_two_backwards_sums_in_the_slot():
    """A second backward without zero_grad: the parameter already holds a gradient, so the kernels must not write the
    slot again (autograd would add the alias to itself and double the LAST gradient)."""
    m = _SlotNet()
    red = GradBucketReducer(m.parameters())
    x1, x2 = torch.randn(4, 6), torch.randn(4, 6)
    m(x1).sum().backward()
    g1 = [p.grad.clone() for p in m.parameters()]
    m(x2).sum().backward()
    red.zero_grad()
    m(x2).sum().backward()
    g2 = [p.grad.clone() for p in m.parameters()]
    red.zero_grad()
    m(x1).sum().backward()
    m(x2).sum().backward()
    for p, a, b in zip(m.parameters(), g1, g2):
        assert torch.allclose(p.grad, a + b, rtol=1e-6, atol=1e-6)
        assert p.grad.data_ptr() == p._grad_slot.data_ptr()
