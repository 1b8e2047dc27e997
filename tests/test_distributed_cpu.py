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
