"""GPU tests of (i) the data-parallel train path with the real kernels at world size 1 (two-graph split around the
bucketed all-reduce, gradient slots inside the reducer's buckets, deferred launch), (ii) the capture bookkeeping of
GraphedTrainStep (one update per batch, one graph per input shape, re-capture after load_state_dict), (iii) gradient
accumulation over two backwards on the slot path, and (iv) the in-kernel Philox streams (sigma, noise, dropout):
distribution, independence across steps / layers / ranks."""
import numpy as np
import pytest
import torch

from oracle import i2r_oracle as O
from tests.helpers import rel_err, rms_rel_err, tiny_args

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(seed=0):
    from imageni2r_amd import TS2img_Karras
    torch.manual_seed(seed)
    m = TS2img_Karras(tiny_args(device=DEV), DEV)
    O.perturb_zero_init(m.net.state_dict(), seed=1234)
    return m.to(DEV).train()


def _batches(m, n, sizes=None, seed=11):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for it in range(n):
        B = sizes[it] if sizes else 4
        ts = torch.rand(B, 24, 3, generator=gen)
        ts[0, [2, 3], :] = float("nan")
        _, mask = m.ts_to_img_with_mask(ts.to(DEV))
        out.append((m.ts_to_img(torch.rand(B, 24, 3, generator=gen).to(DEV)), mask))
    return out


def test_graphed_step_with_reducer_world1_matches_plain_graph():
    """GraphedTrainStep(reducer=GradBucketReducer) splits the capture after backward (graph 1: forward + backward
    writing into the reducer's buckets; eager: launch_all / finish; graph 2: clip + AdamW + EMA).  At world size 1 the
    collective is the identity, so losses and weights must equal the single-graph step's."""
    from imageni2r_amd import FusedAdamW, GraphedTrainStep
    from imageni2r_amd.parallel import GradBucketReducer
    runs = []
    for with_reducer in (False, True):
        m = _model()
        red = GradBucketReducer(m.net.parameters(), bucket_bytes=1 << 20) if with_reducer else None
        opt = FusedAdamW(m.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
        step = GraphedTrainStep(m, opt, reducer=red)
        losses = [step(x, k).item() for x, k in _batches(m, 5)]
        if with_reducer:
            assert len(red.buckets) > 1
            g1, g2 = step._entries[(4, 3, 8, 8)][:2]
            assert g2 is not None                       # the two-graph split was taken
            for p in m.net.parameters():                # gradients live inside the reducer's buckets
                assert p.grad is not None and p.grad.data_ptr() == red._view(p).data_ptr()
            # the weighted path at world 1 with n_local == n_total is the identity as well
            red.set_local_count(4, n_total=4)
            losses.append(step(*_batches(m, 1, seed=12)[0]).item())
            red.set_local_count(None)
        else:
            losses.append(step(*_batches(m, 1, seed=12)[0]).item())
        runs.append((losses, [p.detach().cpu().clone() for p in m.net.parameters()],
                     int(m.model_ema.num_updates.item())))
    (l0, p0, n0), (l1, p1, n1) = runs
    assert n0 == n1 == 6
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-3 * abs(a), (l0, l1)   # fp32 atomics in wgrad / GroupNorm dgamma: not bit-equal
    for a, b in zip(p0, p1):
        e_max, e_rms = rel_err(b.numpy(), a.numpy()), rms_rel_err(b.numpy(), a.numpy())
        assert e_rms < 1e-3 and e_max < 2e-2, (e_max, e_rms)


def test_every_call_issues_one_round_of_bucket_reductions_capturing_or_not():
    """Ranks capture when THEY first meet an input shape (a short last batch on one rank only), so a capture must
    issue exactly what a replay issues: one start per bucket, in bucket order -- no collective from the warm-up steps
    or from resetting the arrival counters (those would pair with another rank's real step)."""
    from imageni2r_amd import FusedAdamW, GraphedTrainStep
    from imageni2r_amd.parallel import GradBucketReducer
    m = _model()
    red = GradBucketReducer(m.net.parameters(), bucket_bytes=1 << 20)
    opt = FusedAdamW(m.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
    step = GraphedTrainStep(m, opt, reducer=red)
    red.set_local_count(4, n_total=4)   # weighting on: at world size 1 a bucket "start" is then a real kernel launch
    started = []
    orig = red._start
    red._start = lambda b: (started.append(b), orig(b))[1]
    sizes = [4, 4, 2, 4, 2]             # captures at calls 0 (batch 4) and 2 (batch 2), replays otherwise
    for x, k in _batches(m, len(sizes), sizes):
        started.clear()
        assert np.isfinite(step(x, k).item())
        assert started == list(range(len(red.buckets))), (tuple(x.shape), started)
    assert len(step._entries) == 2
    assert int(m.model_ema.num_updates.item()) == len(sizes)


def test_one_update_per_batch_and_one_graph_per_shape():
    """The reference's loaders have no drop_last (utils/utils_data.py:243): the short last batch of an epoch and the
    first full batch after it must each cost ONE optimiser step, and each shape is captured once."""
    from imageni2r_amd import FusedAdamW, GraphedTrainStep
    m = _model()
    opt = FusedAdamW(m.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
    step = GraphedTrainStep(m, opt)
    sizes = [4, 4, 2, 4, 4, 2]
    for i, (x, k) in enumerate(_batches(m, len(sizes), sizes)):
        assert np.isfinite(step(x, k).item())
        assert int(step.rng_counter.item()) == i + 1
        assert int(m.model_ema.num_updates.item()) == i + 1
        assert float(opt.state[next(iter(m.net.parameters()))]["step"]) == i + 1
    assert sorted(step._entries) == [(2, 3, 8, 8), (4, 3, 8, 8)]
    # capture is side-effect free: the first step from a given state equals the eager step from the same state
    ma, mb = _model(), _model()
    x, k = _batches(ma, 1)[0]
    outs = []
    for mm, graph in ((ma, False), (mb, True)):
        o = FusedAdamW(mm.net.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0, ema=(mm.model_ema, mm.net))
        loss = GraphedTrainStep(mm, o, use_graph=graph)(x, k).item()
        outs.append((loss, [p.detach().cpu().clone() for p in mm.net.parameters()]))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[0][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        # one sign-like Adam step: where |g| ~ eps the last bits of the wgrad atomics flip single elements' updates
        e_max, e_rms = rel_err(b.numpy(), a.numpy()), rms_rel_err(b.numpy(), a.numpy())
        assert e_rms < 1e-3 and e_max < 5e-2, (e_max, e_rms)


def test_load_state_dict_rebinds_the_moments_and_recaptures():
    from imageni2r_amd import FusedAdamW, GraphedTrainStep
    m = _model()
    opt = FusedAdamW(m.net.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
    step = GraphedTrainStep(m, opt)
    bs = _batches(m, 4)
    step(*bs[0]); step(*bs[1])
    saved = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict()["state"][0].items()}
    import copy
    sd = copy.deepcopy(opt.state_dict())  # (state_dict() hands out the live tensors)
    step(*bs[2])
    gen0 = opt.state_generation
    opt.load_state_dict(sd)                      # back to the moments after two steps (new tensors)
    assert opt.state_generation == gen0 + 1
    p0 = next(iter(m.net.parameters()))
    assert torch.equal(opt.state[p0]["exp_avg"], saved["exp_avg"].to(DEV))
    loaded_ptr = opt.state[p0]["exp_avg"].data_ptr()
    step(*bs[3])                                 # must re-capture and update the LOADED moment tensors
    assert opt.state[p0]["exp_avg"].data_ptr() == loaded_ptr
    assert not torch.equal(opt.state[p0]["exp_avg"], saved["exp_avg"].to(DEV))
    assert float(opt.state[p0]["step"]) == 3.0
    assert float(opt.state_dict()["state"][0]["step"]) == 3.0


def test_gradient_accumulation_on_the_slot_path():
    """Two backwards without zero_grad: gradients add up (the second backward must not overwrite the slot and have
    autograd add the alias to itself)."""
    from imageni2r_amd import FusedAdamW
    m = _model().eval()
    opt = FusedAdamW(m.net.parameters(), lr=1e-4)
    opt.bind_grads()
    (x1, k1), (x2, k2) = _batches(m, 2)
    gen = torch.Generator().manual_seed(1)
    draws = [(torch.randn(4, generator=gen).to(DEV), torch.randn(4, 3, 8, 8, generator=gen).to(DEV)) for _ in range(2)]

    def backward(x, k, d):
        m.inject_randomness(*d)
        try:
            m.loss_fn_irregular(x, k)[0].backward()
        finally:
            m.inject_randomness(None, None)

    single = []
    for (x, k), d in (((x1, k1), draws[0]), ((x2, k2), draws[1])):
        opt.zero_grad()
        backward(x, k, d)
        single.append([p.grad.detach().clone() for p in m.net.parameters()])
    opt.zero_grad()
    backward(x1, k1, draws[0])
    backward(x2, k2, draws[1])
    for p, a, b in zip(m.net.parameters(), *single):
        want = (a + b).cpu().numpy()
        assert rel_err(p.grad.cpu().numpy(), want) < 1e-5
        assert p.grad.data_ptr() == p._grad_slot.data_ptr()


# ---------------------------------------------------------------------------------------------- Philox streams
def test_sigma_and_noise_draws_have_the_reference_distribution():
    """our.py:71-77: rnd ~ N(0,1), sigma = exp(1.2 rnd - 1.2); noise ~ N(0,1) * sigma on observed pixels only."""
    from imageni2r_amd import ops
    B, C, S = 4096, 4, 16
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(B, C, S, S, generator=gen).to(DEV)
    mask = (torch.rand(B, C, S, S, generator=gen) > 0.3).float().to(DEV)
    xn, sigma, weight = ops.edm_noise(x, mask, -1.2, 1.2, 0.5, seed=ops.rng_seed(), offset=7)
    ls = sigma.log().double().cpu().numpy()
    assert abs(ls.mean() + 1.2) < 4 * 1.2 / np.sqrt(B) and abs(ls.std() - 1.2) < 0.05
    # Kolmogorov-Smirnov against N(-1.2, 1.2)
    from scipy import stats
    assert stats.kstest((ls + 1.2) / 1.2, "norm").pvalue > 1e-3
    w = weight.double().cpu().numpy()
    s = sigma.double().cpu().numpy()
    assert np.allclose(w, (s * s + 0.25) / (s * 0.5) ** 2, rtol=1e-5)
    n = ((xn - x) / sigma.view(-1, 1, 1, 1)).double().cpu().numpy()
    mk = mask.cpu().numpy() > 0
    assert np.array_equal(xn.cpu().numpy()[~mk], x.cpu().numpy()[~mk])       # unobserved pixels: no noise at all
    z = n[mk]
    assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.var() - 1.0) < 0.01
    assert abs(stats.skew(z[:200000])) < 0.03 and abs(stats.kurtosis(z[:200000])) < 0.06
    assert stats.kstest(z[:100000], "norm").pvalue > 1e-3
    # neighbouring elements / samples are uncorrelated
    zz = n * mk
    assert abs(np.corrcoef(zz[:, :, :, :-1].ravel(), zz[:, :, :, 1:].ravel())[0, 1]) < 0.005
    assert abs(np.corrcoef(zz[:-1].ravel(), zz[1:].ravel())[0, 1]) < 0.005
    assert abs(np.corrcoef(ls[:-1], ls[1:])[0, 1]) < 0.06


def test_streams_differ_across_steps_offsets_and_ranks():
    from imageni2r_amd import ops
    B, C, S = 256, 4, 16
    x = torch.zeros(B, C, S, S, device=DEV)
    mask = torch.ones_like(x)

    def draw(seed, offset=0, step=None):
        xn, sigma, _ = ops.edm_noise(x, mask, -1.2, 1.2, 0.5, seed=seed, offset=offset, step=step)
        return (xn / sigma.view(-1, 1, 1, 1)).double().cpu().numpy().ravel(), sigma.log().double().cpu().numpy()

    base_n, base_s = draw(ops.rng_seed(), 3)
    again_n, again_s = draw(ops.rng_seed(), 3)
    assert np.array_equal(base_n, again_n) and np.array_equal(base_s, again_s)  # counter-based: reproducible
    others = [draw(ops.rng_seed(), 4)]                                          # next host offset
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    for _ in range(2):                                                          # device step counter 1, 2
        ops.counter_add(ctr, 1)
        others.append(draw(ops.rng_seed(), 0, ctr))
    try:                                                                        # another data-parallel rank
        ops.set_rng_rank(1)
        assert ops.rng_seed() != (torch.initial_seed() & ((1 << 63) - 1))
        others.append(draw(ops.rng_seed(), 3))
        ops.set_rng_rank(5)
        others.append(draw(ops.rng_seed(), 3))
    finally:
        ops.set_rng_rank(None)
    streams = [(base_n, base_s)] + others
    for i in range(len(streams)):
        for j in range(i + 1, len(streams)):
            assert abs(np.corrcoef(streams[i][0], streams[j][0])[0, 1]) < 0.01, (i, j)
            assert abs(np.corrcoef(streams[i][1], streams[j][1])[0, 1]) < 0.25, (i, j)
            assert not np.array_equal(streams[i][1], streams[j][1])


def test_dropout_keep_rate_per_layer_and_independence():
    """F.dropout(p=0.1) of networks.py:177 fused into the GroupNorm kernel: keep rate 0.9 per layer, kept values scaled
    by 1/0.9, masks independent across layers (stream ids), steps and ranks, and regenerated identically in backward."""
    from imageni2r_amd import ops
    B, C, H = 64, 128, 16
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(B, C, H, H, generator=gen).to(DEV)
    gamma, beta = torch.ones(C, device=DEV), torch.full((C,), 3.0, device=DEV)  # beta keeps y away from 0
    plain, _, _ = ops.groupnorm_fwd(x, gamma, beta, 32, 1e-5, None, False)
    ctr = torch.ones(1, dtype=torch.int64, device=DEV)
    masks = {}
    for layer in (1, 2, 37):
        for step in (1, 2):
            ctr.fill_(step)
            y, _, _ = ops.groupnorm_fwd(x, gamma, beta, 32, 1e-5, None, False, False, 0.1, ops.rng_seed(), layer, ctr)
            keep = (y != 0)
            rate = keep.float().mean().item()
            assert abs(rate - 0.9) < 4 * np.sqrt(0.09 / x.numel()) + 1e-4, (layer, step, rate)
            assert torch.allclose(y[keep], plain[keep] / 0.9, rtol=1e-6, atol=1e-6)
            masks[(layer, step, 0)] = keep.cpu().numpy().ravel()
    try:
        ops.set_rng_rank(3)
        ctr.fill_(1)
        y, _, _ = ops.groupnorm_fwd(x, gamma, beta, 32, 1e-5, None, False, False, 0.1, ops.rng_seed(), 1, ctr)
        masks[(1, 1, 3)] = (y != 0).cpu().numpy().ravel()
    finally:
        ops.set_rng_rank(None)
    keys = list(masks)
    for i in range(len(keys)):
        for j in range(i + 1, len(keys)):
            c = np.corrcoef(masks[keys[i]].astype(np.float64), masks[keys[j]].astype(np.float64))[0, 1]
            assert abs(c) < 0.01, (keys[i], keys[j], c)
    # per-channel keep rates: no channel is systematically favoured
    ch = masks[(1, 1, 0)].reshape(B, C, H * H).mean(axis=(0, 2))
    assert np.abs(ch - 0.9).max() < 5 * np.sqrt(0.09 / (B * H * H))


# ---------------------------------------------------------------------------------------------- deterministic mode
@pytest.fixture()
def deterministic():
    from imageni2r_amd import ops
    ops.set_deterministic(True)
    yield
    ops.set_deterministic(False)


@pytest.mark.parametrize("B,Cin,Cout,H,k", [(4, 64, 64, 8, 3), (2, 384, 128, 16, 1), (32, 128, 128, 16, 3),
                                             (3, 12, 128, 16, 3), (5, 256, 12, 4, 3), (8, 512, 1024, 2, 1)])
def test_deterministic_wgrad_matches_the_atomic_one_and_repeats_bit_for_bit(deterministic, B, Cin, Cout, H, k):
    from imageni2r_amd import ops
    gen = torch.Generator().manual_seed(B * Cin + Cout)
    x = torch.randn(B, Cin, H, H, generator=gen).to(DEV)
    dy = torch.randn(B, Cout, H, H, generator=gen).to(DEV)
    dw1, db1 = ops.conv2d_wgrad(x, dy, k)
    dw2, db2 = ops.conv2d_wgrad(x, dy, k)
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)                       # reproducible
    ops.set_deterministic(False)
    dwa, dba = ops.conv2d_wgrad(x, dy, k)
    ops.set_deterministic(True)
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (Cout, Cin, k, k), dy.double().cpu(), padding=k // 2)
    assert rel_err(dw1.cpu().numpy(), ref.numpy()) < 1e-5
    assert rel_err(dw1.cpu().numpy(), dwa.cpu().numpy()) < 1e-5
    assert rel_err(db1.cpu().numpy(), dy.double().sum(dim=(0, 2, 3)).cpu().numpy()) < 1e-5
    # accumulate into pre-filled slots, without a bias
    slot = torch.full((Cout, Cin, k, k), 0.5, device=DEV)
    ops.conv2d_wgrad(x, dy, k, want_bias=False, dw=slot, accumulate=True)
    assert rel_err((slot - 0.5).cpu().numpy(), ref.numpy()) < 1e-5


def test_deterministic_mode_makes_the_replayed_step_bit_identical_to_the_eager_one(deterministic):
    """No atomics anywhere in the step => the graph replay and the eager sequence of the same kernels are the same
    arithmetic in the same order: losses, weights, EMA shadows and Adam moments must be EQUAL, not close."""
    from imageni2r_amd import FusedAdamW, GraphedTrainStep
    runs = []
    for use_graph in (False, True):
        m = _model()
        opt = FusedAdamW(m.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
        step = GraphedTrainStep(m, opt, use_graph=use_graph)
        losses = [step(x, k).item() for x, k in _batches(m, 6)]
        p0 = next(iter(m.net.parameters()))
        runs.append((losses, [p.detach().clone() for p in m.net.parameters()],
                     [b.detach().clone() for b in m.model_ema.buffers()], opt.state[p0]["exp_avg_sq"].clone()))
    (le, pe, ee, ve), (lg, pg, eg, vg) = runs
    assert le == lg, (le, lg)
    assert all(torch.equal(a, b) for a, b in zip(pe, pg))
    assert all(torch.equal(a, b) for a, b in zip(ee, eg))
    assert torch.equal(ve, vg)
    # and the mode is reproducible run to run (a fresh model, the same six batches)
    m = _model()
    opt = FusedAdamW(m.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
    step = GraphedTrainStep(m, opt)
    assert [step(x, k).item() for x, k in _batches(m, 6)] == lg
