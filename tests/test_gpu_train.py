"""GPU tests (-m gpu) of the training-side additions: register-resident GroupNorm over every group geometry, the
split-f16 inference conv on its own, the fused optimiser tail against torch's AdamW + clip_grad_norm_ + the
reference's EMA rule, and the graph-captured train step against the same step run eagerly."""
import copy

import numpy as np
import pytest
import torch

from tests.helpers import assert_close, rel_err, tiny_args
from oracle import i2r_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda"


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from imageni2r_amd import ops as _ops
    return _ops


# (B, C, H): group sizes 16 .. 4096 elements, HW = 1 (generic kernel), 4, 16, 64, 256, 1024; ragged group counts
@pytest.mark.parametrize("B,C,H", [(3, 128, 16), (2, 128, 32), (2, 256, 32), (5, 1024, 2), (3, 512, 2), (7, 256, 4),
                                   (2, 384, 8), (3, 256, 1), (1, 768, 4), (2, 1024, 1), (9, 128, 8), (2, 12 * 4, 16),
                                   (2, 384, 32), (1, 512, 32)])
def test_groupnorm_register_kernels_all_geometries(ops, B, C, H):
    gen = torch.Generator().manual_seed(B * 100 + C + H)
    x = torch.randn(B, C, H, H, generator=gen) * 1.7 + 0.3
    w, b = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    ss = (torch.randn(B, 2 * C, generator=gen) * 0.5).requires_grad_(True)
    dy = torch.randn(B, C, H, H, generator=gen)
    G = min(32, C // 4)
    for use_ss, act in ((True, True), (False, True), (False, False)):
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ssr = ss.detach().clone().requires_grad_(True)
        ref = torch.nn.functional.group_norm(xr, G, wr, br, 1e-5)
        if use_ss:
            ref = ssr[:, C:, None, None] + ref * (ssr[:, :C, None, None] + 1)
        if act:
            ref = torch.nn.functional.silu(ref)
        ref.backward(dy)
        ssd = cu(ss.detach().numpy()) if use_ss else None
        y, mean, rstd = ops.groupnorm_fwd(cu(x.numpy()), cu(w.numpy()), cu(b.numpy()), G, 1e-5, ss=ssd, act=act,
                                          save_stats=True)
        assert_close(y.cpu().numpy(), ref.detach().numpy(), TOL)
        dx, dw, db, dss = ops.groupnorm_bwd(cu(x.numpy()), cu(w.numpy()), cu(b.numpy()), ssd, mean, rstd,
                                            cu(dy.numpy()), G, act)
        assert_close(dx.cpu().numpy(), xr.grad.numpy(), TOL)
        assert_close(dw.cpu().numpy(), wr.grad.numpy(), TOL)
        assert_close(db.cpu().numpy(), br.grad.numpy(), TOL)
        if use_ss:
            assert_close(dss.cpu().numpy(), ssr.grad.numpy(), TOL)


def test_groupnorm_dropout_device_step(ops):
    """Device-resident RNG step: same (seed, id, step) -> same mask; bumping the counter changes it; backward sees
    the forward's mask."""
    torch.manual_seed(5)
    B, C, H, p = 4, 128, 16, 0.1
    x = torch.randn(B, C, H, H)
    xd, wd, bd = x.to(DEV), torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    y0, mean, rstd = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, save_stats=True)
    ya, _, _ = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, drop_p=p, seed=9, offset=3, step=ctr)
    yb, _, _ = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, drop_p=p, seed=9, offset=3, step=ctr)
    assert torch.equal(ya, yb)
    keep = ya != 0
    dy = torch.ones_like(xd)
    dxa, *_ = ops.groupnorm_bwd(xd, wd, bd, None, mean, rstd, dy * keep.float() / (1 - p), 32, False)
    dxb, *_ = ops.groupnorm_bwd(xd, wd, bd, None, mean, rstd, dy, 32, False, drop_p=p, seed=9, offset=3, step=ctr)
    assert_close(dxb.cpu().numpy(), dxa.cpu().numpy(), 1e-5)
    ops.counter_add(ctr, 1)
    assert int(ctr.item()) == 1
    yc, _, _ = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, drop_p=p, seed=9, offset=3, step=ctr)
    assert not torch.equal(ya, yc)
    assert abs((yc != 0).float().mean().item() - (1 - p)) < 0.01
    assert torch.allclose(yc[yc != 0], y0[yc != 0] / (1 - p), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,Cin,Cout,H,k", [(4, 128, 128, 16, 3), (2, 384, 256, 8, 3), (3, 1024, 512, 2, 3),
                                            (2, 12, 128, 16, 3), (2, 128, 12, 16, 3), (2, 256, 768, 8, 1),
                                            (5, 321, 128, 16, 3), (2, 128, 321, 16, 3), (3, 256, 256, 1, 3),
                                            (2, 768, 256, 4, 1), (33, 64, 96, 4, 3), (1, 7, 128, 32, 3),
                                            (2, 13, 40, 4, 1), (3, 128, 384, 16, 1), (3, 128, 256, 5, 1),
                                            (5, 256, 512, 3, 1)])
def test_conv_split_f16_vs_oracle(ops, B, Cin, Cout, H, k):
    """The inference conv on the f16 matrix cores (x = hi + lo split, 3 MFMAs) holds the fp32 bar on every tile edge."""
    gen = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=gen) * 3.0
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, H, generator=gen)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=k // 2) + res.double()
    wd = cu(w.numpy())
    y = ops.conv2d_fwd_h2(cu(x.numpy()), tuple(w.shape), ops.conv2d_pack_h2(wd), cu(b.numpy()), cu(res.numpy()))
    assert_close(y.cpu().numpy(), ref.numpy(), 2e-6)


@pytest.mark.parametrize("B,Cin,Cout,H,k", [(3, 128, 128, 16, 3), (2, 256, 384, 8, 1), (5, 96, 160, 4, 3),
                                            (2, 1024, 512, 2, 3), (2, 128, 12, 16, 3), (3, 128, 384, 16, 1),
                                            (3, 128, 256, 5, 1)])
def test_groupnorm_split_out_feeds_split_conv(ops, B, Cin, Cout, H, k):
    """GroupNorm(+SiLU+modulation) emitting f16 hi/lo pairs, consumed by the f16x2 conv with x_split: the pair of
    kernels matches the fp64 composition to fp32-grade accuracy (the conv does no conversion of its own)."""
    gen = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=gen) * 2.0 + 0.5
    gw, gb = torch.randn(Cin, generator=gen), torch.randn(Cin, generator=gen)
    ss = torch.randn(1, 2 * Cin, generator=gen) * 0.5
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=gen)
    G = min(32, Cin // 4)
    n = torch.nn.functional.group_norm(x.double(), G, gw.double(), gb.double(), 1e-5)
    n = torch.nn.functional.silu(ss[:, Cin:, None, None].double() + n * (ss[:, :Cin, None, None].double() + 1))
    ref = torch.nn.functional.conv2d(n, w.double(), b.double(), padding=k // 2)
    ysp, _, _ = ops.groupnorm_fwd(cu(x.numpy()), cu(gw.numpy()), cu(gb.numpy()), G, 1e-5, ss=cu(ss.numpy()), act=True,
                                  split_out=True)
    y = ops.conv2d_fwd_h2(ysp, tuple(w.shape), ops.conv2d_pack_h2(cu(w.numpy())), cu(b.numpy()), None, x_split=True)
    assert_close(y.cpu().numpy(), ref.numpy(), 5e-6)


@pytest.mark.parametrize("B,Cin,Cout,H", [(3, 128, 128, 16), (5, 96, 160, 4), (2, 1024, 512, 2), (2, 128, 12, 16),
                                          (260, 128, 256, 16), (2, 64, 128, 32)])
def test_conv_f16_single_term_path(ops, B, Cin, Cout, H):
    """set_conv_math("f16"), the optional reduced-precision path: the 3x3 conv on pre-split activations multiplies the
    f16 high halves only.  It computes exactly that product (checked against an fp64 conv of the rounded operands) and
    stays far inside the 1e-2 bar against the unrounded one; the default mode is untouched."""
    import imageni2r_amd
    gen = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=gen) * 2.0 + 0.5
    gw, gb = torch.randn(Cin, generator=gen), torch.randn(Cin, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, H, generator=gen)
    G = min(32, Cin // 4)
    n = torch.nn.functional.silu(torch.nn.functional.group_norm(x.double(), G, gw.double(), gb.double(), 1e-5))
    ref = torch.nn.functional.conv2d(n, w.double(), b.double(), padding=1) + res.double()
    ysp, _, _ = ops.groupnorm_fwd(cu(x.numpy()), cu(gw.numpy()), cu(gb.numpy()), G, 1e-5, act=True, split_out=True)
    wp = ops.conv2d_pack_h2(cu(w.numpy()))
    full = ops.conv2d_fwd_h2(ysp, tuple(w.shape), wp, cu(b.numpy()), cu(res.numpy()), x_split=True)
    imageni2r_amd.set_conv_math("f16")
    try:
        y = ops.conv2d_fwd_h2(ysp, tuple(w.shape), wp, cu(b.numpy()), cu(res.numpy()), x_split=True)
    finally:
        imageni2r_amd.set_conv_math("f16x2")
    assert_close(full.cpu().numpy(), ref.numpy(), 5e-6)
    e = rel_err(y.cpu().numpy(), ref.numpy())
    assert 1e-6 < e < 2e-3, e  # really the reduced product, and well inside 1e-2
    # the rounded operands: hi halves as the GroupNorm kernel stored them, weights as packed (scaled by 64, see the kernel)
    x_hi = (ysp.view(torch.int32) & 0xFFFF).to(torch.int16).view(torch.float16).double().cpu()
    w_hi = (w * 64.0).half().double() / 64.0
    ref_hi = torch.nn.functional.conv2d(x_hi, w_hi, b.double(), padding=1) + res.double()
    assert_close(y.cpu().numpy(), ref_hi.numpy(), 2e-6)


@pytest.mark.parametrize("B,heads,H", [(2, 2, 16), (1, 4, 32), (3, 4, 8), (2, 8, 8), (1, 2, 8)])
def test_attention_split_f16_vs_oracle(ops, B, heads, H):
    """Inference attention on the f16 matrix cores (T = 64, 256, 1024; several key tiles, base-2 online softmax)."""
    import ctypes
    from imageni2r_amd import _lib
    gen = torch.Generator().manual_seed(B + heads + H)
    C = heads * 64
    qkv = torch.randn(B, 3 * C, H, H, generator=gen) * 1.5
    qkv.view(B, heads * 64, 3, H * H)[:, :, 1, (H * H) // 2:] *= 3.0  # late key tiles dominate: running-max rescale
    ref = O.attention(qkv.double(), heads)
    q = cu(qkv.numpy())
    out = torch.empty(B, C, H, H, device=DEV)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.call("i2r_attention_fwd_h2", vp(q), vp(out), B, heads, H * H, 0,
              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert_close(out.cpu().numpy(), ref.numpy(), 5e-6)
    a, _ = ops.attention_fwd(q, heads)  # the dispatcher picks the same kernel in f16x2 mode
    assert torch.equal(a, out)


@pytest.mark.parametrize("B,C,H", [(3, 128, 16), (2, 256, 8), (1, 128, 32)])
def test_qkv_conv_split_output_feeds_presplit_attention(ops, B, C, H):
    """The qkv 1x1 conv storing split pairs (q rows pre-scaled) + the attention kernel that only permutes them:
    the pair matches the fp64 composition, and the separate (fp32 qkv) route, to fp32-grade accuracy."""
    heads = C // 64
    gen = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, H, generator=gen)
    w = torch.randn(3 * C, C, 1, 1, generator=gen) / np.sqrt(C) * 1.5
    b = torch.randn(3 * C, generator=gen) * 0.5
    qkv_ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double())
    ref = O.attention(qkv_ref, heads)
    wp = ops.conv2d_pack_h2(cu(w.numpy()))
    xd, bd = cu(x.numpy()), cu(b.numpy())
    qkv_sp = ops.conv2d_fwd_h2(xd, tuple(w.shape), wp, bd, None, qkv_out=True)
    a_sp, _ = ops.attention_fwd(qkv_sp, heads, presplit=True)
    assert_close(a_sp.cpu().numpy(), ref.numpy(), 5e-6)
    a_plain, _ = ops.attention_fwd(ops.conv2d_fwd_h2(xd, tuple(w.shape), wp, bd, None), heads)
    assert_close(a_sp.cpu().numpy(), a_plain.cpu().numpy(), 5e-6)


@pytest.mark.parametrize("B,C1,C2,H,Cout", [(3, 256, 128, 8, 256), (2, 512, 512, 2, 512), (2, 256, 256, 4, 256),
                                            (2, 256, 128, 32, 128), (1, 128, 128, 32, 128),
                                            (5, 128, 128, 16, 128), (2, 64, 32, 8, 64)])
def test_concat_free_groupnorm_and_skip_conv(ops, B, C1, C2, H, Cout):
    """Decoder skip connections without materialising torch.cat (networks.py:458): GroupNorm and the 1x1 skip conv
    read [x | skip] in place; groups that straddle the two tensors included (C1 not a multiple of the group width)."""
    gen = torch.Generator().manual_seed(C1 + C2 + H)
    x1 = torch.randn(B, C1, H, H, generator=gen) * 1.3
    x2 = torch.randn(B, C2, H, H, generator=gen) * 0.7 + 0.4
    C = C1 + C2
    gw, gb = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    w = torch.randn(Cout, C, 1, 1, generator=gen) / np.sqrt(C)
    bias = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, H, generator=gen)
    cat = torch.cat([x1, x2], 1).double()
    G = min(32, C // 4)
    ref_n = torch.nn.functional.silu(torch.nn.functional.group_norm(cat, G, gw.double(), gb.double(), 1e-5))
    ref_c = torch.nn.functional.conv2d(cat, w.double(), bias.double()) + res.double()
    a, b = cu(x1.numpy()), cu(x2.numpy())
    y = ops.groupnorm_cat_fwd(a, b, cu(gw.numpy()), cu(gb.numpy()), G, 1e-5, act=True)
    assert_close(y.cpu().numpy(), ref_n.numpy(), 5e-6)
    ysp = ops.groupnorm_cat_fwd(a, b, cu(gw.numpy()), cu(gb.numpy()), G, 1e-5, act=True, split_out=True)
    y1, _, _ = ops.groupnorm_fwd(cu(cat.float().numpy()), cu(gw.numpy()), cu(gb.numpy()), G, 1e-5, act=True,
                                 split_out=True)
    assert torch.equal(ysp.view(torch.int32), y1.view(torch.int32))  # same bits as the single-tensor kernel
    wd = cu(w.numpy())
    c = ops.conv1x1_cat_fwd_h2(a, b, tuple(w.shape), ops.conv2d_pack_h2(wd), cu(bias.numpy()), cu(res.numpy()))
    assert_close(c.cpu().numpy(), ref_c.numpy(), 2e-6)


def _ref_tail(params, grads, state, shadows, n_upd, lr, wd, max_norm, warmup, steps):
    """torch AdamW + clip_grad_norm_ + the EMA rule of models/ema.py:26-47 on the CPU."""
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.AdamW(ps, lr=lr, weight_decay=wd)
    sh = [s.clone() for s in shadows]
    for t in range(steps):
        for p, g in zip(ps, grads[t]):
            p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ps, max_norm)
        opt.step()
        n_upd += 1
        decay = min(np.float32(0.9999), np.float32(1 + n_upd) / np.float32(10 + n_upd))
        omd = float(np.float32(1.0) - decay)
        with torch.no_grad():
            for s, p in zip(sh, ps):
                if n_upd > warmup:
                    s.sub_(omd * (s - p))
                else:
                    s.copy_(p)
    return [p.detach() for p in ps], sh


def test_fused_adamw_clip_ema_vs_torch():
    from imageni2r_amd import FusedAdamW, TS2img_Karras
    torch.manual_seed(0)
    args = tiny_args(device=DEV, ema_warmup=2)
    m = TS2img_Karras(args, DEV)
    O.perturb_zero_init(m.net.state_dict(), seed=1234)
    m = m.to(DEV)
    names = [k for k, _ in m.net.named_parameters()]
    params0 = [p.detach().cpu().clone() for p in m.net.parameters()]
    shadows0 = [dict(m.model_ema.named_buffers())[m.model_ema.m_name2s_name[k]].detach().cpu().clone() for k in names]
    gen = torch.Generator().manual_seed(3)
    steps = 5
    # gradients large enough that the clip is active on some steps and inactive on others
    grads = [[torch.randn(p.shape, generator=gen) * (0.02 if t % 2 else 1e-4) for p in params0] for t in range(steps)]
    opt = FusedAdamW(m.net.parameters(), lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0, ema=(m.model_ema, m.net))
    for t in range(steps):
        opt.zero_grad()
        for p, g in zip(m.net.parameters(), grads[t]):
            if p.grad is None:
                p.grad = g.to(DEV)
            else:
                p.grad.copy_(g)
        opt.step()
        m.on_train_batch_end()  # must be a no-op now
        ref_norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads[t])).item()
        assert abs(opt.grad_norm().item() - ref_norm) <= 1e-5 * ref_norm
    ref_p, ref_s = _ref_tail(params0, grads, None, shadows0, 0, 1e-3, 1e-2, 1.0, 2, steps)
    assert int(m.model_ema.num_updates.item()) == steps
    bufs = dict(m.model_ema.named_buffers())
    for k, p, rp, rs in zip(names, m.net.parameters(), ref_p, ref_s):
        assert_close(p.detach().cpu().numpy(), rp.numpy(), 2e-6, k)
        assert_close(bufs[m.model_ema.m_name2s_name[k]].cpu().numpy(), rs.numpy(), 2e-6, k)
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == steps


def test_graphed_train_step_matches_eager():
    """Same seeds, same device RNG counter: the replayed graph and the eager sequence of the same kernels give the
    same losses and weights (up to the fp32 atomics of split wgrad), and dropout/noise really change per replay."""
    from imageni2r_amd import FusedAdamW, GraphedTrainStep, TS2img_Karras
    runs = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        m = TS2img_Karras(tiny_args(device=DEV), DEV)
        O.perturb_zero_init(m.net.state_dict(), seed=1234)
        m = m.to(DEV).train()
        opt = FusedAdamW(m.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
        step = GraphedTrainStep(m, opt, use_graph=use_graph)
        gen = torch.Generator().manual_seed(11)
        losses = []
        for it in range(6):
            ts = torch.rand(4, 24, 3, generator=gen)
            ts[0, [2, 3], :] = float("nan")
            _, mask = m.ts_to_img_with_mask(ts.to(DEV))
            x_tilde = m.ts_to_img(torch.rand(4, 24, 3, generator=gen).to(DEV))
            losses.append(step(x_tilde, mask).item())
        assert all(np.isfinite(losses))
        runs.append((losses, [p.detach().cpu().clone() for p in m.net.parameters()], int(step.rng_counter.item()),
                     int(m.model_ema.num_updates.item())))
    (le, pe, ce, ne), (lg, pg, cg, ng) = runs
    # the capture's warm-up steps are undone and stream capture records without executing: every batch got exactly
    # one update in both modes (run_irregular.py:220-230 has one optimiser step per batch)
    assert ce == cg == 6 and ne == ng == 6
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-3 * abs(a), (le, lg)
    for a, b in zip(pe, pg):  # Adam's sign-like first steps amplify the last-bit noise of the atomics in wgrad
        assert_close(b.numpy(), a.numpy(), 5e-3)
    assert len(set(np.round(lg, 6))) > 1  # noise / sigma draws differ between replays


def test_pack_plan_skips_the_fp32_layouts_f16x2_layers_do_not_read(monkeypatch):
    """pack_all_weights packs, per step, only the fp32 layouts some kernel reads: layers whose forward (and data
    gradient) ran on the f16x2 kernels drop out of the plan after the first step.  Same weights, bit for bit, as with
    every layout packed (I2R_PACK_ALL=1) -- deterministic mode, replayed graph, batch big enough that the tiny net's
    8x8 level crosses the f16x2 thresholds."""
    from imageni2r_amd import FusedAdamW, GraphedTrainStep, TS2img_Karras, ops as iops
    from imageni2r_amd.networks import Conv2d
    iops.set_deterministic(True)
    runs = []
    try:
        for pack_all in ("1", None):
            if pack_all:
                monkeypatch.setenv("I2R_PACK_ALL", pack_all)
            else:
                monkeypatch.delenv("I2R_PACK_ALL", raising=False)
            torch.manual_seed(0)
            m = TS2img_Karras(tiny_args(device=DEV, batch_size=512), DEV)
            O.perturb_zero_init(m.net.state_dict(), seed=1234)
            m = m.to(DEV).train()
            opt = FusedAdamW(m.net.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0,
                             ema=(m.model_ema, m.net))
            step = GraphedTrainStep(m, opt)
            gen = torch.Generator().manual_seed(11)
            for it in range(3):
                _, mask = m.ts_to_img_with_mask(torch.rand(512, 24, 3, generator=gen).to(DEV))
                loss = step(m.ts_to_img(torch.rand(512, 24, 3, generator=gen).to(DEV)), mask)
            assert np.isfinite(loss.item())
            net = m.net.model
            convs = [c for c in net.modules() if isinstance(c, Conv2d) and c.weight is not None]
            jobs = net._pack_plan(dgrad=True)[1]
            runs.append((len(jobs), len(convs), [p.detach().cpu().clone() for p in m.net.parameters()]))
    finally:
        iops.set_deterministic(False)
    (n_all, n_convs, p_all), (n_few, _, p_few) = runs
    assert n_all == 2 * n_convs and n_few < n_all, (n_all, n_few, n_convs)
    for a, b in zip(p_all, p_few):
        assert torch.equal(a, b)


def test_ema_scope_between_graphed_steps(monkeypatch):
    """Evaluation inside the pointer-swapped ``ema_scope`` between replayed training steps: it samples with the shadow
    weights the fused optimiser tail has been updating, gives exactly what the reference's copy sequence gives, and
    leaves every address the training graph and the optimiser's pointer table hold as it was."""
    from imageni2r_amd import DiffusionProcess, FusedAdamW, GraphedTrainStep, TS2img_Karras
    torch.manual_seed(0)
    args = tiny_args(device=DEV)
    m = TS2img_Karras(args, DEV)
    O.perturb_zero_init(m.net.state_dict(), seed=1234)
    m = m.to(DEV).train()
    opt = FusedAdamW(m.net.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
    step = GraphedTrainStep(m, opt)
    gen = torch.Generator().manual_seed(5)

    def batch():
        ts = torch.rand(4, 24, 3, generator=gen)
        ts[1, [5, 6], :] = float("nan")
        _, mask = m.ts_to_img_with_mask(ts.to(DEV))
        return m.ts_to_img(torch.rand(4, 24, 3, generator=gen).to(DEV)), mask

    for _ in range(3):
        step(*batch())
    key = opt._table_key
    ptrs = [p.data_ptr() for p in m.net.parameters()]
    xT = torch.randn(2, 3, 8, 8, generator=gen)
    m.eval()
    with m.ema_scope():
        out_swap = DiffusionProcess(args, m.net, (3, 8, 8)).sample(xT)
    monkeypatch.setenv("I2R_EMA_SCOPE", "copy")
    with m.ema_scope():
        out_copy = DiffusionProcess(args, m.net, (3, 8, 8), use_graph=False).sample(xT)
    monkeypatch.delenv("I2R_EMA_SCOPE")
    out_live = DiffusionProcess(args, m.net, (3, 8, 8), use_graph=False).sample(xT)
    assert torch.equal(out_swap, out_copy) and not torch.equal(out_swap, out_live)
    m.train()
    assert ptrs == [p.data_ptr() for p in m.net.parameters()]
    before = [p.detach().clone() for p in m.net.parameters()]
    losses = [step(*batch()).item() for _ in range(3)]
    assert all(np.isfinite(losses)) and opt._table_key == key
    assert int(m.model_ema.num_updates.item()) == 6  # one per batch: the capture's warm-up steps leave no trace
    assert any(not torch.equal(a, b) for a, b in zip(before, m.net.parameters()))  # the graph still trains THESE weights
    m.eval()
    with m.ema_scope():
        out2 = DiffusionProcess(args, m.net, (3, 8, 8)).sample(xT)  # replays the first evaluation's graph
    assert len(m.net._i2r_sampler_graphs) == 1 and not torch.equal(out2, out_swap)


def test_slot_path_gradients_vs_oracle():
    """The training step's in-place gradient path -- gradient slots, grouped `affine` products (one launch for all
    blocks, GroupNorm backwards writing d[scale|shift] into its slab), atomically accumulated GroupNorm dgamma/dbeta --
    gives every parameter gradient of the golden loss within the path's 1e-4 bar of the oracle's autograd."""
    from imageni2r_amd import FusedAdamW, TS2img_Karras, ops as _ops
    from tests.helpers import load_golden
    g = load_golden("g4_net.npz")
    a = tiny_args()
    torch.manual_seed(0)
    m = TS2img_Karras(tiny_args(device=DEV), DEV)
    O.perturb_zero_init(m.net.state_dict(), seed=1234)
    m = m.to(DEV).eval()
    spec = O.unet_spec(a.img_resolution, a.input_channels, a.unet_channels, a.ch_mult, a.attn_resolution)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.net.state_dict().items()
          if not k.endswith("resample_filter")}
    ref = O.masked_edm_loss(sd, spec, torch.from_numpy(g["loss_x_tilde"]), torch.from_numpy(g["loss_mask"]),
                            torch.from_numpy(g["loss_rnd"]), torch.from_numpy(g["loss_noise"]))
    ref.backward()
    opt = FusedAdamW(m.net.parameters(), lr=1e-4, max_grad_norm=1.0, ema=(m.model_ema, m.net))
    opt.bind_grads()
    for rep in range(2):  # twice: the second pass must not see anything left over from the first
        opt.zero_grad()
        assert m.net.model.affine_all(torch.zeros(2, 4 * a.unet_channels, device=DEV, requires_grad=True)) is not None
        _ops.set_slots_zeroed(True)
        m.inject_randomness(cu(g["loss_rnd"]), cu(g["loss_noise"]))
        try:
            loss, _ = m.loss_fn_irregular(cu(g["loss_x_tilde"]), cu(g["loss_mask"]))
            loss.backward()
        finally:
            m.inject_randomness(None, None)
            _ops.set_slots_zeroed(False)
        assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item())
        for k, p in m.net.named_parameters():
            assert p.grad is not None and p.grad.data_ptr() == p._grad_slot.data_ptr(), k
            assert_close(p.grad.cpu().numpy(), sd[k].grad.numpy(), 1e-4, (rep, k))


@pytest.mark.parametrize("B,Cin,Cout,Hl,split", [(3, 128, 128, 8, True), (2, 256, 256, 4, False), (2, 64, 96, 16, True),
                                                  (5, 512, 512, 2, True), (2, 96, 64, 1, False)])
def test_conv_fused_upsample_input_and_residual(ops, B, Cin, Cout, Hl, split):
    """The decoder's up blocks in f16x2 inference: the nearest x2 upsample of conv0's input and of conv1's residual
    happens in the conv's addressing; results equal the materialised-upsample composition."""
    gen = torch.Generator().manual_seed(Cin + Cout + Hl)
    x = torch.randn(B, Cin, Hl, Hl, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, Hl, Hl, generator=gen)
    up = lambda t: t.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    wp = ops.conv2d_pack_h2(cu(w.numpy()))
    if split:  # GroupNorm hands over split pairs: upsampling them is the same pure copy
        gw, gb = torch.randn(Cin, generator=gen), torch.randn(Cin, generator=gen)
        G = min(32, Cin // 4)
        xin = torch.nn.functional.silu(torch.nn.functional.group_norm(x.double(), G, gw.double(), gb.double(), 1e-5))
        xd, _, _ = ops.groupnorm_fwd(cu(x.numpy()), cu(gw.numpy()), cu(gb.numpy()), G, 1e-5, None, True, split_out=True)
    else:
        xin, xd = x.double(), cu(x.numpy())
    ref = torch.nn.functional.conv2d(up(xin), w.double(), b.double(), padding=1) + up(res.double())
    y = ops.conv2d_fwd_h2(xd, tuple(w.shape), wp, cu(b.numpy()), cu(res.numpy()), x_split=split, up_in=True, up_res=True)
    assert tuple(y.shape) == (B, Cout, 2 * Hl, 2 * Hl)
    assert_close(y.cpu().numpy(), ref.numpy(), 5e-6)
    # residual-only upsampling on a plain conv (conv1 of the up block)
    x2 = torch.randn(B, Cout, 2 * Hl, 2 * Hl, generator=gen)
    w2 = torch.randn(Cout, Cout, 3, 3, generator=gen) / np.sqrt(Cout * 9)
    ref2 = torch.nn.functional.conv2d(x2.double(), w2.double(), b.double(), padding=1) + up(res.double())
    y2 = ops.conv2d_fwd_h2(cu(x2.numpy()), tuple(w2.shape), ops.conv2d_pack_h2(cu(w2.numpy())), cu(b.numpy()),
                           cu(res.numpy()), up_res=True)
    assert_close(y2.cpu().numpy(), ref2.numpy(), 5e-6)
