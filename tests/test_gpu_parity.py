"""GPU parity tests (-m gpu): every HIP entry point, called through the C ABI (ctypes), against the golden vectors
captured from the reference and against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): delay embedding and masks bit-exact; fp32 outputs and gradients <= 1e-4 relative
(max |a-b| / max |b|)."""
import json

import numpy as np
import pytest
import torch

from tests.helpers import (assert_close, as_f32, as_f64, golden_meta, inverse_close, load_golden, probe_vec, rel_err,
                           tiny_args)
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


# ------------------------------------------------------------------------------------------------ delay embedding
def test_delay_embedding_bit_exact(ops):
    from imageni2r_amd import DelayEmbedder
    g = load_golden("g1_delay.npz")
    for m in golden_meta(g):
        t = m["tag"]
        emb = DelayEmbedder(DEV, m["L"], m["d"], m["E"], m["B"], m["C"])
        ts = cu(as_f32(g[t + "_ts"]))
        img = emb.ts_to_img(ts)
        assert np.array_equal(img.cpu().numpy().view(np.uint32), g[t + "_img"]), t
        img2, mask = emb.ts_to_img_with_mask(ts)
        assert np.array_equal(img2.cpu().numpy().view(np.uint32), g[t + "_img"]), t
        assert np.array_equal(mask.cpu().numpy(), g[t + "_mask"]), t
        assert tuple(emb.img_shape) == tuple(g[t + "_img_shape"])
        for name, conv in (("f32", as_f32), ("f64", as_f64)):
            y = emb.img_to_ts(cu(conv(g[t + f"_inv_in_{name}"])))
            assert y.dtype == torch.float32
            assert np.array_equal(y.cpu().numpy().view(np.uint32), g[t + f"_inv_out_{name}"]), (t, name)


def test_delay_embedding_random_geometries_bit_exact(ops):
    """The HIP transforms on the reference's vectors for 48 random geometries (tests/golden/g10_delay_fuzz.npz)."""
    from imageni2r_amd import DelayEmbedder
    g = load_golden("g10_delay_fuzz.npz")
    for m in golden_meta(g):
        t = m["tag"]
        emb = DelayEmbedder(DEV, m["L"], m["d"], m["E"], m["B"], m["C"])
        img, mask = emb.ts_to_img_with_mask(cu(as_f32(g[t + "_ts"])))
        ref = as_f32(g[t + "_img"])
        assert np.array_equal(img.cpu().numpy().view(np.uint32), g[t + "_img"]), m
        assert np.array_equal(mask.cpu().numpy(), (~np.isnan(ref)).astype(np.float32)), m
        assert tuple(emb.img_shape) == tuple(g[t + "_shape"]), m
        for name, conv in (("f32", as_f32), ("f64", as_f64)):
            y = emb.img_to_ts(cu(conv(g[t + f"_in_{name}"])))
            assert y.dtype == torch.float32
            assert inverse_close(y.cpu().numpy(), g[t + f"_out_{name}"], m), (m, name)   # bits up to 4 cells per step
            assert np.array_equal(y.cpu().numpy(), O.img_to_ts(conv(g[t + f"_in_{name}"]), m["L"], m["d"], m["E"]))


def test_delay_embedding_full_size_roundtrip_and_backward(ops):
    """BASELINE shapes B1 / C / D: embed -> inverse reproduces the series up to the 1-ulp averaging of
    overlapped cells, and the adjoint kernel equals the oracle's autograd."""
    from imageni2r_amd import DelayEmbedder
    for (B, L, C, d, E) in ((32, 96, 12, 6, 16), (32, 96, 321, 6, 16), (32, 768, 7, 24, 32)):
        gen = torch.Generator().manual_seed(L + C)
        ts = torch.rand(B, L, C, generator=gen)
        emb = DelayEmbedder(DEV, L, d, E, B, C)
        img = emb.ts_to_img(ts.to(DEV))
        assert np.array_equal(img.cpu().numpy(), O.ts_to_img(ts.numpy(), d, E))
        back = emb.img_to_ts(img)
        assert np.array_equal(back.cpu().numpy(), O.img_to_ts(img.cpu().numpy(), L, d, E))
        assert (back.cpu() - ts).abs().max() < 2e-7
    cols, S = O.delay_geometry(96, 6, 16)
    dimg = torch.randn(2, 3, S, S)
    dts = ops.ts_to_img_bwd(dimg.to(DEV), 96, 6, 16, cols).cpu().numpy()
    idx = O.delay_index_map(96, 6, 16)
    ref = np.zeros((2, 96, 3), dtype=np.float64)
    for r in range(16):
        for c in range(cols):
            if idx[r, c] >= 0:
                ref[:, idx[r, c], :] += dimg[:, :, r, c].numpy()
    assert_close(dts, ref, 1e-6)


def test_delay_edge_cases(ops):
    from imageni2r_amd import DelayEmbedder
    # delay > embedding: some time steps have no image cell and must come back as 0 (img_transformations.py:153-155)
    emb = DelayEmbedder(DEV, 24, 10, 8, 2, 1)
    ts = torch.rand(2, 24, 1)
    img = emb.ts_to_img(ts.to(DEV))
    assert np.array_equal(img.cpu().numpy(), O.ts_to_img(ts.numpy(), 10, 8))
    back = emb.img_to_ts(img).cpu().numpy()
    assert np.array_equal(back, O.img_to_ts(img.cpu().numpy(), 24, 10, 8))
    assert (back[:, 8:10, :] == 0).all()
    # all-NaN series
    ts = torch.full((1, 24, 2), float("nan"))
    emb = DelayEmbedder(DEV, 24, 3, 8, 1, 2)
    img, mask = emb.ts_to_img_with_mask(ts.to(DEV))
    assert np.array_equal(mask.cpu().numpy(), O.observed_mask(O.ts_to_img(ts.numpy(), 3, 8)))


# ------------------------------------------------------------------------------------------------ layers (golden G3)
def test_pos_embedding(ops):
    g = load_golden("g3_layers.npz")
    out = ops.pos_embedding(cu(g["pe_in"]), 128).cpu().numpy()
    assert_close(out, g["pe_out"], 1e-6)


@pytest.mark.parametrize("C,H", [(64, 8), (128, 4), (96, 2)])
def test_groupnorm_golden(ops, C, H):
    g = load_golden("g3_layers.npz")
    t = f"gn_C{C}_H{H}"
    x, w, b, dy = cu(g[t + "_x"]), cu(g[t + "_w"]), cu(g[t + "_b"]), cu(g[t + "_dy"])
    G = min(32, C // 4)
    y, mean, rstd = ops.groupnorm_fwd(x, w, b, G, 1e-5, save_stats=True)
    assert_close(y.cpu().numpy(), g[t + "_y"], TOL)
    dx, dw, db, _ = ops.groupnorm_bwd(x, w, b, None, mean, rstd, dy, G, False)
    assert_close(dx.cpu().numpy(), g[t + "_dx"], TOL)
    assert_close(dw.cpu().numpy(), g[t + "_dw"], TOL)
    assert_close(db.cpu().numpy(), g[t + "_db"], TOL)


def test_groupnorm_fused_modes_vs_oracle(ops):
    torch.manual_seed(11)
    B, C, H = 3, 128, 8
    x = torch.randn(B, C, H, H) * 2 + 0.5
    w, b = torch.randn(C), torch.randn(C)
    ss = (torch.randn(B, 2 * C) * 0.5).requires_grad_(True)
    dy = torch.randn(B, C, H, H)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    scale, shift = ss[:, :C, None, None], ss[:, C:, None, None]
    ref = torch.nn.functional.silu(shift + O.group_norm(xr, wr, br) * (scale + 1))
    ref.backward(dy)
    y, mean, rstd = ops.groupnorm_fwd(cu(x.numpy()), cu(w.numpy()), cu(b.numpy()), 32, 1e-5,
                                      ss=cu(ss.detach().numpy()), act=True, save_stats=True)
    assert_close(y.cpu().numpy(), ref.detach().numpy(), TOL)
    dx, dw, db, dss = ops.groupnorm_bwd(cu(x.numpy()), cu(w.numpy()), cu(b.numpy()), cu(ss.detach().numpy()), mean,
                                        rstd, cu(dy.numpy()), 32, True)
    assert_close(dx.cpu().numpy(), xr.grad.numpy(), TOL)
    assert_close(dw.cpu().numpy(), wr.grad.numpy(), TOL)
    assert_close(db.cpu().numpy(), br.grad.numpy(), TOL)
    assert_close(dss.cpu().numpy(), ss.grad.numpy(), TOL)
    # broadcast (batch-1) modulation as used by the sampler + SiLU-only mode
    y1, _, _ = ops.groupnorm_fwd(cu(x.numpy()), cu(w.numpy()), cu(b.numpy()), 32, 1e-5,
                                 ss=cu(ss.detach().numpy()[:1]), act=True)
    ref1 = torch.nn.functional.silu(ss[:1, C:, None, None] + O.group_norm(x, w, b) * (ss[:1, :C, None, None] + 1))
    assert_close(y1.cpu().numpy(), ref1.detach().numpy(), TOL)
    y2, _, _ = ops.groupnorm_fwd(cu(x.numpy()), cu(w.numpy()), cu(b.numpy()), 32, 1e-5, act=True)
    assert_close(y2.cpu().numpy(), torch.nn.functional.silu(O.group_norm(x, w, b)).numpy(), TOL)


def test_groupnorm_dropout_is_consistent(ops):
    """Dropout masks are regenerated from (seed, offset): forward keep-rate ~ 1-p, surviving values are scaled by
    1/(1-p), and backward uses the same mask."""
    torch.manual_seed(5)
    B, C, H, p = 4, 128, 16, 0.1
    x = torch.randn(B, C, H, H)
    w, b = torch.ones(C), torch.zeros(C)
    xd, wd, bd = cu(x.numpy()), cu(w.numpy()), cu(b.numpy())
    y0, mean, rstd = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, save_stats=True)
    y1, _, _ = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, drop_p=p, seed=123, offset=7)
    y1b, _, _ = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, drop_p=p, seed=123, offset=7)
    y2, _, _ = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, drop_p=p, seed=123, offset=8)
    assert torch.equal(y1, y1b) and not torch.equal(y1, y2)
    keep = (y1 != 0)
    rate = keep.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01
    assert torch.allclose(y1[keep], y0[keep] / (1 - p), rtol=1e-6, atol=1e-7)
    dy = torch.ones_like(xd)
    dxa, *_ = ops.groupnorm_bwd(xd, wd, bd, None, mean, rstd, dy * keep.float() / (1 - p), 32, False)
    dxb, *_ = ops.groupnorm_bwd(xd, wd, bd, None, mean, rstd, dy, 32, False, drop_p=p, seed=123, offset=7)
    assert_close(dxb.cpu().numpy(), dxa.cpu().numpy(), 1e-5)


@pytest.mark.parametrize("name", ["c3", "c1", "c3up", "c3down", "k0up", "k0down", "c3_1x1sp"])
def test_conv_golden(ops, name):
    from imageni2r_amd import Conv2d
    g = load_golden("g3_layers.npz")
    t = "conv_" + name
    has_w = (t + "_w") in g
    x = cu(g[t + "_x"]).requires_grad_(True)
    kernel = 0 if not has_w else g[t + "_w"].shape[-1]
    conv = Conv2d(x.shape[1], g[t + "_y"].shape[1], kernel, up="up" in name, down="down" in name)
    if has_w:
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(g[t + "_w"]))
            conv.bias.copy_(torch.from_numpy(g[t + "_b"]))
    conv = conv.to(DEV)
    y = conv(x)
    y.backward(cu(g[t + "_dy"]))
    assert_close(y.detach().cpu().numpy(), g[t + "_y"], TOL)
    assert_close(x.grad.cpu().numpy(), g[t + "_dx"], TOL)
    if has_w:
        assert_close(conv.weight.grad.cpu().numpy(), g[t + "_dw"], TOL)
        assert_close(conv.bias.grad.cpu().numpy(), g[t + "_db"], TOL)


@pytest.mark.parametrize("B,Cin,Cout,H,k", [(4, 128, 128, 16, 3), (2, 384, 256, 8, 3), (3, 1024, 512, 2, 3),
                                            (2, 12, 128, 16, 3), (2, 128, 12, 16, 3), (2, 256, 768, 8, 1),
                                            (5, 321, 128, 16, 3), (2, 128, 321, 16, 3), (3, 256, 256, 1, 3),
                                            (2, 768, 256, 4, 1), (33, 64, 96, 4, 3), (1, 7, 128, 32, 3),
                                            (16, 256, 256, 4, 3), (40, 96, 160, 8, 3), (2, 13, 40, 4, 1)])
def test_conv_shapes_vs_oracle(ops, B, Cin, Cout, H, k):
    """Every tile variant / edge (ragged Cout, ragged positions, tiny K, 1x1 spatial) of fwd, dgrad and wgrad."""
    gen = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, H, generator=gen)
    dy = torch.randn(B, Cout, H, H, generator=gen)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr, wr, br, padding=k // 2) + res
    ref.backward(dy)
    y = ops.conv2d_fwd(cu(x.numpy()), cu(w.numpy()), cu(b.numpy()), cu(res.numpy()))
    assert_close(y.cpu().numpy(), ref.detach().numpy(), TOL)
    dx = ops.conv2d_dgrad(cu(dy.numpy()), cu(w.numpy()))
    assert_close(dx.cpu().numpy(), xr.grad.numpy(), TOL)
    dw, db = ops.conv2d_wgrad(cu(x.numpy()), cu(dy.numpy()), k)
    assert_close(dw.cpu().numpy(), wr.grad.numpy(), TOL)
    assert_close(db.cpu().numpy(), br.grad.numpy(), TOL)


def test_conv_c_abi_convenience_entry_points(ops):
    """i2r_conv2d_fwd_f32 / i2r_conv2d_dgrad_f32 (pack-on-the-fly forms) called directly through ctypes, with and
    without room for the split-K slabs in the workspace."""
    import ctypes
    from imageni2r_amd import _lib
    gen = torch.Generator().manual_seed(77)
    B, Cin, Cout, H, k = 3, 512, 256, 2, 3
    x = torch.randn(B, Cin, H, H, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=gen)
    dy = torch.randn(B, Cout, H, H, generator=gen)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1)
    xd, wd, bd, dyd = cu(x.numpy()), cu(w.numpy()), cu(b.numpy()), cu(dy.numpy())
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pf = _lib.query("i2r_conv2d_packed_floats", Cout, Cin, k, 0)
    sk = _lib.query("i2r_conv2d_splitk_floats", B, Cin, H, H, Cout, k)
    assert sk > 0  # this shape (12 positions, K = 4608) must want split-K
    for extra in (0, sk):
        ws = torch.empty(pf + extra, device=DEV)
        out = torch.empty(B, Cout, H, H, device=DEV)
        _lib.call("i2r_conv2d_fwd_f32", vp(xd), vp(wd), vp(bd), None, vp(out), vp(ws), pf + extra, B, Cin, H, H, Cout,
                  k, st)
        assert_close(out.cpu().numpy(), ref.numpy(), TOL)
    pfd = _lib.query("i2r_conv2d_packed_floats", Cout, Cin, k, 1)
    skd = _lib.query("i2r_conv2d_splitk_floats", B, Cout, H, H, Cin, k)
    ws = torch.empty(pfd + skd, device=DEV)
    dx = torch.empty(B, Cin, H, H, device=DEV)
    _lib.call("i2r_conv2d_dgrad_f32", vp(dyd), vp(wd), vp(ws), pfd + skd, vp(dx), B, Cin, H, H, Cout, k, st)
    assert_close(dx.cpu().numpy(), ref_dx.numpy(), TOL)
    with pytest.raises(_lib.I2RError):  # workspace too small for the packed weights -> loud error
        _lib.call("i2r_conv2d_fwd_f32", vp(xd), vp(wd), vp(bd), None, vp(out), vp(ws), 16, B, Cin, H, H, Cout, k, st)


@pytest.mark.parametrize("heads,C,H", [(1, 64, 4), (2, 128, 8), (2, 128, 1)])
def test_attention_golden(ops, heads, C, H):
    g = load_golden("g3_layers.npz")
    t = f"attn_h{heads}_C{C}_H{H}"
    qkv = cu(g[t + "_qkv"])
    a, lse = ops.attention_fwd(qkv, heads, save_lse=True)
    assert_close(a.cpu().numpy(), g[t + "_a"], TOL)
    dqkv = ops.attention_bwd(qkv, a, lse, cu(g[t + "_da"]), heads)
    assert_close(dqkv.cpu().numpy(), g[t + "_dqkv"], TOL)


@pytest.mark.parametrize("B,heads,H", [(2, 2, 16), (1, 4, 32), (3, 8, 2), (2, 4, 4), (1, 2, 8)])
def test_attention_shapes_vs_oracle(ops, B, heads, H):
    """T = 256 and 1024 (flash path, several key tiles), T = 4, 16 (LDS path), T = 64."""
    gen = torch.Generator().manual_seed(B + heads + H)
    C = heads * 64
    qkv = (torch.randn(B, 3 * C, H, H, generator=gen) * 1.5).requires_grad_(True)
    da = torch.randn(B, C, H, H, generator=gen)
    ref = O.attention(qkv, heads)
    ref.backward(da)
    q = cu(qkv.detach().numpy())
    a, lse = ops.attention_fwd(q, heads, save_lse=True)
    assert_close(a.cpu().numpy(), ref.detach().numpy(), TOL)
    dqkv = ops.attention_bwd(q, a, lse, cu(da.numpy()), heads)
    assert_close(dqkv.cpu().numpy(), qkv.grad.numpy(), TOL)


def test_attention_large_scores_online_softmax(ops):
    """Force the running-max rescale branch: one key tile far larger than the earlier ones."""
    gen = torch.Generator().manual_seed(3)
    B, heads, H = 1, 1, 16
    qkv = torch.randn(B, 192, H, H, generator=gen)
    qkv.view(B, 64, 3, H * H)[:, :, 1, 200:] *= 6.0  # keys in late tiles dominate
    ref = O.attention(qkv, heads)
    a, _ = ops.attention_fwd(cu(qkv.numpy()), heads)
    assert_close(a.cpu().numpy(), ref.numpy(), TOL)


def test_linear_vs_oracle(ops):
    gen = torch.Generator().manual_seed(9)
    for M, K, N, act in ((1, 128, 512, True), (32, 512, 512, True), (7, 512, 1024, False), (4, 32, 33, False)):
        x = torch.randn(M, K, generator=gen, requires_grad=True)
        w = (torch.randn(N, K, generator=gen) / np.sqrt(K)).requires_grad_(True)
        b = torch.randn(N, generator=gen, requires_grad=True)
        dy = torch.randn(M, N, generator=gen)
        ref = O.linear(x, w, b)
        ref = torch.nn.functional.silu(ref) if act else ref
        ref.backward(dy)
        xd, wd, bd = cu(x.detach().numpy()).requires_grad_(True), cu(w.detach().numpy()).requires_grad_(True), \
            cu(b.detach().numpy()).requires_grad_(True)
        y = ops.LinearFn.apply(xd, wd, bd, act)
        y.backward(cu(dy.numpy()))
        assert_close(y.detach().cpu().numpy(), ref.detach().numpy(), TOL)
        assert_close(xd.grad.cpu().numpy(), x.grad.numpy(), TOL)
        assert_close(wd.grad.cpu().numpy(), w.grad.numpy(), TOL)
        assert_close(bd.grad.cpu().numpy(), b.grad.numpy(), TOL)


@pytest.mark.parametrize("name,cin,cout,kw", [("plain", 16, 16, {}), ("down", 16, 16, dict(down=True)),
                                              ("up", 16, 16, dict(up=True)), ("chg", 24, 16, {}),
                                              ("attn", 32, 64, dict(attention=True))])
def test_unet_block_golden(ops, name, cin, cout, kw):
    from imageni2r_amd import UNetBlock
    g = load_golden("g3_layers.npz")
    t = "blk_" + name
    blk = UNetBlock(in_channels=cin, out_channels=cout, emb_channels=32, dropout=0.10, **kw).eval()
    with torch.no_grad():
        for k, p in blk.named_parameters():
            p.copy_(torch.from_numpy(g[t + "_p_" + k]))
    blk = blk.to(DEV)
    x = cu(g[t + "_x"]).requires_grad_(True)
    y = blk(x, cu(g[t + "_emb"]))
    y.backward(cu(g[t + "_dy"]))
    assert_close(y.detach().cpu().numpy(), g[t + "_y"], TOL)
    assert_close(x.grad.cpu().numpy(), g[t + "_dx"], TOL)
    for k, p in blk.named_parameters():
        assert_close(p.grad.cpu().numpy(), g[t + "_g_" + k], TOL, k)


# ------------------------------------------------------------------------------------------------ net / loss / sampler
@pytest.fixture(scope="module")
def tiny_gpu():
    from imageni2r_amd import TS2img_Karras
    torch.manual_seed(0)
    m = TS2img_Karras(tiny_args(device=DEV), DEV)
    O.perturb_zero_init(m.net.state_dict(), seed=1234)
    return m.to(DEV).eval()


def test_denoiser_golden(tiny_gpu):
    g = load_golden("g4_net.npz")
    with torch.no_grad():
        D = tiny_gpu.net(cu(g["x"]), cu(g["sigma"]))
        assert_close(D.cpu().numpy(), g["D_per_sample"], TOL)
        D2 = tiny_gpu.net(cu(g["x"]).double(), torch.tensor(1.7, dtype=torch.float64))
        assert D2.dtype == torch.float32
        assert_close(D2.cpu().numpy(), g["D_scalar_sigma"], TOL)


def test_denoiser_and_sampler_golden_f16_option(tiny_gpu):
    """The optional reduced-precision inference path (north_star: 1e-2): denoiser output and the 18-step sampler
    against the same reference vectors as the default path."""
    import imageni2r_amd
    from imageni2r_amd import DiffusionProcess
    g, g5 = load_golden("g4_net.npz"), load_golden("g5_sampler.npz")
    imageni2r_amd.set_conv_math("f16")
    try:
        with torch.no_grad():
            D = tiny_gpu.net(cu(g["x"]), cu(g["sigma"]))
        e = rel_err(D.cpu().numpy(), g["D_per_sample"])
        assert TOL * 1e-2 < e < 1e-2, e
        out = DiffusionProcess(tiny_args(device=DEV), tiny_gpu.net, (3, 8, 8), use_graph=False).sample(
            torch.from_numpy(g5["xT"]))
        assert_close(out.cpu().numpy(), g5["out"], 1e-2)
    finally:
        imageni2r_amd.set_conv_math("f16x2")
    with torch.no_grad():  # and the default path is back
        assert_close(tiny_gpu.net(cu(g["x"]), cu(g["sigma"])).cpu().numpy(), g["D_per_sample"], TOL)


@pytest.fixture(params=["0", "1"], ids=["fwd_f32", "fwd_f16x2"])
def train_fwd(request):
    """Training conv forwards on the exact fp32 kernels and on the f16x2-split kernels (what "auto" picks for convs
    with >= 2048 positions; the tiny net is below that, so the mode is forced here).  Same 1e-4 bar for both."""
    from imageni2r_amd import ops
    before = ops._TRAIN_FWD_H2[0]
    ops.set_train_fwd_h2(request.param)
    ops.new_pack_epoch()
    yield request.param
    ops.set_train_fwd_h2(before)
    ops.new_pack_epoch()


def test_loss_and_gradients_golden(tiny_gpu, train_fwd):
    g = load_golden("g4_net.npz")
    m = tiny_gpu
    m.zero_grad()
    m.inject_randomness(cu(g["loss_rnd"]), cu(g["loss_noise"]))
    try:
        loss, to_log = m.loss_fn_irregular(cu(g["loss_x_tilde"]), cu(g["loss_mask"]))
        assert isinstance(to_log["karras loss"], float)
        assert abs(loss.item() - g["loss_value"][0]) <= TOL * abs(g["loss_value"][0])
        loss.backward()
        names = json.loads(bytes(g["grad_names"]).decode())
        params = dict(m.net.named_parameters())
        assert list(params.keys()) == names
        summ = g["grad_summary"]
        for i, k in enumerate(names):
            gr = params[k].grad.double().cpu().numpy().ravel()
            got = np.array([gr.sum(), (gr * gr).sum(), (gr * probe_vec(gr.size, 9000 + i)).sum()])
            scale = max(np.sqrt(summ[i][1]) * np.sqrt(gr.size), 1e-12)
            assert abs(got[1] - summ[i][1]) <= 2 * TOL * max(summ[i][1], 1e-20), k
            assert abs(got[0] - summ[i][0]) <= TOL * scale, k
            assert abs(got[2] - summ[i][2]) <= TOL * scale, k
            if ("grad_full_" + k) in g:
                assert_close(params[k].grad.cpu().numpy(), g["grad_full_" + k], TOL, k)
        # mask=None branch (NaN-derived mask)
        x_img = m.ts_to_img(cu(as_f32(g["loss_ts"])))
        loss2, _ = m.loss_fn_irregular(x_img, None)
        assert abs(loss2.item() - g["loss_value_nanmask"][0]) <= TOL * abs(g["loss_value_nanmask"][0])
    finally:
        m.inject_randomness(None, None)
        m.zero_grad()


def test_full_gradients_vs_oracle(tiny_gpu, train_fwd):
    """Every parameter gradient, element by element, against the oracle's autograd on the same inputs."""
    g = load_golden("g4_net.npz")
    m = tiny_gpu
    a = tiny_args()
    spec = O.unet_spec(a.img_resolution, a.input_channels, a.unet_channels, a.ch_mult, a.attn_resolution)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.net.state_dict().items()
          if not k.endswith("resample_filter")}
    ref = O.masked_edm_loss(sd, spec, torch.from_numpy(g["loss_x_tilde"]), torch.from_numpy(g["loss_mask"]),
                            torch.from_numpy(g["loss_rnd"]), torch.from_numpy(g["loss_noise"]))
    ref.backward()
    m.zero_grad()
    m.inject_randomness(cu(g["loss_rnd"]), cu(g["loss_noise"]))
    try:
        loss, _ = m.loss_fn_irregular(cu(g["loss_x_tilde"]), cu(g["loss_mask"]))
        loss.backward()
    finally:
        m.inject_randomness(None, None)
    worst = 0.0
    for k, p in m.net.named_parameters():
        e = rel_err(p.grad.cpu().numpy(), sd[k].grad.numpy())
        worst = max(worst, e)
        assert e < TOL, (k, e)
    m.zero_grad()


def test_sampler_golden(tiny_gpu):
    from imageni2r_amd import DiffusionProcess
    g = load_golden("g5_sampler.npz")
    args = tiny_args(device=DEV)
    for use_graph in (False, True):
        proc = DiffusionProcess(args, tiny_gpu.net, (3, 8, 8), use_graph=use_graph)
        out = proc.sampling(sampling_number=2, xT=torch.from_numpy(g["xT"]))
        assert out.dtype == torch.float64 and tuple(out.shape) == (2, 3, 8, 8)
        assert_close(out.cpu().numpy(), g["out"], TOL, use_graph)
        if use_graph:  # replay gives the same answer again, and a new xT gives a different one
            out2 = proc.sampling(sampling_number=2, xT=torch.from_numpy(g["xT"]))
            assert torch.equal(out, out2)
            out3 = proc.sampling(sampling_number=2)
            assert not torch.equal(out, out3)
    tiny_gpu.ts_img.img_shape = (2, 3, 8, 7)
    ts = tiny_gpu.img_to_ts(out)
    ref_ts = as_f32(g["ts_out"])
    assert_close(ts.cpu().numpy(), ref_ts, TOL)


def test_sampler_fp64_latents_enter_the_state_unrounded(tiny_gpu):
    """models/sampler.py:45 is `latents.to(torch.float64) * t_steps[0]`: latents handed over in fp64 are not rounded to
    fp32 on the way in (sampling(N) itself draws fp32, so only a caller's own xT can be fp64)."""
    from imageni2r_amd import DiffusionProcess, ops as _ops
    gen = torch.Generator().manual_seed(31)
    xT = torch.randn(2, 3, 8, 8, generator=gen, dtype=torch.float64)
    x0 = _ops.heun_init(xT.to(DEV), 80.0)
    assert torch.equal(x0.cpu(), xT * 80.0) and not torch.equal(x0.cpu(), xT.float().double() * 80.0)
    args = tiny_args(device=DEV)
    sd = {k: v.detach().cpu() for k, v in tiny_gpu.net.state_dict().items()}
    a = tiny_args()
    spec = O.unet_spec(a.img_resolution, a.input_channels, a.unet_channels, a.ch_mult, a.attn_resolution)
    with torch.no_grad():
        ref = O.heun_sample(lambda xx, s: O.edm_precond(sd, spec, xx, s), xT)
    for use_graph in (False, True):
        out = DiffusionProcess(args, tiny_gpu.net, (3, 8, 8), use_graph=use_graph).sample(xT)
        assert out.dtype == torch.float64
        assert_close(out.cpu().numpy(), ref.numpy(), TOL, use_graph)


def test_sampler_two_concurrent_half_batch_graphs(tiny_gpu):
    """Large batches are sampled as two half-batch graphs replayed side by side on two streams: same series as one
    graph over the whole batch (per-series arithmetic does not depend on the batch), and stable across replays."""
    from imageni2r_amd import DiffusionProcess
    args = tiny_args(device=DEV)
    xT = torch.randn(8, 3, 8, 8, generator=torch.Generator().manual_seed(5))
    one = DiffusionProcess(args, tiny_gpu.net, (3, 8, 8))
    one.dual_stream_min = 0
    ref = one.sample(xT)
    two = DiffusionProcess(args, tiny_gpu.net, (3, 8, 8))
    two.dual_stream_min = 4
    out = two.sample(xT)
    halves = [k for k in two._graphs if k[0] == (4, 3, 8, 8)]  # (the graph cache lives on the net, shared by both)
    assert len(halves) == 2 and tuple(out.shape) == (8, 3, 8, 8)
    assert_close(out.cpu().numpy(), ref.cpu().numpy(), 1e-5)
    assert torch.equal(out, two.sample(xT))
    g = load_golden("g5_sampler.npz")  # and the golden pair still comes out right through the split path
    two.dual_stream_min = 2
    assert_close(two.sample(torch.from_numpy(g["xT"])).cpu().numpy(), g["out"], TOL)


def test_sampler_bulk_request_is_sampled_in_pieces(tiny_gpu):
    """BASELINE config E (bulk generation): a request above what one replay may hold (the kernels' 2 GiB addressing
    limit) is cut into pieces sampled one after the other -- same series, in order, ragged tail included."""
    from imageni2r_amd import DiffusionProcess
    args = tiny_args(device=DEV)
    xT = torch.randn(11, 3, 8, 8, generator=torch.Generator().manual_seed(9))
    one = DiffusionProcess(args, tiny_gpu.net, (3, 8, 8))
    one.dual_stream_min = 0
    ref = one.sample(xT)
    for dual_min, cap in ((4, 2), (0, 3), (6, 4)):
        proc = DiffusionProcess(args, tiny_gpu.net, (3, 8, 8))
        proc.dual_stream_min, proc.max_graph_batch = dual_min, cap
        out = proc.sample(xT)
        assert out.dtype == torch.float64 and tuple(out.shape) == (11, 3, 8, 8)
        assert_close(out.cpu().numpy(), ref.cpu().numpy(), 1e-5, (dual_min, cap))
    assert max(k[0][0] for k in tiny_gpu.net._i2r_sampler_graphs if k[0][1:] == (3, 8, 8)) == 11  # only `one` took all 11


def test_ema_scope_swap_samples_with_ema_weights_and_reuses_graphs(monkeypatch):
    """run_irregular.py:250-259: a new DiffusionProcess per evaluation inside ``ema_scope``.  The pointer-swapped scope
    gives the same series as the reference's copy sequence, the second evaluation replays the first one's graph, and
    after shadow weights change (training in between) the replay follows them."""
    from imageni2r_amd import DiffusionProcess, TS2img_Karras
    torch.manual_seed(11)
    m = TS2img_Karras(tiny_args(device=DEV), DEV).to(DEV)  # as run_irregular.py:114 does
    with torch.no_grad():
        for p in m.net.parameters():  # zero-init layers would make every denoiser output trivial
            if float(p.abs().max()) == 0:
                p.normal_(0, 0.05)
        for b in m.model_ema.buffers():
            if b.dtype == torch.float32 and b.ndim > 0:
                b.copy_(torch.randn_like(b) * 0.05)
    args = tiny_args(device=DEV)
    xT = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(3))
    live = DiffusionProcess(args, m.net, (3, 8, 8), use_graph=False).sample(xT)
    monkeypatch.setenv("I2R_EMA_SCOPE", "copy")
    with m.ema_scope():
        ref = DiffusionProcess(args, m.net, (3, 8, 8), use_graph=False).sample(xT)
    monkeypatch.setenv("I2R_EMA_SCOPE", "swap")
    assert not torch.equal(ref, live)
    with m.ema_scope():
        out = DiffusionProcess(args, m.net, (3, 8, 8)).sample(xT)
    n_graphs = len(m.net._i2r_sampler_graphs)
    assert torch.equal(out, ref)  # same kernels, same weights: bit-equal
    assert torch.equal(DiffusionProcess(args, m.net, (3, 8, 8), use_graph=False).sample(xT), live)  # weights are back
    with torch.no_grad():
        for b in m.model_ema.buffers():
            if b.dtype == torch.float32 and b.ndim > 0:
                b.mul_(1.5)
    with m.ema_scope():
        out2 = DiffusionProcess(args, m.net, (3, 8, 8)).sample(xT)
        ref2 = DiffusionProcess(args, m.net, (3, 8, 8), use_graph=False).sample(xT)
    assert len(m.net._i2r_sampler_graphs) == n_graphs == 1
    assert torch.equal(out2, ref2) and not torch.equal(out2, out)


def test_training_step_with_dropout_runs(tiny_gpu):
    """train() mode: dropout live, in-kernel Philox noise; loss finite, every parameter gets a finite gradient."""
    m = tiny_gpu
    m.train()
    try:
        ts = torch.rand(4, 24, 3)
        ts[0, [2, 3], :] = float("nan")
        x_img, mask = m.ts_to_img_with_mask(ts.to(DEV))
        x_tilde = m.ts_to_img(torch.rand(4, 24, 3).to(DEV))
        loss, _ = m.loss_fn_irregular(x_tilde, mask)
        loss.backward()
        assert torch.isfinite(loss)
        for k, p in m.net.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
        m.on_train_batch_end()
    finally:
        m.eval()
        m.zero_grad()


def test_chunked_attention_tail_matches_whole_batch(tiny_gpu):
    """Inference at large batch runs norm2 -> qkv -> attention -> proj over sample chunks (cache residency); every
    series must come out exactly as from the whole-batch pass."""
    from imageni2r_amd import networks, ops
    net = tiny_gpu.net
    x = torch.randn(8, 3, 8, 8, generator=torch.Generator().manual_seed(9)).to(DEV)
    sig = torch.tensor([0.7], device=DEV)
    with torch.no_grad():
        old = ops.knob("attn_chunk_bytes")
        try:
            ops.set_knob("attn_chunk_bytes", 0)
            ref = net(x, sig)
            ops.set_knob("attn_chunk_bytes", 3 * 64 * 64 * 4 * 2)  # two samples' qkv at the 8x8 level
            blk = next(b for b in net.model.enc.values() if isinstance(b, networks.UNetBlock) and b.num_heads)
            assert blk._attn_chunk(torch.empty(8, 64, 8, 8)) == 2
            out = net(x, sig)
        finally:
            ops.set_knob("attn_chunk_bytes", old)
    assert torch.equal(out, ref)
