"""Pins the CPU oracle (oracle/i2r_oracle.py) against the golden vectors captured from the reference itself
(tests/golden/make_golden.py).  Runs on CPU; no GPU, no /root/reference needed."""
import json

import numpy as np
import pytest
import torch

from tests.helpers import as_f32, as_f64, golden_meta, inverse_close, load_golden, probe_vec, rel_err
from oracle import i2r_oracle as O

TOL = 2e-5  # oracle and reference both run torch-CPU fp32 kernels; only op-order differences remain


def test_delay_embedding_bit_exact():
    g = load_golden("g1_delay.npz")
    for m in golden_meta(g):
        t = m["tag"]
        ts = as_f32(g[t + "_ts"])
        img = O.ts_to_img(ts, m["d"], m["E"])
        ref = as_f32(g[t + "_img"])
        assert img.shape == ref.shape
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), t
        assert np.array_equal(O.observed_mask(img), g[t + "_mask"]), t
        cols, S = O.delay_geometry(m["L"], m["d"], m["E"])
        assert tuple(g[t + "_img_shape"]) == (m["B"], m["C"], m["E"], cols) and S == m["S"]
        mp = [(tt, r, c) for tt, cells in enumerate(O.inverse_positions(m["L"], m["d"], m["E"])) for (r, c) in cells]
        ref_mp = sorted(map(tuple, g[t + "_mapping"].tolist()))
        assert sorted(mp) == ref_mp, t
        # per-t order must be the reference's scan order too
        ref_by_t = {}
        for tt, r, c in g[t + "_mapping"].tolist():
            ref_by_t.setdefault(tt, []).append((r, c))
        for tt, cells in enumerate(O.inverse_positions(m["L"], m["d"], m["E"])):
            assert ref_by_t.get(tt, []) == cells
        for name, conv in (("f32", as_f32), ("f64", as_f64)):
            x = conv(g[t + f"_inv_in_{name}"])
            y = O.img_to_ts(x, m["L"], m["d"], m["E"])
            assert np.array_equal(y.view(np.uint32), g[t + f"_inv_out_{name}"]), (t, name)


def test_delay_embedding_random_geometries_bit_exact():
    """48 random (seq_len, delay, embedding, channels) the reference accepts (g10: delay > embedding, partial last
    column, one-column images, seq_len == embedding ...): image bits, cached shape, inverse for fp32 / fp64 input."""
    g = load_golden("g10_delay_fuzz.npz")
    metas = golden_meta(g)
    assert len(metas) == 48
    exact = 0
    for m in metas:
        t = m["tag"]
        img = O.ts_to_img(as_f32(g[t + "_ts"]), m["d"], m["E"])
        assert np.array_equal(img.view(np.uint32), g[t + "_img"]), m
        cols, S = O.delay_geometry(m["L"], m["d"], m["E"])
        assert tuple(g[t + "_shape"]) == (m["B"], m["C"], m["E"], cols) and S == m["S"], m
        exact += max(len(c) for c in O.inverse_positions(m["L"], m["d"], m["E"])) <= 4
        for name, conv in (("f32", as_f32), ("f64", as_f64)):
            y = O.img_to_ts(conv(g[t + f"_in_{name}"]), m["L"], m["d"], m["E"])
            assert inverse_close(y, g[t + f"_out_{name}"], m), (m, name)
    assert exact == 42  # (geometries of the draw in the bit-exact class)


def test_layers():
    g = load_golden("g3_layers.npz")
    T = lambda k: torch.from_numpy(g[k])
    assert rel_err(O.positional_embedding(T("pe_in"), 128).numpy(), g["pe_out"]) < 1e-6
    for C, H in ((64, 8), (128, 4), (96, 2)):
        t = f"gn_C{C}_H{H}"
        x = T(t + "_x").requires_grad_(True)
        w = T(t + "_w").requires_grad_(True)
        b = T(t + "_b").requires_grad_(True)
        y = O.group_norm(x, w, b)
        y.backward(T(t + "_dy"))
        assert rel_err(y.detach().numpy(), g[t + "_y"]) < TOL
        assert rel_err(x.grad.numpy(), g[t + "_dx"]) < TOL
        assert rel_err(w.grad.numpy(), g[t + "_dw"]) < TOL
        assert rel_err(b.grad.numpy(), g[t + "_db"]) < TOL
    for name, kw in (("c3", {}), ("c1", {}), ("c3up", dict(up=True)), ("c3down", dict(down=True)),
                     ("k0up", dict(up=True)), ("k0down", dict(down=True)), ("c3_1x1sp", {})):
        t = "conv_" + name
        x = T(t + "_x").requires_grad_(True)
        has_w = (t + "_w") in g
        w = T(t + "_w").requires_grad_(True) if has_w else None
        b = T(t + "_b").requires_grad_(True) if has_w else None
        y = O.conv2d(x, w, b, **kw)
        y.backward(T(t + "_dy"))
        assert rel_err(y.detach().numpy(), g[t + "_y"]) < TOL, t
        assert rel_err(x.grad.numpy(), g[t + "_dx"]) < TOL, t
        if has_w:
            assert rel_err(w.grad.numpy(), g[t + "_dw"]) < TOL, t
            assert rel_err(b.grad.numpy(), g[t + "_db"]) < TOL, t
    for heads, C, H in ((1, 64, 4), (2, 128, 8), (2, 128, 1)):
        t = f"attn_h{heads}_C{C}_H{H}"
        qkv = T(t + "_qkv").requires_grad_(True)
        a = O.attention(qkv, heads)
        a.backward(T(t + "_da"))
        assert rel_err(a.detach().numpy(), g[t + "_a"]) < TOL, t
        assert rel_err(qkv.grad.numpy(), g[t + "_dqkv"]) < TOL, t


@pytest.mark.parametrize("name,cin,cout,kw", [("plain", 16, 16, {}), ("down", 16, 16, dict(down=True)),
                                              ("up", 16, 16, dict(up=True)), ("chg", 24, 16, {}),
                                              ("attn", 32, 64, dict(heads=1))])
def test_unet_block(name, cin, cout, kw):
    g = load_golden("g3_layers.npz")
    t = "blk_" + name
    sd = {k[len(t) + 3:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith(t + "_p_")}
    bs = O.BlockSpec("b", "block", cin, cout, up=kw.get("up", False), down=kw.get("down", False),
                     heads=kw.get("heads", 0), skip_conv=cin != cout)
    x = torch.from_numpy(g[t + "_x"]).requires_grad_(True)
    y = O.unet_block({"b." + k: v for k, v in sd.items()}, "b", bs, x, torch.from_numpy(g[t + "_emb"]))
    y.backward(torch.from_numpy(g[t + "_dy"]))
    assert rel_err(y.detach().numpy(), g[t + "_y"]) < TOL
    assert rel_err(x.grad.numpy(), g[t + "_dx"]) < TOL
    for k, v in sd.items():
        assert rel_err(v.grad.numpy(), g[t + "_g_" + k]) < 5e-5, k


def test_net_loss_grads_sampler(tiny_state):
    m, spec = tiny_state
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.net.state_dict().items()
          if not k.endswith("resample_filter")}
    g = load_golden("g4_net.npz")
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        D = O.edm_precond(sd, spec, x, torch.from_numpy(g["sigma"]))
        assert rel_err(D.numpy(), g["D_per_sample"]) < TOL
        D2 = O.edm_precond(sd, spec, x.double(), torch.tensor(1.7, dtype=torch.float64))
        assert rel_err(D2.numpy(), g["D_scalar_sigma"]) < TOL
    mask = torch.from_numpy(g["loss_mask"])
    loss = O.masked_edm_loss(sd, spec, torch.from_numpy(g["loss_x_tilde"]), mask, torch.from_numpy(g["loss_rnd"]),
                             torch.from_numpy(g["loss_noise"]))
    assert abs(loss.item() - g["loss_value"][0]) < 1e-5 * abs(g["loss_value"][0])
    loss.backward()
    names = json.loads(bytes(g["grad_names"]).decode())
    summ = g["grad_summary"]
    for i, k in enumerate(names):
        gr = sd[k].grad.double().numpy().ravel()
        got = np.array([gr.sum(), (gr * gr).sum(), (gr * probe_vec(gr.size, 9000 + i)).sum()])
        scale = max(np.sqrt(summ[i][1]) * np.sqrt(gr.size), 1e-12)
        assert abs(got[1] - summ[i][1]) <= 2e-4 * max(summ[i][1], 1e-20), k
        assert abs(got[0] - summ[i][0]) <= 1e-4 * scale, k
        assert abs(got[2] - summ[i][2]) <= 1e-4 * scale, k
        if ("grad_full_" + k) in g:
            assert rel_err(sd[k].grad.numpy(), g["grad_full_" + k]) < 1e-4, k
    # NaN-derived mask branch
    x_img = torch.from_numpy(O.ts_to_img(as_f32(g["loss_ts"]), 3, 8))
    loss2 = O.masked_edm_loss(sd, spec, x_img, None, torch.from_numpy(g["loss_rnd"]),
                              torch.from_numpy(g["loss_noise"]))
    assert abs(loss2.item() - g["loss_value_nanmask"][0]) < 1e-5 * abs(g["loss_value_nanmask"][0])

    g5 = load_golden("g5_sampler.npz")
    assert np.array_equal(O.karras_sigmas(18).numpy(), g5["t_steps"])
    with torch.no_grad():
        out = O.heun_sample(lambda xx, s: O.edm_precond(sd, spec, xx, s), torch.from_numpy(g5["xT"]))
    assert out.dtype == torch.float64
    assert rel_err(out.numpy(), g5["out"]) < 1e-4
    ts = O.img_to_ts(g5["out"], 24, 3, 8)
    assert np.array_equal(ts.view(np.uint32), g5["ts_out"])
