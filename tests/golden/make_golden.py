#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself on CPU.

Run in the build container only (it needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

The reference is imported, never copied.  Outputs are small .npz/.json files of inputs and expected
outputs (data, not source).  Inputs are regenerated from fixed seeds by the tests, and also stored where
cheap so the fixtures are self-describing.
"""
import hashlib
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

# the reference ends img_to_ts with an unconditional .cuda() (img_transformations.py:170); on this CPU-only
# container make it the identity for THIS process only.
torch.Tensor.cuda = lambda self, *a, **k: self

from models.img_transformations import DelayEmbedder  # noqa: E402  (reference)
from models import networks as rnet  # noqa: E402  (reference)
from models.our import TS2img_Karras  # noqa: E402  (reference)
from models.sampler import DiffusionProcess  # noqa: E402  (reference)
from models.ema import LitEma  # noqa: E402  (reference)

from oracle.i2r_oracle import perturb_norm_affine, perturb_zero_init  # noqa: E402  (ours; deterministic perturbations)
from tests.helpers import FULLSIZE, fullsize_inputs, tensor_sha  # noqa: E402  (ours; seeded inputs shared with the tests)

TINY = dict(img_resolution=8, input_channels=3, ch_mult=[1, 2], unet_channels=64, attn_resolution=[8, 4],
            delay=3, embedding=8, seq_len=24, batch_size=4, ema=True, ema_warmup=2, diffusion_steps=18,
            beta1=1e-5, betaT=1e-2, deterministic=False, device="cpu")


def tiny_args(**kw):
    d = dict(TINY)
    d.update(kw)
    return SimpleNamespace(**d)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def probe_vec(n, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(n)


def grad_summary(g, seed):
    g64 = g.detach().double().numpy().ravel()
    return np.array([g64.sum(), (g64 * g64).sum(), (g64 * probe_vec(g64.size, seed)).sum()])


# ---------------------------------------------------------------- G1 delay embedding
def g1_delay():
    out = {}
    cases = [(24, 3, 8), (96, 6, 16), (96, 24, 32), (768, 24, 32), (24, 10, 8), (50, 7, 16)]
    meta = []
    for ci, (L, d, E) in enumerate(cases):
        for C in (1, 5):
            B = 2
            tag = f"L{L}_d{d}_E{E}_C{C}"
            rng = np.random.default_rng(1000 + ci * 10 + C)
            ts = rng.random((B, L, C), dtype=np.float32)
            miss = rng.permutation(L)[: int(0.3 * L)]
            ts[0, miss, :] = np.nan
            ts[1, rng.permutation(L)[: L // 5], :] = np.nan
            emb = DelayEmbedder("cpu", L, d, E, B, C)
            img = emb.ts_to_img(torch.from_numpy(ts))
            mask = torch.isnan(img).float() * -1 + 1
            out[tag + "_ts"] = bits(ts)
            out[tag + "_img"] = bits(img.numpy())
            out[tag + "_mask"] = mask.numpy()
            out[tag + "_img_shape"] = np.array(emb.img_shape, dtype=np.int64)
            mp = [(t, r, c) for t, cells in emb.mapping.items() for (r, c) in cells]
            out[tag + "_mapping"] = np.array(mp, dtype=np.int32)
            S = img.shape[-1]
            for dt, name in ((np.float32, "f32"), (np.float64, "f64")):
                x = rng.standard_normal((B, C, S, S)).astype(dt)
                y = emb.img_to_ts(torch.from_numpy(x))
                assert y.dtype == torch.float32
                out[tag + f"_inv_in_{name}"] = bits(x)
                out[tag + f"_inv_out_{name}"] = bits(y.numpy())
            meta.append(dict(tag=tag, L=L, d=d, E=E, C=C, B=B, S=int(S)))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g1_delay.npz"), **out)
    print("g1_delay", len(meta), "cases")


# ---------------------------------------------------------------- G2 init order
def g2_init():
    res = {}
    for seed in (0, 7):
        torch.manual_seed(seed)
        m = TS2img_Karras(tiny_args(), "cpu")
        sd = m.net.state_dict()
        res[str(seed)] = {k: hashlib.sha256(v.numpy().tobytes()).hexdigest()[:16] for k, v in sd.items()}
        res[str(seed) + "_nparams"] = int(sum(p.numel() for p in m.net.parameters()))
    # one full-size shape (sine-24, SURVEY shape A): parameter count + key list only
    torch.manual_seed(0)
    a = tiny_args(ch_mult=[1, 2, 2, 2], unet_channels=128, attn_resolution=[8, 4, 2], input_channels=5)
    m = TS2img_Karras(a, "cpu")
    res["A_nparams"] = int(sum(p.numel() for p in m.net.parameters()))
    res["A_keys"] = list(m.net.state_dict().keys())
    res["A_state_len"] = len(m.state_dict())
    with open(os.path.join(HERE, "g2_init.json"), "w") as f:
        json.dump(res, f, indent=0)
    print("g2_init nparams tiny", res["0_nparams"], "A", res["A_nparams"])


# ---------------------------------------------------------------- G3 layers
def g3_layers():
    out = {}
    torch.manual_seed(3)
    # positional embedding
    pe = rnet.PositionalEmbedding(num_channels=128)
    xs = torch.tensor([-2.1, -0.5, 0.0, 0.3, 1.09], dtype=torch.float32)
    out["pe_in"] = xs.numpy()
    out["pe_out"] = pe(xs).numpy()
    # group norm
    for C, H in ((64, 8), (128, 4), (96, 2)):
        gn = rnet.GroupNorm(C)
        with torch.no_grad():
            gn.weight.copy_(torch.randn(C))
            gn.bias.copy_(torch.randn(C))
        x = torch.randn(2, C, H, H, requires_grad=True)
        y = gn(x)
        dy = torch.randn_like(y)
        y.backward(dy)
        t = f"gn_C{C}_H{H}"
        out[t + "_x"] = x.detach().numpy(); out[t + "_w"] = gn.weight.detach().numpy(); out[t + "_b"] = gn.bias.detach().numpy()
        out[t + "_y"] = y.detach().numpy(); out[t + "_dy"] = dy.numpy()
        out[t + "_dx"] = x.grad.numpy(); out[t + "_dw"] = gn.weight.grad.numpy(); out[t + "_db"] = gn.bias.grad.numpy()
    # conv variants
    init = dict(init_mode='kaiming_uniform', init_weight=np.sqrt(1 / 3), init_bias=np.sqrt(1 / 3))
    for name, kw, cin, cout, H in (("c3", dict(kernel=3), 5, 16, 8), ("c1", dict(kernel=1), 12, 9, 4),
                                   ("c3up", dict(kernel=3, up=True), 8, 8, 4), ("c3down", dict(kernel=3, down=True), 8, 8, 8),
                                   ("k0up", dict(kernel=0, up=True), 6, 6, 4), ("k0down", dict(kernel=0, down=True), 6, 6, 8),
                                   ("c3_1x1sp", dict(kernel=3), 8, 8, 1)):
        conv = rnet.Conv2d(in_channels=cin, out_channels=cout, **kw, **init)
        x = torch.randn(2, cin, H, H, requires_grad=True)
        y = conv(x)
        dy = torch.randn_like(y)
        y.backward(dy)
        t = "conv_" + name
        out[t + "_x"] = x.detach().numpy(); out[t + "_y"] = y.detach().numpy(); out[t + "_dy"] = dy.numpy()
        out[t + "_dx"] = x.grad.numpy()
        if conv.weight is not None:
            out[t + "_w"] = conv.weight.detach().numpy(); out[t + "_b"] = conv.bias.detach().numpy()
            out[t + "_dw"] = conv.weight.grad.numpy(); out[t + "_db"] = conv.bias.grad.numpy()
    # attention op fwd + bwd (interleaved qkv as in UNetBlock.forward)
    for heads, C, H in ((1, 64, 4), (2, 128, 8), (2, 128, 1)):
        B = 2
        qkv = torch.randn(B, 3 * C, H, H, requires_grad=True)
        q, k, v = qkv.reshape(B * heads, C // heads, 3, -1).unbind(2)
        w = rnet.AttentionOp.apply(q, k)
        a = torch.einsum('nqk,nck->ncq', w, v).reshape(B, C, H, H)
        da = torch.randn_like(a)
        a.backward(da)
        t = f"attn_h{heads}_C{C}_H{H}"
        out[t + "_qkv"] = qkv.detach().numpy(); out[t + "_a"] = a.detach().numpy()
        out[t + "_da"] = da.numpy(); out[t + "_dqkv"] = qkv.grad.numpy()
    # UNet blocks (eval mode -> dropout off); zero-init layers perturbed
    block_kwargs = dict(emb_channels=32, channels_per_head=64, dropout=0.10, init=init,
                        init_zero=dict(init_mode='kaiming_uniform', init_weight=0, init_bias=0))
    for name, cin, cout, kw, H in (("plain", 16, 16, {}, 8), ("down", 16, 16, dict(down=True), 8),
                                   ("up", 16, 16, dict(up=True), 4), ("chg", 24, 16, {}, 4),
                                   ("attn", 32, 64, dict(attention=True), 4)):
        blk = rnet.UNetBlock(in_channels=cin, out_channels=cout, **kw, **block_kwargs).eval()
        sd = blk.state_dict()
        perturb_zero_init(sd, seed=55)
        with torch.no_grad():
            for k_, v_ in sd.items():
                if k_.startswith("norm"):
                    v_.copy_(torch.randn(v_.shape) * 0.3 + (1.0 if k_.endswith("weight") else 0.0))
        x = torch.randn(2, cin, H, H, requires_grad=True)
        emb = torch.randn(2, 32)
        y = blk(x, emb)
        dy = torch.randn_like(y)
        y.backward(dy)
        t = "blk_" + name
        out[t + "_x"] = x.detach().numpy(); out[t + "_emb"] = emb.numpy(); out[t + "_y"] = y.detach().numpy()
        out[t + "_dy"] = dy.numpy(); out[t + "_dx"] = x.grad.numpy()
        for k_, v_ in blk.named_parameters():
            out[t + "_p_" + k_] = v_.detach().numpy()
            out[t + "_g_" + k_] = v_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g3_layers.npz"), **out)
    print("g3_layers", len(out), "arrays")


# ---------------------------------------------------------------- G4/G5 net, loss grads, sampler
def build_tiny(seed=0):
    torch.manual_seed(seed)
    m = TS2img_Karras(tiny_args(), "cpu")
    perturb_zero_init(m.net.state_dict(), seed=1234)
    return m


def g4_net():
    out = {}
    m = build_tiny(0).eval()
    g = torch.Generator().manual_seed(42)
    B, C, S = 4, 3, 8
    x = torch.randn(B, C, S, S, generator=g)
    sig = torch.tensor([0.05, 0.4, 2.5, 40.0])
    with torch.no_grad():
        out["D_per_sample"] = m.net(x, sig).numpy()
        out["D_scalar_sigma"] = m.net(x.double(), torch.tensor(1.7, dtype=torch.float64)).numpy()
    out["x"] = x.numpy(); out["sigma"] = sig.numpy()
    # loss + grads with injected randomness: patch torch.randn / randn_like in the reference's namespace
    ts = torch.rand(B, 24, C, generator=g)
    ts[0, [1, 5, 6, 20], :] = float("nan")
    ts[2, [0, 3, 11], :] = float("nan")
    x_img = m.ts_to_img(ts)
    mask = torch.isnan(x_img).float() * -1 + 1
    x_tilde = m.ts_to_img(torch.rand(B, 24, C, generator=g))
    rnd = torch.randn(B, 1, 1, 1, generator=g)
    noise = torch.randn(B, C, S, S, generator=g)
    real_randn, real_randn_like = torch.randn, torch.randn_like
    try:
        torch.randn = lambda *a, **k: rnd.clone()
        torch.randn_like = lambda t, **k: noise.clone()
        loss, to_log = m.loss_fn_irregular(x_tilde, mask)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    loss.backward()
    out["loss_ts"] = bits(ts.numpy()); out["loss_mask"] = mask.numpy(); out["loss_x_tilde"] = x_tilde.numpy()
    out["loss_rnd"] = rnd.numpy(); out["loss_noise"] = noise.numpy()
    out["loss_value"] = np.array([loss.item(), to_log["karras loss"]], dtype=np.float64)
    names, summ = [], []
    for i, (k, p) in enumerate(m.net.named_parameters()):
        names.append(k)
        summ.append(grad_summary(p.grad, 9000 + i))
        if p.numel() <= 512:
            out["grad_full_" + k] = p.grad.numpy()
    out["grad_names"] = np.frombuffer(json.dumps(names).encode(), dtype=np.uint8)
    out["grad_summary"] = np.stack(summ)
    # NaN-derived mask path (mask=None): loss only
    m.zero_grad()
    xn = x_img.clone()
    try:
        torch.randn = lambda *a, **k: rnd.clone()
        torch.randn_like = lambda t, **k: noise.clone()
        loss2, _ = m.loss_fn_irregular(xn, None)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    out["loss_value_nanmask"] = np.array([loss2.item()], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g4_net.npz"), **out)
    print("g4_net loss", loss.item(), loss2.item())

    # G5 sampler on the same net
    o5 = {}
    xT = torch.randn(2, C, S, S, generator=torch.Generator().manual_seed(77))
    proc = DiffusionProcess(tiny_args(), m.net, (C, S, S))
    with torch.no_grad():
        res = proc.sampling(sampling_number=2, xT=xT)
    assert res.dtype == torch.float64
    o5["xT"] = xT.numpy(); o5["out"] = res.numpy()
    step = torch.arange(18, dtype=torch.float64)
    t_steps = (80 ** (1 / 7) + step / 17 * (0.002 ** (1 / 7) - 80 ** (1 / 7))) ** 7
    o5["t_steps"] = torch.cat([t_steps, torch.zeros(1, dtype=torch.float64)]).numpy()
    ts_out = m.img_to_ts(res)
    o5["ts_out"] = bits(ts_out.numpy())
    np.savez_compressed(os.path.join(HERE, "g5_sampler.npz"), **o5)
    print("g5_sampler |out|max", res.abs().max().item())


# ---------------------------------------------------------------- G6 EMA
def g6_ema():
    torch.manual_seed(6)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    ema = LitEma(lin, decay=0.9999, use_num_upates=True, warmup=2)
    out = {"names": np.frombuffer(json.dumps(dict(ema.m_name2s_name)).encode(), dtype=np.uint8)}
    for k, v in lin.named_parameters():
        out["p0_" + k] = v.detach().numpy().copy()
    for step in range(1, 6):
        with torch.no_grad():
            for i, p in enumerate(lin.parameters()):
                p.add_(0.1 * (step + i))
        ema(lin)
        for k, v in ema.named_buffers():
            out[f"s{step}_{k}"] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g6_ema.npz"), **out)
    print("g6_ema done")


# ---------------------------------------------------------------- G7 every BASELINE shape at full size
def g7_fullsize(only=None):
    """Shapes A, B1, C, D through the REFERENCE at full model size: init hashes of all 550 tensors,
    D(x, sigma) for per-sample and 0-dim fp64 sigma, loss_fn_irregular + every parameter's gradient summary with the
    random draws injected, one sampling(1, xT) fp64 output and its img_to_ts bits.  GroupNorm gamma/beta and the
    zero-initialised layers are perturbed deterministically (oracle.perturb_*) so no index error can hide."""
    import time
    path = os.path.join(HERE, "g7_fullsize.npz")
    out = dict(np.load(path)) if (only and os.path.exists(path)) else {}
    ipath = os.path.join(HERE, "g7_init.json")
    init = json.load(open(ipath)) if (only and os.path.exists(ipath)) else {}
    for name in FULLSIZE:
        if only and name not in only:
            continue
        t0 = time.time()
        args, S, d = fullsize_inputs(name)
        C = args.input_channels
        torch.manual_seed(0)
        m = TS2img_Karras(args, "cpu")
        sd = m.net.state_dict()
        init[name] = {"nparams": int(sum(p.numel() for p in m.net.parameters())), "state_len": len(m.state_dict()),
                      "sha": {k: hashlib.sha256(v.numpy().tobytes()).hexdigest()[:16] for k, v in sd.items()},
                      "inputs": {k: tensor_sha(v) for k, v in d.items()}}
        perturb_zero_init(sd, seed=1234)
        perturb_norm_affine(sd, seed=4321)
        m.eval()
        with torch.no_grad():
            out[name + "/D_per_sample"] = m.net(d["x"], d["sigma"]).numpy()
            out[name + "/D_scalar_sigma"] = m.net(d["x"][:1].double(), torch.tensor(1.7, dtype=torch.float64)).numpy()
        x_img = m.ts_to_img(d["ts"])
        mask = torch.isnan(x_img).float() * -1 + 1
        x_tilde = m.ts_to_img(d["ts_tilde"])
        out[name + "/img_bits"] = bits(x_img.numpy())
        real_randn, real_randn_like = torch.randn, torch.randn_like
        try:
            torch.randn = lambda *a, **k: d["rnd"].clone()
            torch.randn_like = lambda t, **k: d["noise"].clone()
            loss, to_log = m.loss_fn_irregular(x_tilde, mask)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        loss.backward()
        out[name + "/loss_value"] = np.array([loss.item(), to_log["karras loss"]], dtype=np.float64)
        names, summ, small, index = [], [], [], []
        off = 0
        for i, (k, p) in enumerate(m.net.named_parameters()):
            names.append(k)
            summ.append(grad_summary(p.grad, 9000 + i))
            if p.numel() <= 1024:  # biases, norm weights: kept element by element
                small.append(p.grad.numpy().ravel())
                index.append([i, off, p.numel()])
                off += p.numel()
        out[name + "/grad_summary"] = np.stack(summ)
        out[name + "/grad_small"] = np.concatenate(small)
        out[name + "/grad_small_index"] = np.array(index, dtype=np.int64)
        init[name]["grad_names"] = names
        m.zero_grad()
        proc = DiffusionProcess(args, m.net, (C, S, S))
        with torch.no_grad():
            res = proc.sampling(sampling_number=1, xT=d["xT"])
        assert res.dtype == torch.float64 and tuple(res.shape) == (1, C, S, S)
        out[name + "/sample"] = res.numpy()
        out[name + "/sample_ts_bits"] = bits(m.img_to_ts(res).numpy())
        print(f"g7 {name}: params {init[name]['nparams']} loss {loss.item():.6f} |sample|max {res.abs().max().item():.4f} "
              f"({time.time() - t0:.1f} s)", flush=True)
    np.savez_compressed(path, **out)
    with open(ipath, "w") as f:
        json.dump(init, f, indent=0)


# ---------------------------------------------------------------- G8 evaluation metrics
def metric_inputs(seed=0, n=160, L=24, C=6):
    """Two seeded sets of series in (0, 1): 'real' = noisy sines, 'generated' = the same family with a different noise
    level and phase law (distinguishable, but not trivially)."""
    rng = np.random.default_rng(seed)
    t = np.arange(L)[None, :, None]
    f = rng.uniform(0.05, 0.3, (n, 1, C)); ph = rng.uniform(0, 2 * np.pi, (n, 1, C))
    real = 0.5 + 0.4 * np.sin(2 * np.pi * f * t + ph) + 0.02 * rng.standard_normal((n, L, C))
    f = rng.uniform(0.05, 0.3, (n, 1, C)); ph = rng.uniform(0, np.pi, (n, 1, C))
    gen = 0.5 + 0.4 * np.sin(2 * np.pi * f * t + ph) + 0.05 * rng.standard_normal((n, L, C))
    return real.astype(np.float32), gen.astype(np.float32)


def g8_metrics():
    from metrics.correlation_score import calculate_pearson_correlation  # (reference)
    from metrics.discriminative_torch import discriminative_score_metrics  # (reference)
    from metrics.predictive_metrics import predictive_score_metrics  # (reference)
    real, gen = metric_inputs()
    args = SimpleNamespace(input_size=real.shape[2], device="cpu")
    out = {"n": int(real.shape[0]), "L": int(real.shape[1]), "C": int(real.shape[2]),
           "real_sha": hashlib.sha256(real.tobytes()).hexdigest()[:16],
           "gen_sha": hashlib.sha256(gen.tobytes()).hexdigest()[:16], "cases": []}
    for seed in (0, 5):
        np.random.seed(seed); torch.manual_seed(seed)
        disc = float(discriminative_score_metrics(real, gen, args))
        np.random.seed(seed); torch.manual_seed(seed)
        pred = float(predictive_score_metrics(real, gen))
        np.random.seed(seed); torch.manual_seed(seed)
        corr = calculate_pearson_correlation(real, gen)
        out["cases"].append({"seed": seed, "disc": disc, "pred": pred, "corr_mean": float(corr[0])})
        print("g8 seed", seed, out["cases"][-1], flush=True)
    with open(os.path.join(HERE, "g8_metrics.json"), "w") as f:
        json.dump(out, f, indent=1)


def g9_context_fid():
    """Context-FID from the reference's own classes (metrics/context_fid.py).  `Context_FID` hard-codes device=0, so its
    five statements are issued here with device='cpu' -- same constructor arguments, same call order -- plus one
    ragged case (NaN-padded heads / tails, which takes the centring branch of fit)."""
    from metrics.context_fid import TS2Vec, calculate_fid_score  # (reference)
    real, gen = metric_inputs()
    out = {"real_sha": hashlib.sha256(real.tobytes()).hexdigest()[:16],
           "gen_sha": hashlib.sha256(gen.tobytes()).hexdigest()[:16], "cases": []}

    def one(ori, fake, seed, tag):
        np.random.seed(seed); torch.manual_seed(seed)
        model = TS2Vec(input_dims=ori.shape[-1], device="cpu", batch_size=8, lr=0.001, output_dims=320,
                       max_train_length=3000)
        losses = model.fit(ori.copy(), verbose=False)
        r1 = model.encode(ori.copy(), encoding_window="full_series")
        r2 = model.encode(fake.copy(), encoding_window="full_series")
        idx = np.random.permutation(ori.shape[0])
        fid = float(calculate_fid_score(r1[idx], r2[idx]))
        case = {"tag": tag, "seed": seed, "fid": fid, "epoch_losses": [float(v) for v in losses],
                "n_iters": int(model.n_iters), "repr_real_sum": float(r1.astype(np.float64).sum()),
                "repr_gen_sum": float(r2.astype(np.float64).sum()),
                "repr_real_head": [float(v) for v in r1[0, :8]], "repr_gen_head": [float(v) for v in r2[-1, :8]]}
        out["cases"].append(case)
        print("g9", tag, seed, fid, losses[-1], flush=True)

    one(real, gen, 0, "dense")
    one(real, gen, 3, "dense")
    ragged_r, ragged_g = real.copy(), gen.copy()
    ragged_r[:, :3] = np.nan; ragged_r[::2, -2:] = np.nan   # a common all-NaN head, a tail on every other series
    ragged_g[1::3, 5] = np.nan
    one(ragged_r, ragged_g, 1, "ragged")
    with open(os.path.join(HERE, "g9_context_fid.json"), "w") as f:
        json.dump(out, f, indent=1)


def delay_fuzz_cases(n=48, seed=2024):
    """Random delay-embedding geometries (L, d, E, C) that the reference itself accepts (it raises when the columns
    outnumber the embedding, img_transformations.py:86-98)."""
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        E = int(rng.choice([2, 3, 4, 8, 12, 16, 24, 32]))
        d = int(rng.integers(1, E + 5))
        L = int(rng.integers(E, E + d * (E - 1) + 1))
        C = int(rng.choice([1, 2, 3, 7]))
        try:
            DelayEmbedder("cpu", L, d, E, 1, C).ts_to_img(torch.zeros(1, L, C))
        except (IndexError, RuntimeError):
            continue
        cases.append((L, d, E, C))
    return cases


def g10_delay_fuzz():
    """ts_to_img / mask / img_to_ts (fp32 and fp64 input) of the REFERENCE on 48 random geometries, incl. delay >
    embedding (uncovered time steps), partial last columns, one-column images and L == E."""
    out, meta = {}, []
    for i, (L, d, E, C) in enumerate(delay_fuzz_cases()):
        rng = np.random.default_rng(7000 + i)
        B = 2
        ts = rng.random((B, L, C), dtype=np.float32)
        ts[0, rng.permutation(L)[: max(1, int(0.3 * L))], :] = np.nan
        emb = DelayEmbedder("cpu", L, d, E, B, C)
        img = emb.ts_to_img(torch.from_numpy(ts))
        S = img.shape[-1]
        t = f"c{i}"
        out[t + "_ts"] = bits(ts)
        out[t + "_img"] = bits(img.numpy())
        out[t + "_shape"] = np.array(emb.img_shape, dtype=np.int64)
        for dt, name in ((np.float32, "f32"), (np.float64, "f64")):
            x = rng.standard_normal((B, C, S, S)).astype(dt)
            out[t + f"_in_{name}"] = bits(x)
            out[t + f"_out_{name}"] = bits(emb.img_to_ts(torch.from_numpy(x)).numpy())
        meta.append(dict(tag=t, L=L, d=d, E=E, C=C, B=B, S=int(S)))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g10_delay_fuzz.npz"), **out)
    print("g10_delay_fuzz", len(meta), "cases", sorted({(m["L"], m["d"], m["E"]) for m in meta})[:6], "...")


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "g10":
        g10_delay_fuzz()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        g9_context_fid()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        g8_metrics()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g7":
        g7_fullsize(sys.argv[2:] or None)
        sys.exit(0)
    g1_delay()
    g2_init()
    g3_layers()
    g4_net()
    g6_ema()
    g7_fullsize()
    g8_metrics()
    g9_context_fid()
    g10_delay_fuzz()
    for f in sorted(os.listdir(HERE)):
        if f.endswith((".npz", ".json")):
            print(f, os.path.getsize(os.path.join(HERE, f)))
