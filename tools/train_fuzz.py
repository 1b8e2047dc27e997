#!/usr/bin/env python3
"""Random-shape check of the TRAINING kernels against torch autograd in fp64: conv forward (fp32 MFMA), data gradient
(fp32 and f16x2 form), weight + bias gradient (atomic and deterministic slab form), GroupNorm forward / backward in its
fused modes (scale-shift, SiLU), attention forward / backward (fp32 and f16x2 kernels, gradients down to 1e-8).

    python tools/train_fuzz.py [cases=200] [seed=0]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"
TOL = 3e-5
worst, bad = {}, 0


def check(kind, got, ref, what):
    global bad
    g, r = got.double(), ref.double()
    scale = float(r.abs().max().clamp_min(1e-30))
    e_max = float((g - r).abs().max()) / scale
    e_rms = float((g - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt().clamp_min(1e-30))
    worst[kind] = max(worst.get(kind, 0.0), e_max, e_rms)
    if not (e_max < TOL and e_rms < TOL):
        bad += 1
        print(f"FAIL {kind}: max {e_max:.2e} rms {e_rms:.2e}  {what}", flush=True)


for it in range(N):
    g = torch.Generator(device=dev).manual_seed(5000 + it)
    kind = it % 3
    if kind == 0:  # ---- conv
        k = int(rng.choice([1, 3]))
        H = int(rng.choice([1, 2, 4, 5, 8, 12, 16, 32]))
        W = H if rng.random() < 0.7 else int(rng.choice([2, 4, 6, 8, 16]))
        B = int(rng.integers(1, 48)) if H * W <= 256 else int(rng.integers(1, 12))
        Cin = int(rng.choice([3, 12, 64, 100, 128, 256, 384, 512]))
        Cout = int(rng.choice([3, 12, 40, 64, 128, 130, 256, 512]))
        what = f"conv k{k} B{B} {Cin}->{Cout} {H}x{W}"
        x = torch.randn(B, Cin, H, W, device=dev, generator=g)
        w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (k * k * Cin) ** 0.5
        b = torch.randn(Cout, device=dev, generator=g)
        dy = torch.randn(B, Cout, H, W, device=dev, generator=g) * float(10.0 ** rng.uniform(-6, 0))
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        ref = torch.nn.functional.conv2d(xr, wr, br, padding=k // 2)
        ref.backward(dy.double())
        check("conv_fwd_f32", ops.conv2d_fwd(x, w, b, None), ref.detach(), what)
        check("conv_dgrad_f32", ops.conv2d_dgrad(dy, w), xr.grad, what)
        check("conv_dgrad_h2", ops.conv2d_dgrad_h2(dy, w.shape, ops.conv2d_pack_h2_dgrad(w)), xr.grad, what)
        dw, db = ops.conv2d_wgrad(x, dy, k)
        check("conv_wgrad", dw, wr.grad, what)
        check("conv_bgrad", db, br.grad, what)
        ops.set_deterministic(True)
        try:
            dw, db = ops.conv2d_wgrad(x, dy, k)
        finally:
            ops.set_deterministic(False)
        check("conv_wgrad_det", dw, wr.grad, what)
        check("conv_bgrad_det", db, br.grad, what)
    elif kind == 1:  # ---- GroupNorm (the reference's group count: min(32, C // 4))
        C = int(rng.choice([64, 96, 128, 256, 384, 512, 768, 1024]))
        H = int(rng.choice([1, 2, 4, 8, 16, 32]))
        B = int(rng.integers(1, 24)) if C * H * H <= 131072 else int(rng.integers(1, 5))
        G = min(32, C // 4)
        mode = int(rng.integers(0, 3))  # 0 plain, 1 SiLU, 2 scale-shift + SiLU
        what = f"gn B{B} C{C} {H}x{H} mode{mode}"
        x = torch.randn(B, C, H, H, device=dev, generator=g) * 2 + 0.5
        w, b = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
        ss = torch.randn(B, 2 * C, device=dev, generator=g) * 0.5 if mode == 2 else None
        dy = torch.randn(B, C, H, H, device=dev, generator=g)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        ref = torch.nn.functional.group_norm(xr, G, wr, br, 1e-5)
        sr = None
        if mode == 2:
            sr = ss.double().requires_grad_(True)
            ref = sr[:, C:, None, None] + ref * (sr[:, :C, None, None] + 1)
        if mode >= 1:
            ref = torch.nn.functional.silu(ref)
        ref.backward(dy.double())
        y, mean, rstd = ops.groupnorm_fwd(x, w, b, G, 1e-5, ss=ss, act=mode >= 1, save_stats=True)
        check("gn_fwd", y, ref.detach(), what)
        dx, dw, db, dss = ops.groupnorm_bwd(x, w, b, ss, mean, rstd, dy, G, mode >= 1)
        check("gn_dx", dx, xr.grad, what)
        check("gn_dgamma", dw, wr.grad, what)
        check("gn_dbeta", db, br.grad, what)
        if mode == 2:
            check("gn_dss", dss, sr.grad, what)
        # the training step's form: bypass gradient added to dx, dgamma / dbeta accumulated into zero-filled slots
        add = torch.randn(B, C, H, H, device=dev, generator=g)
        sg, sb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        prev = ops.slots_zeroed()
        ops.set_slots_zeroed(True)
        try:
            dx2, dw2, db2, _ = ops.groupnorm_bwd(x, w, b, ss, mean, rstd, dy, G, mode >= 1, dgamma=sg, dbeta=sb, add=add)
        finally:
            ops.set_slots_zeroed(prev)
        check("gn_dx_plus_add", dx2, xr.grad + add.double(), what)
        check("gn_dgamma_slot", dw2, wr.grad, what)
        check("gn_dbeta_slot", db2, br.grad, what)
    else:  # ---- attention
        heads = int(rng.choice([1, 2, 4, 8, 16]))
        H = int(rng.choice([1, 2, 4, 8, 16, 32]))
        B = int(rng.integers(1, 9)) if H * H * heads <= 4096 else 1
        C = 64 * heads
        what = f"attn B{B} heads{heads} T{H * H}"
        qkv = torch.randn(B, 3 * C, H, H, device=dev, generator=g) * float(rng.uniform(0.5, 2.5))
        da = torch.randn(B, C, H, H, device=dev, generator=g) * float(10.0 ** rng.uniform(-8, 0))
        qr = qkv.double().requires_grad_(True)
        q, kk, v = qr.reshape(B * heads, 64, 3, H * H).unbind(2)
        wgt = torch.einsum("ncq,nck->nqk", q, kk / 8.0).softmax(dim=2)
        ref = torch.einsum("nqk,nck->ncq", wgt, v).reshape(B, C, H, H)
        ref.backward(da.double())
        a, lse = ops.attention_fwd(qkv, heads, save_lse=True)
        check("attn_fwd_train", a, ref.detach(), what)
        check("attn_bwd", ops.attention_bwd(qkv, a, lse, da, heads), qr.grad, what)  # f16x2 kernels when T % 128 == 0
        ops.set_attn_bwd_h2(False)
        try:
            check("attn_bwd_f32", ops.attention_bwd(qkv, a, lse, da, heads), qr.grad, what)
        finally:
            ops.set_attn_bwd_h2(True)
        if ops.attention_h2_ok(H * H):
            check("attn_fwd_h2", ops.attention_fwd(qkv, heads)[0], ref.detach(), what)
print(f"{N} cases, failures {bad}; worst errors:", {k: f"{v:.1e}" for k, v in sorted(worst.items())})
sys.exit(1 if bad else 0)
