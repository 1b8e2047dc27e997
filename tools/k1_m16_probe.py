#!/usr/bin/env python3
"""Stand-alone check of conv1x1_h2_m16_kernel (the 16x16x32-MFMA form of the long-K / residual-adding 1x1 convs) on
PADDED buffers: after each launch the guards must be untouched, the inputs unmodified, and the result must equal the
fp64 product of the split operands on EVERY sample (5e-6) and the kernels it replaces (I2R_H2_M16_K1=0; 2e-6; the qkv
split-pair output is compared after decoding the pairs).  Every shape is then timed both ways."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
SWITCH = sys.argv[1] if len(sys.argv) > 1 else "I2R_H2_M16_K1"   # the kernel switch under test ("1" new, "0" the others)
GUARD = 1 << 16
SENT = 12345.0


def padded(t):
    n = (t.numel() + 63) // 64 * 64
    big = torch.full((n + 2 * GUARD,), SENT, dtype=torch.float32, device=dev)
    view = big[GUARD: GUARD + t.numel()].view(t.shape)
    view.copy_(t)
    return view, big, t.numel()


def guards_ok(big, n):
    return bool((big[:GUARD] == SENT).all()) and bool((big[GUARD + (n + 63) // 64 * 64:] == SENT).all())


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def pairs_to_f64(t):
    words = t.view(torch.int32)
    return ((words & 0xFFFF).to(torch.int16).view(torch.float16).double()
            + ((words >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double())


# (B, Cin, Cin2 [second source of a channel concat, 0 = none], Cout, H, split input, residual, qkv split-pair output)
CASES = [(1024, 256, 0, 768, 8, True, False, True),      # qkv conv of the 256-channel levels
         (1024, 256, 0, 256, 8, True, True, False),      # proj conv + residual
         (512, 128, 0, 128, 16, True, True, False),      # proj conv of the 128-channel levels
         (1027, 256, 0, 768, 8, True, False, True),      # ragged last tile
         (4096, 256, 0, 768, 4, True, False, True),
         (4096, 256, 0, 256, 4, True, True, False),
         (1024, 128, 128, 128, 16, False, False, False),  # decoder skip convs: channel concat, fp32 input
         (1024, 256, 256, 256, 8, False, False, False),
         (1024, 256, 128, 128, 16, False, False, False),
         (2048, 256, 256, 256, 4, False, False, False),
         (256, 256, 0, 768, 16, True, False, True),       # shape D's 16x16 level
         (130, 128, 0, 128, 32, True, True, False),
         (520, 192, 0, 160, 16, True, True, False),       # three chunks (odd), Cout not a multiple of 128
         (512, 512, 0, 384, 8, False, True, False),
         (512, 128, 0, 384, 16, False, False, False),     # (fp32 input: not the persistent kernel's case)
         (2048, 128, 0, 384, 16, True, False, True),      # qkv conv of the 128-channel levels at the bench chunk
         (2048, 128, 0, 128, 16, True, True, False),
         (4096, 256, 0, 768, 8, True, False, True)]
worst = 0.0
for (B, Cin, Cin2, Cout, H, split, has_res, qkv) in CASES:
    gen = torch.Generator().manual_seed(B + Cin + Cout + H)
    Ct = Cin + Cin2
    x = torch.randn(B, Ct, H, H, generator=gen) * 2.0 + 0.5
    w = torch.randn(Cout, Ct, 1, 1, generator=gen) / np.sqrt(Ct)
    b = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, H, generator=gen) if has_res else None
    if split:
        xin, _, _ = ops.groupnorm_fwd(x.to(dev), torch.ones(Ct, device=dev), torch.zeros(Ct, device=dev), 32, 1e-5,
                                      act=False, split_out=True)
        xv = pairs_to_f64(xin)
    else:
        xin = x.to(dev)
        xv = xin.double()
    wp = ops.conv2d_pack_h2(w.to(dev))
    GB = 1 << 18
    wp_big = torch.full((wp.numel() + 2 * GB,), 0xAB, dtype=torch.uint8, device=dev)
    wps = wp_big[GB: GB + wp.numel()]
    wps.copy_(wp)
    if Cin2:
        x1, x1_big, x1_n = padded(xin[:, :Cin].contiguous())
        x2, x2_big, x2_n = padded(xin[:, Cin:].contiguous())
    else:
        x1, x1_big, x1_n = padded(xin)
        x2 = x2_big = None
    rs = rs_big = None
    if has_res:
        rs, rs_big, rs_n = padded(res.to(dev))
    out, out_big, out_n = padded(torch.zeros(B, Cout, H, H))
    bd = b.to(dev)
    torch.cuda.synchronize()

    def run(o):
        if Cin2:
            y_ = ops.conv1x1_cat_fwd_h2(x1, x2, tuple(w.shape), wps, bd, rs)
            if o is not None:
                o.copy_(y_)
                return o
            return y_
        return ops.conv2d_fwd_h2(x1, tuple(w.shape), wps, bd, rs, x_split=split, out=o, qkv_out=qkv)

    os.environ[SWITCH] = "1"
    y = run(out if not Cin2 else None)
    torch.cuda.synchronize()
    ok = guards_ok(x1_big, x1_n) and (x2_big is None or guards_ok(x2_big, x2_n)) and (rs_big is None or guards_ok(rs_big, rs_n))
    ok = ok and (Cin2 or guards_ok(out_big, out_n))
    ok = ok and bool((wp_big[:GB] == 0xAB).all()) and bool((wp_big[GB + wp.numel():] == 0xAB).all()) and torch.equal(wps, wp)
    os.environ[SWITCH] = "0"
    y_old = run(None)
    torch.cuda.synchronize()
    ref = torch.einsum("oc,bchw->bohw", w[:, :, 0, 0].double().to(dev), xv) + b.double().to(dev).view(1, -1, 1, 1)
    if has_res:
        ref = ref + res.double().to(dev)
    if qkv:
        yd, yod = pairs_to_f64(y), pairs_to_f64(y_old)
        # q rows (channel % 3 == 0 in the interleaved (head, d, {q, k, v}) order) carry the softmax scale
        ref = ref.clone()
        ref[:, 0::3] *= float(np.float32(0.18033688011112042))   # I2R_QSCALE (i2r_common.h): log2(e) / sqrt(64)
    else:
        yd, yod = y.double(), y_old.double()
    err = float((yd - ref).abs().max() / ref.abs().max())
    err_old = float((yd - yod).abs().max() / yod.abs().max())
    if err > 5e-6:
        bad = ((yd - ref).abs() > 1e-4 * ref.abs().max()).flatten(1).any(1).nonzero().flatten().tolist()
        print(f"  wrong in {len(bad)} of {B} samples {bad[:24]}", flush=True)
    flop = 2.0 * B * H * H * Cout * Ct
    nbytes = 4.0 * B * H * H * (Ct + Cout * (2 if has_res else 1))
    tmp = torch.empty_like(y_old)
    os.environ[SWITCH] = "1"
    t_new = timed(lambda: run(tmp))
    os.environ[SWITCH] = "0"
    t_old = timed(lambda: run(tmp))
    os.environ[SWITCH] = "1"
    worst = max(worst, err)
    print(f"B{B} {Cin}{'+' + str(Cin2) if Cin2 else ''}->{Cout} @{H} split={int(split)} res={int(has_res)} qkv={int(qkv)}: "
          f"err vs fp64 {err:.2e}, vs old kernel {err_old:.2e}, guards intact {ok};  old {t_old * 1e3:.0f} us "
          f"({flop / t_old * 1e-9:.0f} TF, {nbytes / t_old * 1e-9:.2f} TB/s)  m16 {t_new * 1e3:.0f} us "
          f"({flop / t_new * 1e-9:.0f} TF, {nbytes / t_new * 1e-9:.2f} TB/s)  x{t_old / t_new:.3f}", flush=True)
    assert ok and err < 5e-6 and err_old < 2e-6, "FAILED"
    del x1_big, x2_big, wp_big, out_big, rs_big, y, y_old, tmp
print(f"probe ok, worst error vs fp64 {worst:.2e}")
