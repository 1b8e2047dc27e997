#!/usr/bin/env python3
"""Stand-alone check of conv3x3_h2_m16_kernel (the 16x16x32-MFMA form of the large 3x3 convs) on PADDED buffers:
activations, packed weights, residual and output sit in the middle of guard-filled allocations; after each launch the
guards must be untouched, the inputs unmodified, and the result must equal the fp64 conv of the split operands (5e-6)
and the 32x32x16 kernel's result (I2R_H2_M16=0; 2e-6).  Every shape is then timed on both kernels.  Run before pytest
whenever the kernel's loader or epilogue indexing changes.  One launch at a time, each synchronised and reported."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from imageni2r_amd import ops  # noqa: E402

dev = "cuda"
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


# (B, Cin, Cout, H, split input, residual, up_in, up_res)
CASES = [(512, 128, 128, 16, True, True, False, False),
         (1027, 256, 256, 8, True, True, False, False),     # ragged last position tile
         (128, 128, 128, 32, True, False, False, False),    # rows of 32: the row-rotate fill
         (130, 256, 128, 32, True, True, False, False),
         (4096, 256, 256, 4, True, True, False, False),     # four image rows per 16-lane row
         (520, 96, 160, 16, True, True, False, False),      # Cout not a multiple of 128, three 32-channel blocks
         (512, 256, 256, 16, True, False, True, False),     # input upsampled x2 on the fly
         (512, 256, 256, 16, True, True, True, True),       # ... and the residual too (decoder up blocks)
         (1024, 256, 256, 8, True, True, True, True),
         (128, 256, 256, 32, True, True, True, True),
         (512, 128, 256, 16, False, True, False, False),    # fp32 input split in the kernel (training forwards)
         (512, 512, 256, 16, True, False, False, False),
         (8192, 512, 512, 2, True, True, False, False),     # 2x2 images: eight image rows per 16-lane row
         (8200, 1024, 512, 2, True, False, False, False),
         (2048, 256, 256, 4, True, True, False, False),     # exactly 256 tiles: one workgroup per CU
         (2050, 512, 256, 4, True, False, True, False)]
worst = 0.0
for (B, Cin, Cout, H, split, has_res, up_in, up_res) in CASES:
    gen = torch.Generator().manual_seed(B + Cin + Cout + H)
    Hin = H // 2 if up_in else H
    x = torch.randn(B, Cin, Hin, Hin, generator=gen) * 2.0 + 0.5
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=gen)
    Hr = H // 2 if up_res else H
    res = torch.randn(B, Cout, Hr, Hr, generator=gen) if has_res else None
    if split:
        xin, _, _ = ops.groupnorm_fwd(x.to(dev), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 32, 1e-5,
                                      act=True, split_out=True)
        words = xin.view(torch.int32)
        xv = ((words & 0xFFFF).to(torch.int16).view(torch.float16).double()
              + ((words >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double())
    else:
        xin = x.to(dev)
        xv = xin.double()
    wp = ops.conv2d_pack_h2(w.to(dev))
    xs, xs_big, xs_n = padded(xin)
    GB = 1 << 18
    wp_big = torch.full((wp.numel() + 2 * GB,), 0xAB, dtype=torch.uint8, device=dev)
    wps = wp_big[GB: GB + wp.numel()]
    wps.copy_(wp)
    rs = rs_big = None
    if has_res:
        rs, rs_big, rs_n = padded(res.to(dev))
    out, out_big, out_n = padded(torch.zeros(B, Cout, H, H))
    bd = b.to(dev)
    torch.cuda.synchronize()

    def run(o):
        return ops.conv2d_fwd_h2(xs, tuple(w.shape), wps, bd, rs, x_split=split, out=o, up_in=up_in, up_res=up_res)

    os.environ["I2R_H2_M16"] = "1"
    y = run(out)
    torch.cuda.synchronize()
    ok = guards_ok(xs_big, xs_n) and guards_ok(out_big, out_n) and (rs_big is None or guards_ok(rs_big, rs_n))
    ok = ok and bool((wp_big[:GB] == 0xAB).all()) and bool((wp_big[GB + wp.numel():] == 0xAB).all())
    ok = ok and torch.equal(wps, wp) and torch.equal(xs.view(torch.int32), xin.view(torch.int32))
    os.environ["I2R_H2_M16"] = "0"
    y_old = run(None)
    torch.cuda.synchronize()
    # fp64 reference on a sample of the batch (first, middle, last samples)
    pick = list(range(B)) if B * H * H * Cin <= (1 << 26) else sorted({0, 1, B // 2, B - 1})
    xr = xv[pick]
    if up_in:
        xr = xr.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = torch.nn.functional.conv2d(xr, w.double().to(dev), b.double().to(dev), padding=1)
    if has_res:
        rr = res[pick].double().to(dev)
        ref = ref + (rr.repeat_interleave(2, 2).repeat_interleave(2, 3) if up_res else rr)
    err = float((y[pick].double() - ref).abs().max() / ref.abs().max())
    err_old64 = float((y_old[pick].double() - ref).abs().max() / ref.abs().max())
    if err > 5e-6 or err_old64 > 5e-6:
        bad = ((y[pick].double() - ref).abs() > 1e-4 * ref.abs().max()).flatten(1).any(1).nonzero().flatten().tolist()
        bad_old = ((y_old[pick].double() - ref).abs() > 1e-4 * ref.abs().max()).flatten(1).any(1).nonzero().flatten()
        print(f"  new kernel wrong in {len(bad)} of {len(pick)} checked samples {bad[:24]}; 32x32x16 kernel wrong in "
              f"{bad_old.numel()} {bad_old.tolist()[:24]} (err {err_old64:.2e})", flush=True)
    err_old = float((y.double() - y_old.double()).abs().max() / y_old.double().abs().max())
    flop = 2.0 * B * H * H * Cout * Cin * 9
    tmp = torch.empty_like(y_old)
    os.environ["I2R_H2_M16"] = "1"
    t_new = timed(lambda: run(tmp))
    os.environ["I2R_H2_M16"] = "0"
    t_old = timed(lambda: run(tmp))
    os.environ["I2R_H2_M16"] = "1"
    worst = max(worst, err)
    print(f"B{B} {Cin}->{Cout} @{H} split={int(split)} res={int(has_res)} up={int(up_in)}{int(up_res)}: err vs fp64 "
          f"{err:.2e}, vs 32x32x16 kernel {err_old:.2e}, guards intact {ok};  32x32x16 {t_old * 1e3:.0f} us "
          f"({flop / t_old * 1e-9:.0f} TF)  16x16x32 {t_new * 1e3:.0f} us ({flop / t_new * 1e-9:.0f} TF)  "
          f"x{t_old / t_new:.3f}", flush=True)
    assert ok and err < 5e-6 and err_old < 2e-6, "FAILED"
    del xs_big, wp_big, out_big, rs_big, y, y_old, tmp
print(f"probe ok, worst error vs fp64 {worst:.2e}")
