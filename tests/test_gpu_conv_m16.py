"""conv3x3_h2_m16_kernel -- the large 3x3 convolutions (models/networks.py:87 via conv0 / conv1 of every UNetBlock) on
v_mfma_f32_16x16x32_f16 with alternating compute / staging wave groups.  Taken from 256 workgroup tiles (256 positions
x 128 output channels) on: the bench batches, not the batch-2 goldens.  Checked here, through the C ABI:

* against the fp64 convolution of the very operands the kernel multiplies (the split pairs), 5e-6 -- fp32-grade;
* against the 32x32x16 kernel it replaces (`I2R_H2_M16=0`, read per call), 2e-6;
* on every image width it takes (2, 4, 8, 16, 32: each has its own tap-border handling), ragged last tiles, output
  channel counts that do not fill the last tile, input / residual upsampled on the fly, un-split fp32 input;
* inside guard bands: nothing outside the output is written, no input is modified."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GUARD = 1 << 14
SENT = 777.0

# (B, Cin, Cout, H, split input, residual, up_in, up_res)
CASES = [(512, 128, 128, 16, True, True, False, False),
         (1027, 256, 256, 8, True, True, False, False),
         (130, 128, 256, 32, True, True, False, False),
         (2048, 256, 256, 4, True, True, False, False),
         (520, 96, 160, 16, True, True, False, False),
         (512, 256, 256, 16, True, True, True, True),
         (128, 128, 128, 32, True, False, True, False),
         (1024, 256, 256, 8, True, True, True, True),
         (512, 128, 256, 16, False, True, False, False),
         (8200, 512, 512, 2, True, True, False, False),
         (2048, 512, 512, 2, True, True, False, False),    # 2x2 images are taken from 128 tiles on (a half batch)
         (2050, 1024, 512, 2, True, False, False, False)]


def _padded(t):
    n = (t.numel() + 63) // 64 * 64
    big = torch.full((n + 2 * GUARD,), SENT, dtype=torch.float32, device=DEV)
    view = big[GUARD: GUARD + t.numel()].view(t.shape)
    view.copy_(t)
    return view, big, n


def _guards_ok(big, n):
    return bool((big[:GUARD] == SENT).all()) and bool((big[GUARD + n:] == SENT).all())


@pytest.fixture
def m16_env():
    prev = os.environ.get("I2R_H2_M16")
    yield
    if prev is None:
        os.environ.pop("I2R_H2_M16", None)
    else:
        os.environ["I2R_H2_M16"] = prev


@pytest.mark.parametrize("B,Cin,Cout,H,split,has_res,up_in,up_res", CASES)
def test_m16_kernel_against_fp64_and_the_kernel_it_replaces(m16_env, B, Cin, Cout, H, split, has_res, up_in, up_res):
    from imageni2r_amd import ops
    gen = torch.Generator().manual_seed(B + Cin + Cout + H)
    Hin = H // 2 if up_in else H
    x = torch.randn(B, Cin, Hin, Hin, generator=gen) * 2.0 + 0.5
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=gen)
    Hr = H // 2 if up_res else H
    res = torch.randn(B, Cout, Hr, Hr, generator=gen) if has_res else None
    if split:
        xin, _, _ = ops.groupnorm_fwd(x.to(DEV), torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), 32, 1e-5,
                                      act=True, split_out=True)
        words = xin.view(torch.int32)
        xv = ((words & 0xFFFF).to(torch.int16).view(torch.float16).double()
              + ((words >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double())
    else:
        xin = x.to(DEV)
        xv = xin.double()
    wp = ops.conv2d_pack_h2(w.to(DEV))
    xs, xs_big, xs_n = _padded(xin)
    out, out_big, out_n = _padded(torch.zeros(B, Cout, H, H))
    rs = rs_big = None
    if has_res:
        rs, rs_big, rs_n = _padded(res.to(DEV))
    bd = b.to(DEV)

    def run(o):
        return ops.conv2d_fwd_h2(xs, tuple(w.shape), wp, bd, rs, x_split=split, out=o, up_in=up_in, up_res=up_res)

    os.environ["I2R_H2_M16"] = "1"
    y = run(out)
    torch.cuda.synchronize()
    assert _guards_ok(out_big, out_n) and _guards_ok(xs_big, xs_n) and (rs_big is None or _guards_ok(rs_big, rs_n))
    assert torch.equal(xs.view(torch.int32), xin.view(torch.int32))
    os.environ["I2R_H2_M16"] = "0"
    y_old = run(None)
    pick = sorted({0, 1, B // 2, B - 1})
    xr = xv[pick]
    if up_in:
        xr = xr.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = torch.nn.functional.conv2d(xr, w.double().to(DEV), b.double().to(DEV), padding=1)
    if has_res:
        rr = res[pick].double().to(DEV)
        ref = ref + (rr.repeat_interleave(2, 2).repeat_interleave(2, 3) if up_res else rr)
    err = float((y[pick].double() - ref).abs().max() / ref.abs().max())
    err_old = float((y.double() - y_old.double()).abs().max() / y_old.double().abs().max())
    assert err < 5e-6, err
    assert 0.0 < err_old < 2e-6, err_old   # (> 0: the two kernels sum in different orders -- equal bits would mean
    #                                         the switch did not select the new kernel)


def test_m16_kernel_is_not_taken_where_it_is_not_built(m16_env):
    """Below 256 tiles, with Cin not a multiple of 32, or for <= 64 output channels the launch stays on the 32x32x16
    kernels: the switch then changes nothing, bit for bit."""
    from imageni2r_amd import ops
    for (B, Cin, Cout, H) in ((8, 128, 128, 16), (512, 48, 128, 16), (512, 128, 64, 16)):
        gen = torch.Generator().manual_seed(B + Cin)
        x = torch.randn(B, Cin, H, H, generator=gen).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9)).to(DEV)
        wp = ops.conv2d_pack_h2(w)
        os.environ["I2R_H2_M16"] = "1"
        a = ops.conv2d_fwd_h2(x, tuple(w.shape), wp)
        os.environ["I2R_H2_M16"] = "0"
        b = ops.conv2d_fwd_h2(x, tuple(w.shape), wp)
        assert torch.equal(a, b), (B, Cin, Cout, H)
