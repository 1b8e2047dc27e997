"""conv1x1_h2_m16_kernel -- the long-contraction / residual-adding 1x1 convs (the qkv and proj convs of the attention
blocks, models/networks.py:181-186, and the decoder's skip convs over a channel concat, :458) on
v_mfma_f32_16x16x32_f16, as one software pipeline over all of a workgroup's tiles.  Taken from 256 workgroup tiles
(256 positions x 128 output channels) on, Cin a multiple of 64.  Checked through the C ABI:

* against the fp64 product of the very operands the kernel multiplies, every sample, 5e-6 -- fp32-grade;
* against the kernels it replaces (`I2R_H2_M16_K1=0`, read per call), 2e-6;
* split-pair (qkv) output, residual, channel concat with both sources, ragged last tile, Cout that does not fill its
  last tile, an odd number of 64-channel chunks, un-split fp32 input;
* inside guard bands: nothing outside the output is written."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GUARD = 1 << 14
SENT = 777.0
QSCALE = float(np.float32(0.18033688011112042))   # I2R_QSCALE (csrc/i2r_common.h)

# (B, Cin, Cin2 [second source of a concat], Cout, H, split input, residual, qkv output)
CASES = [(1024, 256, 0, 768, 8, True, False, True),
         (1024, 256, 0, 256, 8, True, True, False),
         (512, 128, 0, 128, 16, True, True, False),
         (1027, 256, 0, 768, 8, True, False, True),
         (4096, 256, 0, 256, 4, True, True, False),
         (1024, 128, 128, 128, 16, False, False, False),
         (1024, 256, 128, 128, 16, False, False, False),
         (520, 192, 0, 160, 16, True, True, False),
         (512, 512, 0, 384, 8, False, True, False)]


def _padded(t):
    n = (t.numel() + 63) // 64 * 64
    big = torch.full((n + 2 * GUARD,), SENT, dtype=torch.float32, device=DEV)
    view = big[GUARD: GUARD + t.numel()].view(t.shape)
    view.copy_(t)
    return view, big, n


def _guards_ok(big, n):
    return bool((big[:GUARD] == SENT).all()) and bool((big[GUARD + n:] == SENT).all())


def _pairs(t):
    words = t.view(torch.int32)
    return ((words & 0xFFFF).to(torch.int16).view(torch.float16).double()
            + ((words >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double())


@pytest.fixture
def k1_env():
    prev, prev_ws = os.environ.get("I2R_H2_M16_K1"), os.environ.get("I2R_H2_WS")
    os.environ["I2R_H2_WS"] = "0"   # (the weight-stationary kernel would take the residual-free cases first)
    yield
    for name, val in (("I2R_H2_M16_K1", prev), ("I2R_H2_WS", prev_ws)):
        if val is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = val


@pytest.mark.parametrize("B,Cin,Cin2,Cout,H,split,has_res,qkv", CASES)
def test_k1_m16_kernel_against_fp64_and_the_kernels_it_replaces(k1_env, B, Cin, Cin2, Cout, H, split, has_res, qkv):
    from imageni2r_amd import ops
    gen = torch.Generator().manual_seed(B + Cin + Cout + H)
    Ct = Cin + Cin2
    x = torch.randn(B, Ct, H, H, generator=gen) * 2.0 + 0.5
    w = torch.randn(Cout, Ct, 1, 1, generator=gen) / np.sqrt(Ct)
    b = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, H, generator=gen) if has_res else None
    if split:
        xin, _, _ = ops.groupnorm_fwd(x.to(DEV), torch.ones(Ct, device=DEV), torch.zeros(Ct, device=DEV), 32, 1e-5,
                                      act=False, split_out=True)
        xv = _pairs(xin)
    else:
        xin = x.to(DEV)
        xv = xin.double()
    wp = ops.conv2d_pack_h2(w.to(DEV))
    bd = b.to(DEV)
    rs = res.to(DEV) if has_res else None
    out, out_big, out_n = _padded(torch.zeros(B, Cout, H, H))

    def run(o):
        if Cin2:
            return ops.conv1x1_cat_fwd_h2(xin[:, :Cin].contiguous(), xin[:, Cin:].contiguous(), tuple(w.shape), wp, bd, rs)
        return ops.conv2d_fwd_h2(xin, tuple(w.shape), wp, bd, rs, x_split=split, out=o, qkv_out=qkv)

    os.environ["I2R_H2_M16_K1"] = "1"
    y = run(out)
    torch.cuda.synchronize()
    assert Cin2 or _guards_ok(out_big, out_n)
    os.environ["I2R_H2_M16_K1"] = "0"
    y_old = run(None)
    ref = torch.einsum("oc,bchw->bohw", w[:, :, 0, 0].double().to(DEV), xv) + b.double().to(DEV).view(1, -1, 1, 1)
    if has_res:
        ref = ref + res.double().to(DEV)
    if qkv:
        yd, yod = _pairs(y), _pairs(y_old)
        ref[:, 0::3] *= QSCALE
    else:
        yd, yod = y.double(), y_old.double()
    err = float((yd - ref).abs().max() / ref.abs().max())
    err_old = float((yd - yod).abs().max() / yod.abs().max())
    assert err < 5e-6, err
    assert 0.0 < err_old < 2e-6, err_old   # (> 0: another summation order -- equal bits would mean the switch did nothing)


def test_k1_m16_kernel_is_not_taken_where_it_is_not_built(k1_env):
    """Below 256 tiles, with Cin not a multiple of 64, for <= 64 output channels or image widths that are not a multiple
    of 4 the launch stays on the kernels of before: the switch then changes nothing, bit for bit."""
    from imageni2r_amd import ops
    for (B, Cin, Cout, H) in ((8, 256, 256, 8), (1024, 96, 128, 8), (1024, 256, 64, 8), (8192, 512, 512, 2)):
        gen = torch.Generator().manual_seed(B + Cin)
        x = torch.randn(B, Cin, H, H, generator=gen).to(DEV)
        w = (torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)).to(DEV)
        r = torch.randn(B, Cout, H, H, generator=gen).to(DEV)
        wp = ops.conv2d_pack_h2(w)
        os.environ["I2R_H2_M16_K1"] = "1"
        a = ops.conv2d_fwd_h2(x, tuple(w.shape), wp, None, r)
        os.environ["I2R_H2_M16_K1"] = "0"
        b = ops.conv2d_fwd_h2(x, tuple(w.shape), wp, None, r)
        assert torch.equal(a, b), (B, Cin, Cout, H)
