"""conv1x1_h2_ws_kernel -- the residual-free 1x1 convs with Cin = 128 / 256 (the qkv convs of the attention blocks,
models/networks.py:181) weight-stationary on v_mfma_f32_16x16x32_f16, decoupled waves, 8-byte pair stores straight from
the accumulators.  Taken from 32768 positions on.  Checked through the C ABI:

* against the fp64 product of the very operands the kernel multiplies, every sample, 5e-6 -- fp32-grade;
* against the tiled kernel (`I2R_H2_WS=0`, `I2R_H2_M16_K1=0`, `I2R_H2_PERSIST=0`), 2e-6;
* split-pair (qkv) and plain output, both contraction lengths, ragged position counts, Cout that does not fill its
  last tile, un-split fp32 input, and a bias (the first form packed the bias add into a v_pk_fma_f32 that went wrong);
* inside guard bands: nothing outside the output is written;
* not taken with a residual / other channel counts: the switch then changes nothing, bit for bit."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GUARD = 1 << 14
SENT = 777.0
QSCALE = float(np.float32(0.18033688011112042))   # I2R_QSCALE (csrc/i2r_common.h)
SWITCHES = ("I2R_H2_WS", "I2R_H2_M16_K1", "I2R_H2_PERSIST")

# (B, Cin, Cout, H, split input, qkv output)
CASES = [(1024, 256, 768, 8, True, True),
         (2048, 128, 384, 16, True, True),
         (1027, 256, 768, 8, True, True),
         (4100, 256, 768, 4, True, True),
         (1024, 256, 200, 8, True, False),
         (300, 128, 130, 16, False, False),
         (70, 128, 384, 32, True, True),
         (1100, 256, 256, 8, False, False),
         (2048, 256, 768, 4, True, True)]    # exactly 32768 positions: the smallest launch that takes it


@pytest.fixture
def env():
    prev = {k: os.environ.get(k) for k in SWITCHES}
    yield
    for k, v in prev.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _pairs(t):
    words = t.view(torch.int32)
    return ((words & 0xFFFF).to(torch.int16).view(torch.float16).double()
            + ((words >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double())


def _set(ws):
    os.environ["I2R_H2_WS"] = "1" if ws else "0"
    os.environ["I2R_H2_M16_K1"] = "1" if ws else "0"
    os.environ["I2R_H2_PERSIST"] = "1" if ws else "0"


@pytest.mark.parametrize("B,Cin,Cout,H,split,qkv", CASES)
def test_ws_kernel_against_fp64_and_the_tiled_kernel(env, B, Cin, Cout, H, split, qkv):
    from imageni2r_amd import ops
    gen = torch.Generator().manual_seed(B + Cin + Cout + H)
    x = torch.randn(B, Cin, H, H, generator=gen) * 2.0 + 0.5
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)
    b = torch.randn(Cout, generator=gen)
    if split:
        xin = ops.groupnorm_fwd(x.to(DEV), torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), 32, 1e-5,
                                act=False, split_out=True)[0]
        xv = _pairs(xin)
    else:
        xin = x.to(DEV)
        xv = xin.double()
    wp = ops.conv2d_pack_h2(w.to(DEV))
    bd = b.to(DEV)
    n = (B * Cout * H * H + 63) // 64 * 64
    big = torch.full((n + 2 * GUARD,), SENT, dtype=torch.float32, device=DEV)
    out = big[GUARD: GUARD + B * Cout * H * H].view(B, Cout, H, H)
    _set(True)
    y = ops.conv2d_fwd_h2(xin, tuple(w.shape), wp, bd, None, x_split=split, out=out, qkv_out=qkv)
    torch.cuda.synchronize()
    assert bool((big[:GUARD] == SENT).all()) and bool((big[GUARD + n:] == SENT).all())
    _set(False)
    y_old = ops.conv2d_fwd_h2(xin, tuple(w.shape), wp, bd, None, x_split=split, qkv_out=qkv)
    ref = torch.einsum("oc,bchw->bohw", w[:, :, 0, 0].double().to(DEV), xv) + b.double().to(DEV).view(1, -1, 1, 1)
    if qkv:
        yd, yod = _pairs(y), _pairs(y_old)
        ref[:, 0::3] *= QSCALE
    else:
        yd, yod = y.double(), y_old.double()
    err = float((yd - ref).abs().max() / ref.abs().max())
    err_old = float((yd - yod).abs().max() / yod.abs().max())
    assert err < 5e-6, err
    assert 0.0 < err_old < 2e-6, err_old   # (> 0: another summation order -- equal bits would mean the switch did nothing)


def test_ws_kernel_is_not_taken_where_it_is_not_built(env):
    from imageni2r_amd import ops
    # (B, Cin, Cout, H, residual)
    for (B, Cin, Cout, H, has_res) in ((1024, 256, 256, 8, True), (1024, 192, 384, 8, False), (64, 256, 768, 8, False),
                                       (1024, 256, 64, 8, False)):
        gen = torch.Generator().manual_seed(B + Cin)
        x = torch.randn(B, Cin, H, H, generator=gen).to(DEV)
        w = (torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)).to(DEV)
        r = torch.randn(B, Cout, H, H, generator=gen).to(DEV) if has_res else None
        wp = ops.conv2d_pack_h2(w)
        os.environ["I2R_H2_WS"] = "1"
        a = ops.conv2d_fwd_h2(x, tuple(w.shape), wp, None, r)
        os.environ["I2R_H2_WS"] = "0"
        b = ops.conv2d_fwd_h2(x, tuple(w.shape), wp, None, r)
        assert torch.equal(a, b), (B, Cin, Cout, H, has_res)
