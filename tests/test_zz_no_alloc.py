"""The one entry point that allocates nothing: i2r_conv2d_fwd_h2_noalloc (include/i2r_hip.h) against the existing
i2r_conv2d_fwd_h2, through the C ABI.  Same kernels and split-K plan, deterministic reduction => the same bits.
Two calls in a row, the second on a smaller input (the shared scratch keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _vp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _case(lib, B, Cin, H, Cout, k, seed, with_res=True):
    from imageni2r_amd import _lib, ops
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, H, generator=gen).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)).to(DEV)
    bias = torch.randn(Cout, generator=gen).to(DEV)
    res = torch.randn(B, Cout, H, H, generator=gen).to(DEV) if with_res else None
    wp = ops.conv2d_pack_h2(w)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    want = _lib.query("i2r_conv2d_splitk_h2_floats", B, Cin, H, H, Cout, k)
    ws = torch.empty(max(want, 1), dtype=torch.float32, device=DEV)
    ref = torch.full((B, Cout, H, H), float("nan"), device=DEV)
    _lib.call("i2r_conv2d_fwd_h2", _vp(x), _vp(wp), _vp(bias), _vp(res), _vp(ref), _vp(ws) if want else None, want,
              B, Cin, H, H, Cout, k, 0, stream)
    out = torch.full((B, Cout, H, H), float("nan"), device=DEV)
    _lib.call("i2r_conv2d_fwd_h2_noalloc", _vp(x), _vp(wp), _vp(bias), _vp(res), _vp(out), B, Cin, H, H, Cout, k, 0,
              stream)
    torch.cuda.synchronize()
    return want, out, ref, (x, w, bias, res)


def test_noalloc_entry_point_matches_workspace_entry_point_twice_in_a_row():
    from imageni2r_amd import _lib
    lib = _lib.load()
    cap = _lib.query("i2r_conv2d_fwd_h2_noalloc_max_floats")
    assert cap == 4 << 20
    # first call: a split-K shape (decoder bottom, 4x4 maps, 32 splits)
    want1, out1, ref1, (x, w, bias, res) = _case(lib, 8, 512, 4, 512, 3, seed=1)
    assert 0 < want1 <= cap
    assert torch.equal(out1, ref1)
    exact = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), bias.double().cpu(), padding=1) + res.double().cpu()
    assert (out1.double().cpu() - exact).abs().max().item() <= 5e-6 * exact.abs().max().item()
    # second call, smaller input: the scratch still holds the first call's slabs
    want2, out2, ref2, _ = _case(lib, 2, 256, 2, 256, 3, seed=2)
    assert 0 < want2 < want1
    assert torch.equal(out2, ref2)
    # a 1x1 split-K shape without a residual, then a shape that needs no scratch at all
    want3, out3, ref3, _ = _case(lib, 4, 1024, 4, 512, 1, seed=3, with_res=False)
    assert torch.equal(out3, ref3)
    want4, out4, ref4, _ = _case(lib, 256, 128, 16, 128, 3, seed=4)
    assert want4 == 0 and torch.equal(out4, ref4)
    # and the first shape again: same bits as before
    _, out5, _, _ = _case(lib, 8, 512, 4, 512, 3, seed=1)
    assert torch.equal(out5, out1)


def test_noalloc_entry_point_rejects_shapes_beyond_its_scratch():
    from imageni2r_amd import _lib, ops
    cap = _lib.query("i2r_conv2d_fwd_h2_noalloc_max_floats")
    B, Cin, H, Cout, k = 64, 1024, 8, 512, 3
    assert _lib.query("i2r_conv2d_splitk_h2_floats", B, Cin, H, H, Cout, k) > cap
    x = torch.zeros(B, Cin, H, H, device=DEV)
    wp = ops.conv2d_pack_h2(torch.zeros(Cout, Cin, k, k, device=DEV))
    out = torch.zeros(B, Cout, H, H, device=DEV)
    with pytest.raises(_lib.I2RError, match="above the built-in"):
        _lib.call("i2r_conv2d_fwd_h2_noalloc", _vp(x), _vp(wp), None, None, _vp(out), B, Cin, H, H, Cout, k, 0,
                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
