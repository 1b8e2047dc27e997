"""Data gradient of the convolutions on the f16x2-split kernel (i2r_conv2d_dgrad_h2): the forward kernel on transposed,
tap-flipped split weights with dy scaled by a per-tensor power of two.  Gradients are tiny (1e-8 .. 1e-3) and span
orders of magnitude inside one tensor: fp32-grade accuracy must hold regardless."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from imageni2r_amd import ops as o
    return o


@pytest.mark.parametrize("mag", [1.0, 1e-3, 1e-7, 3e-12])
def test_absmax_pow2_scale(ops, mag):
    x = torch.randn(5, 7, 33, device=DEV) * mag
    x[2, 3, 5] = -3.7 * mag * 4
    s = ops.absmax_pow2_scale(x).cpu().numpy()
    peak = float(x.abs().max())
    assert s[0] * s[1] == 1.0 and np.log2(s[0]) == np.round(np.log2(s[0]))  # exact power of two and its inverse
    assert 2.0 ** 13 <= peak * s[0] < 2.0 ** 14
    assert ops._scale_scratch(x.device).tolist() == [0, 0]                   # scratch left zero for the next call
    for bad in (0.0, float("inf"), float("nan")):
        y = torch.zeros(100, device=DEV)
        if bad != 0.0:
            y[17] = bad
        assert ops.absmax_pow2_scale(y).cpu().tolist() == [1.0, 1.0]


def test_absmax_pow2_scale_back_to_back_large_and_small(ops):
    """Many calls in a row on one stream sharing the scratch words: tensors from one element to 17 MB (128
    workgroups), unaligned starts, the peak in the first / last element -- every scale right, the scratch words zero
    at the end."""
    g = torch.Generator(device=DEV).manual_seed(5)
    sizes = [1, 3, 64, 1000, 4097, 262144, 1 << 20, (1 << 22) + 5, 1 << 22]
    xs, want = [], []
    for rep in range(3):
        for i, n in enumerate(sizes):
            buf = torch.randn(n + 1, device=DEV, generator=g) * 10.0 ** (-((i + rep) % 9))
            x = buf[1:] if (i + rep) % 2 else buf[:n]
            x = x.contiguous() if not x.is_contiguous() else x
            x[(0 if rep == 0 else n - 1)] *= 37.0
            xs.append(x)
    scales = [ops.absmax_pow2_scale(x) for x in xs]   # no synchronisation in between
    torch.cuda.synchronize()
    for x, s in zip(xs, scales):
        peak = float(x.abs().max())
        s0, s1 = s.tolist()
        assert s0 * s1 == 1.0 and 2.0 ** 13 <= peak * s0 < 2.0 ** 14, (x.numel(), peak, s0)
    assert ops._scale_scratch(xs[0].device).tolist() == [0, 0]


@pytest.mark.parametrize("B,Cin,Cout,H,k", [(4, 64, 64, 16, 3), (2, 128, 256, 8, 1), (32, 128, 128, 16, 3),
                                             (3, 96, 40, 4, 3), (2, 256, 128, 32, 3), (5, 12, 128, 16, 3),
                                             (2, 1024, 512, 2, 3), (7, 384, 128, 16, 1), (3, 24, 20, 6, 3)])
@pytest.mark.parametrize("mag", [1e-2, 1e-7])
def test_dgrad_h2_vs_fp64(ops, B, Cin, Cout, H, k, mag):
    gen = torch.Generator().manual_seed(B * Cin + Cout + H)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    dy = torch.randn(B, Cout, H, H, generator=gen) * mag
    dy *= 10.0 ** (-3.0 * torch.rand(B, 1, 1, 1, generator=gen))            # per-sample loss weights span 1000x
    ref = torch.nn.grad.conv2d_input((B, Cin, H, H), w.double(), dy.double(), padding=k // 2)
    wd, dyd = w.to(DEV), dy.to(DEV)
    dx = ops.conv2d_dgrad_h2(dyd, tuple(w.shape), ops.conv2d_pack_h2_dgrad(wd))
    assert tuple(dx.shape) == (B, Cin, H, H)
    assert_close(dx.cpu().numpy(), ref.numpy(), 5e-6)
    dx32 = ops.conv2d_dgrad(dyd, wd)                                         # the exact fp32-MFMA kernel it replaces
    assert_close(dx.cpu().numpy(), dx32.cpu().numpy(), 5e-6)
    # per-sample accuracy: the weakest-weighted sample is held to the bar on ITS OWN scale (1e-4: the path's bar)
    for b in range(B):
        assert rel_err(dx[b].cpu().numpy(), ref[b].numpy()) < 1e-4, b


def test_forward_kernel_is_unchanged_by_the_scale_hook(ops):
    """The un-split forward path multiplies by an input scale of exactly 1 when none is given: same bits as the
    pre-split path fed the identical values (GroupNorm split_out), and fp32-grade against fp64."""
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(3, 128, 16, 16, generator=gen)
    w = torch.randn(128, 128, 3, 3, generator=gen) / 34
    y = ops.conv2d_fwd_h2(x.to(DEV), tuple(w.shape), ops.conv2d_pack_h2(w.to(DEV)))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert_close(y.cpu().numpy(), ref.numpy(), 2e-6)
