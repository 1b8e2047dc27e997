"""GroupNorm forward for MANY SMALL groups (`gn_fwd_small_kernel`, csrc/groupnorm.hip): the form the 4x4 / 2x2 levels
take at inference batches (>= 16384 groups of <= 128 elements -- batch >= 512; 41 launches per denoiser evaluation of
the bench's 4096 series).  Every fused mode of models/networks.py:104-106 as used at :168 (SiLU), :173 (modulation +
SiLU), :182 (plain) and the decoder concat of :458, against the CPU oracle (1e-4, the path's bar; measured ~1e-6) and
bit for bit against the one-group-per-slot kernel it replaces (`I2R_GN_SMALL=0`, read per call)."""
import os

import numpy as np
import pytest
import torch

from oracle import i2r_oracle as O
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (batch, channels, image side): groups of 16 / 32 / 64 / 128 / 256 elements, >= 16384 of them; 513 and 700 leave a ragged
# last slot (group count not a multiple of the 4 groups a lane slot walks, nor of a workgroup's 32 / 64)
SHAPES = [(512, 128, 2), (513, 256, 2), (512, 512, 2), (600, 1024, 2), (700, 128, 4), (512, 256, 4), (1024, 128, 2),
          (513, 128, 8), (600, 512, 4)]   # 256 elements per group: the 64-lane slots


def _small_off():
    class _Ctx:
        def __enter__(self):
            self.prev = os.environ.get("I2R_GN_SMALL")
            os.environ["I2R_GN_SMALL"] = "0"

        def __exit__(self, *a):
            if self.prev is None:
                del os.environ["I2R_GN_SMALL"]
            else:
                os.environ["I2R_GN_SMALL"] = self.prev
    return _Ctx()


def _ref(x, w, b, ss, act):
    y = O.group_norm(x, w, b)
    if ss is not None:
        C = x.shape[1]
        y = ss[:, C:, None, None] + y * (ss[:, :C, None, None] + 1)
    return torch.nn.functional.silu(y) if act else y


@pytest.mark.parametrize("B,C,H", SHAPES)
def test_small_group_forward_all_fused_modes(B, C, H):
    from imageni2r_amd import ops
    n = C // 32 * H * H
    assert n <= 256 and B * 32 >= 16384  # the dispatch condition of gn_launch<false>
    g = torch.Generator().manual_seed(B + C + H)
    x = torch.randn(B, C, H, H, generator=g) * 1.7 + 0.4
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ss1 = torch.randn(1, 2 * C, generator=g) * 0.5   # batch-1 modulation (the sampler's 0-dim sigma)
    ssB = torch.randn(B, 2 * C, generator=g) * 0.5   # per-sample sigma
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    for ss, act in ((None, False), (None, True), (ss1, True), (ssB, True), (ssB, False)):
        ssd = None if ss is None else ss.to(DEV)
        y, mean, rstd = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, ss=ssd, act=act, save_stats=True)
        with _small_off():
            y0, mean0, rstd0 = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, ss=ssd, act=act, save_stats=True)
        assert torch.equal(y, y0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0), (ss is not None, act)
        assert_close(y.cpu().numpy(), _ref(x, w, b, ss, act).numpy(), 1e-4, (B, C, H, act))
        # split-pair output for the f16x2 convs: the same bits as the kernel it replaces, and hi + lo = the fp32 value
        ys, _, _ = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, ss=ssd, act=act, split_out=True)
        with _small_off():
            ys0, _, _ = ops.groupnorm_fwd(xd, wd, bd, 32, 1e-5, ss=ssd, act=act, split_out=True)
        assert torch.equal(ys.view(torch.int32), ys0.view(torch.int32))
        pair = ys.view(torch.float16).view(*ys.shape, 2).float()
        assert_close((pair[..., 0] + pair[..., 1]).cpu().numpy(), y.cpu().numpy(), 2e-6)
    xm = x.view(B, 32, -1).double()
    assert_close(mean.cpu().numpy(), xm.mean(2).reshape(-1).numpy(), 1e-5)
    assert_close(rstd.cpu().numpy(), (1.0 / torch.sqrt(xm.var(2, unbiased=False) + 1e-5)).reshape(-1).numpy(), 1e-5)


@pytest.mark.parametrize("B,C1,C2,H", [(512, 512, 512, 2), (520, 256, 256, 2), (512, 256, 128, 2), (512, 128, 128, 4),
                                       (520, 256, 256, 4), (512, 256, 128, 4)])
def test_small_group_forward_over_the_decoder_concat(B, C1, C2, H):
    """norm0 of a decoder block reads torch.cat([x, skip], 1) in place (networks.py:458 -> :168)."""
    from imageni2r_amd import ops
    C = C1 + C2
    assert C // 32 * H * H <= 256
    g = torch.Generator().manual_seed(C1 + C2 + H)
    x1, x2 = torch.randn(B, C1, H, H, generator=g), torch.randn(B, C2, H, H, generator=g) * 3 - 1
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    a = [t.to(DEV) for t in (x1, x2, w, b)]
    y = ops.groupnorm_cat_fwd(*a, 32, 1e-5, act=True)
    with _small_off():
        y0 = ops.groupnorm_cat_fwd(*a, 32, 1e-5, act=True)
    assert torch.equal(y, y0)
    assert_close(y.cpu().numpy(), _ref(torch.cat([x1, x2], 1), w, b, None, True).numpy(), 1e-4)
    ys = ops.groupnorm_cat_fwd(*a, 32, 1e-5, act=True, split_out=True)
    with _small_off():
        ys0 = ops.groupnorm_cat_fwd(*a, 32, 1e-5, act=True, split_out=True)
    assert torch.equal(ys.view(torch.int32), ys0.view(torch.int32))


def test_small_group_forward_leaves_memory_behind_the_output_alone():
    """Ragged group count: the slots past the last group must not store (the output sits inside a guard band)."""
    from imageni2r_amd import _lib, ops
    B, C, H = 513, 128, 2
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, C, H, H, generator=g).to(DEV)
    w, b = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    n = x.numel()
    buf = torch.full((n + 8192,), 777.0, device=DEV)
    y = buf[4096:4096 + n].view_as(x)
    _lib.call("i2r_groupnorm_fwd_f32", ops._p(x), ops._p(w), ops._p(b), None, 0, ops._p(y), None, None, B, C, H * H, 32,
              1e-5, 1, 0.0, 0, 0, None, 0, ops._stream())
    torch.cuda.synchronize()
    assert bool((buf[:4096] == 777.0).all()) and bool((buf[4096 + n:] == 777.0).all())
    ref, _, _ = ops.groupnorm_fwd(x, w, b, 32, 1e-5, act=True)
    assert torch.equal(y, ref)
