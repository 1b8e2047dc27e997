"""3x3 f16x2 convs with few output channels run on 64- / 32-row Cout tiles (the 64-channel net, every net's output
conv): same fragments and MFMA order per output element as the 128-row tile, so equal bits (I2R_H2_BM128=1 forces
the latter) -- pair-format and plain inputs, residual, ragged positions and channel counts, DPP and gather paths --
and the kernel's own 1e-4 bar against torch in fp64."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# B, Cin, Cout, H, W, pair-format input, residual
CASES = [(512, 128, 12, 16, 16, 1, 0), (256, 64, 64, 8, 8, 1, 1), (130, 128, 64, 8, 8, 1, 0), (64, 128, 28, 32, 32, 0, 0),
         (300, 100, 33, 12, 12, 0, 1), (32, 128, 5, 8, 8, 1, 0), (200, 64, 20, 6, 10, 0, 1), (70, 14, 64, 8, 8, 0, 0),
         (3, 96, 48, 4, 4, 1, 1)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_%dto%d_%dx%d_s%d_r%d" % c)
def test_bits_equal_the_128_row_tile(case, monkeypatch):
    from imageni2r_amd import ops
    from tests.helpers import assert_close
    B, Cin, Cout, H, W, sp, rs = case
    g = torch.Generator(device=DEV).manual_seed(B + Cout)
    x0 = torch.randn(B, Cin, H, W, device=DEV, generator=g)
    x = x0
    if sp:
        grp = 32 if Cin % 32 == 0 else 2
        x = ops.groupnorm_fwd(x0, torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), grp, 1e-5,
                              split_out=True)[0]
        x0 = torch.nn.functional.group_norm(x0.double(), grp, eps=1e-5)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device=DEV, generator=g)
    res = torch.randn(B, Cout, H, W, device=DEV, generator=g) if rs else None
    wph = ops.conv2d_pack_h2(w)
    monkeypatch.setenv("I2R_H2_BM128", "1")
    full = ops.conv2d_fwd_h2(x, w.shape, wph, b, res, bool(sp)).clone()
    monkeypatch.delenv("I2R_H2_BM128")
    short = ops.conv2d_fwd_h2(x, w.shape, wph, b, res, bool(sp)).clone()
    assert torch.equal(full.view(torch.int32), short.view(torch.int32))
    ref = torch.nn.functional.conv2d(x0.double(), w.double(), b.double(), padding=1)
    if rs:
        ref = ref + res.double()
    assert_close(short.double().cpu().numpy(), ref.cpu().numpy(), 1e-4, case)
