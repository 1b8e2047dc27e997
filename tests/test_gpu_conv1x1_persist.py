"""The weight-stationary 1x1 kernel (conv1x1_h2_persist_kernel: Cin <= 128, no residual, >= 65536 positions) against
the tiled f16x2 kernel on the same operands: same fragments and MFMA order, so the outputs are equal bit for bit --
ragged position counts, ragged Cout tiles, one and two K chunks, plain and (hi, lo)-pair inputs, qkv output, and a
position count that is launched in more than one span of 1024 tiles.  Then the tiled kernel's own bar (1e-4 against
torch in fp64) holds for it too."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# B, Cin, Cout, H, W, pair-format input, qkv output
CASES = [(512, 128, 384, 16, 16, 1, 1), (300, 128, 130, 16, 16, 0, 0), (1100, 100, 200, 8, 8, 0, 0),
         (70, 64, 96, 32, 32, 1, 0), (1200, 128, 256, 16, 16, 0, 0), (67, 128, 384, 32, 32, 1, 1)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_%dto%d_%dx%d_s%d_q%d" % c)
def test_bits_equal_the_tiled_kernel(case):
    from imageni2r_amd import ops
    from tests.helpers import assert_close
    B, Cin, Cout, H, W, sp, qo = case
    g = torch.Generator(device=DEV).manual_seed(B + Cout)
    x0 = torch.randn(B, Cin, H, W, device=DEV, generator=g)
    x = x0
    if sp:  # the pair format the GroupNorm hands these convs
        x = ops.groupnorm_fwd(x0, torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), 32, 1e-5,
                              split_out=True)[0]
        x0 = torch.nn.functional.group_norm(x0.double(), 32, eps=1e-5)
    w = torch.randn(Cout, Cin, 1, 1, device=DEV, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, device=DEV, generator=g)
    wph = ops.conv2d_pack_h2(w)
    out = {}
    before = os.environ.get("I2R_H2_PERSIST")
    before_k1, before_ws = os.environ.get("I2R_H2_M16_K1"), os.environ.get("I2R_H2_WS")
    os.environ["I2R_H2_M16_K1"] = "0"   # the TILED kernel is the reference (not the 16x16x32 forms of the larger launches)
    os.environ["I2R_H2_WS"] = "0"
    try:
        for mode in ("0", "1"):
            os.environ["I2R_H2_PERSIST"] = mode
            out[mode] = ops.conv2d_fwd_h2(x, w.shape, wph, b, None, bool(sp), None, bool(qo)).clone()
    finally:
        if before is None:
            os.environ.pop("I2R_H2_PERSIST", None)
        else:
            os.environ["I2R_H2_PERSIST"] = before
        for name, val in (("I2R_H2_M16_K1", before_k1), ("I2R_H2_WS", before_ws)):
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val
    assert torch.equal(out["0"].view(torch.int32), out["1"].view(torch.int32))
    if not qo:
        ref = torch.nn.functional.conv2d(x0.double(), w.double(), b.double())
        assert_close(out["1"].double().cpu().numpy(), ref.cpu().numpy(), 1e-4, case)
