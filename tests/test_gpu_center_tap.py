"""3x3 convs on 1x1 images (the bottom level of the 8x8-image configs, models/networks.py:68-90 with padding 1): in
f16x2 inference the layer runs the 1x1 conv of its centre tap -- the other eight taps only meet the zero padding.
Checked at the layer: against the full 3x3 launch (I2R_CENTER_TAP=0; 2e-6) and against fp64 conv2d, with bias,
residual, a downsampling layer (2x2 -> 1x1) and pair-format input; and the weight is re-packed when it changes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def tap_env():
    prev = os.environ.get("I2R_CENTER_TAP")
    yield
    if prev is None:
        os.environ.pop("I2R_CENTER_TAP", None)
    else:
        os.environ["I2R_CENTER_TAP"] = prev


@pytest.mark.parametrize("B,Cin,Cout,down,split,has_res", [(4096, 512, 512, False, True, True),
                                                         (130, 1024, 512, False, True, False),
                                                         (2048, 256, 512, True, False, False),
                                                         (7, 96, 40, False, False, True)])
def test_center_tap_conv_on_1x1_images(tap_env, B, Cin, Cout, down, split, has_res):
    import imageni2r_amd
    from imageni2r_amd import networks, ops
    imageni2r_amd.set_conv_math("f16x2")
    torch.manual_seed(B + Cin)
    layer = networks.Conv2d(Cin, Cout, kernel=3, down=down).to(DEV)
    with torch.no_grad():
        layer.weight.copy_(torch.randn_like(layer.weight) / np.sqrt(Cin * 9))
        layer.bias.copy_(torch.randn_like(layer.bias))
    H = 2 if down else 1
    x = torch.randn(B, Cin, H, H, device=DEV) * 1.5 + 0.3
    res = torch.randn(B, Cout, 1, 1, device=DEV) if has_res else None
    xin, xv = x, x.double()
    if split:   # pair-format input, as GroupNorm hands it to the conv in the sampler
        xin, _, _ = ops.groupnorm_fwd(x, torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), 32, 1e-5, act=True,
                                      split_out=True)
        words = xin.view(torch.int32)
        xv = ((words & 0xFFFF).to(torch.int16).view(torch.float16).double()
              + ((words >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double())
    with torch.no_grad():
        os.environ["I2R_CENTER_TAP"] = "1"
        y = layer(xin, residual=res, x_split=split)
        os.environ["I2R_CENTER_TAP"] = "0"
        y_full = layer(xin, residual=res, x_split=split)
        if down:
            xv = torch.nn.functional.avg_pool2d(xv, 2)
        ref = torch.nn.functional.conv2d(xv, layer.weight.double(), layer.bias.double(), padding=1)
        if has_res:
            ref = ref + res.double()
        assert tuple(y.shape) == (B, Cout, 1, 1)
        assert float((y.double() - ref).abs().max() / ref.abs().max()) < 5e-6
        assert float((y - y_full).abs().max() / y_full.abs().max()) < 2e-6   # (often the same bits: the other taps add zeros)
        assert getattr(layer, "_wphc", None) is not None and layer._wphc.numel() < layer._wph.numel()   # (the centre-tap pack exists)
        # a written weight is packed again
        os.environ["I2R_CENTER_TAP"] = "1"
        layer.weight.mul_(2.0)
        y2 = layer(xin, residual=None, x_split=split)
        ref2 = torch.nn.functional.conv2d(xv, layer.weight.double(), layer.bias.double(), padding=1)
        assert float((y2.double() - ref2).abs().max() / ref2.abs().max()) < 5e-6


@pytest.mark.parametrize("B,Cin,Cin2,Cout,split,has_res", [(2048, 256, 0, 256, True, True), (4096, 512, 0, 768, True, False),
                                                        (130, 256, 256, 256, False, False), (7, 72, 0, 40, False, True),
                                                        (1000, 256, 128, 256, False, False), (2048, 100, 0, 64, False, False)])
def test_1x1_conv_on_1x1_images_vector_loads(B, Cin, Cin2, Cout, split, has_res):
    """On 1x1 images a position's channels are consecutive floats: the tiled 1x1 kernel fetches them with 16-byte
    loads.  Same bits as with dword loads (I2R_H2_HW1_SCALAR=1), fp32-grade against fp64; channel counts that do not fill
    the last 64-channel chunk, a channel concat (both sources), pair-format input, residual."""
    from imageni2r_amd import ops
    import imageni2r_amd
    imageni2r_amd.set_conv_math("f16x2")
    gen = torch.Generator().manual_seed(B + Cin + Cout)
    Ct = Cin + Cin2
    x = (torch.randn(B, Ct, 1, 1, generator=gen) * 1.5 + 0.3).to(DEV)
    w = (torch.randn(Cout, Ct, 1, 1, generator=gen) / np.sqrt(Ct)).to(DEV)
    b = torch.randn(Cout, generator=gen).to(DEV)
    res = torch.randn(B, Cout, 1, 1, generator=gen).to(DEV) if has_res else None
    xin, xv = x, x.double()
    if split:
        xin, _, _ = ops.groupnorm_fwd(x, torch.ones(Ct, device=DEV), torch.zeros(Ct, device=DEV), 32, 1e-5, act=True,
                                      split_out=True)
        words = xin.view(torch.int32)
        xv = ((words & 0xFFFF).to(torch.int16).view(torch.float16).double()
              + ((words >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double())
    wp = ops.conv2d_pack_h2(w)

    def run():
        if Cin2:
            return ops.conv1x1_cat_fwd_h2(xin[:, :Cin].contiguous(), xin[:, Cin:].contiguous(), tuple(w.shape), wp, b, res)
        return ops.conv2d_fwd_h2(xin, tuple(w.shape), wp, b, res, x_split=split)

    prev = os.environ.pop("I2R_H2_HW1_SCALAR", None)
    try:
        y = run()
        os.environ["I2R_H2_HW1_SCALAR"] = "1"
        y_scalar = run()
    finally:
        os.environ.pop("I2R_H2_HW1_SCALAR", None)
        if prev is not None:
            os.environ["I2R_H2_HW1_SCALAR"] = prev
    ref = torch.einsum("oc,bchw->bohw", w[:, :, 0, 0].double(), xv) + b.double().view(1, -1, 1, 1)
    if has_res:
        ref = ref + res.double()
    assert torch.equal(y, y_scalar)
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 5e-6
