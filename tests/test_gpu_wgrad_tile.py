"""The 3x3 weight gradient from halo tiles in LDS (conv_wgrad3_tile_kernel, csrc/conv_wgrad.hip) and the 1x1 weight
gradient from wide tiles (conv_wgrad1_tile_kernel) -- the autograd of torch.nn.functional.conv2d w.r.t. weight and bias
(models/networks.py:87) -- against torch autograd in fp64 and against the gather kernel they replace
(I2R_WGRAD_TILE=0), through the C-ABI: whole and partial channel tiles, the image widths they take, one and many
position splits, accumulate mode, the deterministic slab form (bit-identical run to run).
Tolerance 3e-5 max-relative and RMS-relative (fp32 accumulation over up to 32k positions)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 3e-5
CASES = [  # B, Cin, Cout, H, W
    (32, 128, 128, 16, 16),   # B1's first level
    (8, 128, 128, 32, 32),    # D's first level
    (32, 256, 256, 8, 8),
    (3, 40, 130, 32, 32),     # partial ci and co tiles
    (5, 100, 33, 16, 16),
    (37, 64, 96, 8, 8),       # chunk count not a multiple of 4 waves
    (2, 32, 32, 32, 32),      # a single tile
    (9, 384, 128, 16, 16),    # decoder concat widths
    (4, 64, 64, 32, 16),      # H != W (two rows of 16 ... four rows per chunk)
]
CASES_1X1 = [  # B, Cin, Cout, H, W
    (32, 128, 384, 16, 16),   # B1's qkv projection
    (8, 128, 128, 32, 32),    # D's attention output projection
    (32, 256, 768, 8, 8),
    (3, 100, 130, 32, 32),    # partial ci and co tiles
    (5, 64, 128, 8, 8),
    (2, 384, 128, 16, 16),    # two ci tiles, the second half empty
    (4, 512, 256, 16, 16),
    (37, 200, 300, 8, 8),
    (33, 256, 256, 4, 8),     # HW = 32: one chunk per sample
]


def _err(got, ref):
    g, r = got.double(), ref.double()
    e_max = float((g - r).abs().max() / r.abs().max().clamp_min(1e-30))
    e_rms = float((g - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt().clamp_min(1e-30))
    return max(e_max, e_rms)


def _problem(B, Cin, Cout, H, W, seed, ks=3):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g)
    dy = torch.randn(B, Cout, H, W, device="cuda", generator=g) * 0.1
    w = torch.zeros(Cout, Cin, ks, ks, device="cuda", dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, device="cuda", dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), w, b, padding=ks // 2).backward(dy.double())
    return x, dy, w.grad, b.grad


@pytest.mark.parametrize("ks,case", [(3, c) for c in CASES] + [(1, c) for c in CASES_1X1],
                         ids=lambda c: "k%d" % c if isinstance(c, int) else "B%d_%dto%d_%dx%d" % c)
def test_tile_wgrad_against_fp64_and_the_gather_kernel(ks, case, monkeypatch):
    from imageni2r_amd import ops
    B, Cin, Cout, H, W = case
    x, dy, dw_ref, db_ref = _problem(B, Cin, Cout, H, W, 100 + Cin + Cout, ks)
    monkeypatch.setenv("I2R_WGRAD_TILE", "1")
    monkeypatch.setenv("I2R_WGRAD_TILE_MINPOS", "0")
    dw, db = ops.conv2d_wgrad(x, dy, ks)
    assert _err(dw, dw_ref) < TOL and _err(db, db_ref) < TOL, (_err(dw, dw_ref), _err(db, db_ref))
    # without a bias gradient
    dw_nb, none = ops.conv2d_wgrad(x, dy, ks, want_bias=False)
    assert none is None and _err(dw_nb, dw_ref) < TOL
    # accumulate into a slot that already holds a gradient
    slot_w, slot_b = dw_ref.float().clone(), db_ref.float().clone()
    ops.conv2d_wgrad(x, dy, ks, dw=slot_w, db=slot_b, accumulate=True)
    assert _err(slot_w, 2 * dw_ref) < TOL and _err(slot_b, 2 * db_ref) < TOL
    # the deterministic slab form: right, and bit-identical run to run
    ops.set_deterministic(True)
    try:
        d1, b1 = ops.conv2d_wgrad(x, dy, ks)
        d2, b2 = ops.conv2d_wgrad(x, dy, ks)
        slot_w, slot_b = dw_ref.float().clone(), db_ref.float().clone()
        ops.conv2d_wgrad(x, dy, ks, dw=slot_w, db=slot_b, accumulate=True)
    finally:
        ops.set_deterministic(False)
    assert _err(d1, dw_ref) < TOL and _err(b1, db_ref) < TOL
    assert torch.equal(d1, d2) and torch.equal(b1, b2)
    assert _err(slot_w, 2 * dw_ref) < TOL and _err(slot_b, 2 * db_ref) < TOL
    # and the kernel it replaces gives the same gradient to rounding
    monkeypatch.setenv("I2R_WGRAD_TILE", "0")
    dw_old, db_old = ops.conv2d_wgrad(x, dy, ks)
    assert _err(dw, dw_old) < TOL and _err(db, db_old) < TOL


def test_tile_wgrad_declines_what_it_cannot_take(monkeypatch):
    """Widths other than 8 / 16 / 32, narrow layers and misaligned views go to the gather kernel -- and come out right."""
    from imageni2r_amd import ops
    monkeypatch.setenv("I2R_WGRAD_TILE", "1")
    monkeypatch.setenv("I2R_WGRAD_TILE_MINPOS", "0")
    for B, Cin, Cout, H, W in [(6, 64, 64, 12, 12), (16, 3, 64, 16, 16), (64, 128, 128, 4, 4), (4, 64, 16, 16, 16)]:
        x, dy, dw_ref, db_ref = _problem(B, Cin, Cout, H, W, 7)
        dw, db = ops.conv2d_wgrad(x, dy, 3)
        assert _err(dw, dw_ref) < TOL and _err(db, db_ref) < TOL, (B, Cin, Cout, H, W)
    for B, Cin, Cout, H, W in [(40, 128, 128, 4, 4), (9, 32, 128, 16, 16), (9, 128, 64, 16, 16), (7, 128, 128, 5, 8)]:
        x, dy, dw_ref, db_ref = _problem(B, Cin, Cout, H, W, 9, ks=1)
        dw, db = ops.conv2d_wgrad(x, dy, 1)
        assert _err(dw, dw_ref) < TOL and _err(db, db_ref) < TOL, (B, Cin, Cout, H, W)
    # an operand that starts 4 bytes into its allocation
    x, dy, dw_ref, db_ref = _problem(4, 64, 64, 16, 16, 8)
    buf = torch.empty(x.numel() + 1, device="cuda")
    xv = buf[1:].view_as(x).copy_(x)
    assert xv.data_ptr() % 16 != 0 and xv.is_contiguous()
    dw, db = ops.conv2d_wgrad(xv, dy, 3)
    assert _err(dw, dw_ref) < TOL and _err(db, db_ref) < TOL
