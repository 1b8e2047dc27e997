"""AttentionOp.backward + the einsum backward (models/networks.py:121-126, :184) on the f16x2-split kernels
(i2r_attention_bwd_h2) against torch autograd in fp64 and against the exact fp32 kernels, through the C-ABI: T = 64 ...
1024, one to eight heads, gradients of ordinary size and eight orders of magnitude below fp16's range (the power-of-two
scaling), large scores.  Tolerance 3e-5 max-relative and RMS-relative per q / k / v gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 3e-5


def _err(got, ref):
    g, r = got.double(), ref.double()
    e_max = float((g - r).abs().max() / r.abs().max().clamp_min(1e-300))
    e_rms = float((g - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt().clamp_min(1e-300))
    return max(e_max, e_rms)


def _reference(qkv, da, heads):
    B, C3, T = qkv.shape
    qr = qkv.double().requires_grad_(True)
    q, k, v = qr.reshape(B * heads, 64, 3, T).unbind(2)
    w = torch.einsum("ncq,nck->nqk", q, k / 8.0).softmax(dim=2)
    torch.einsum("nqk,nck->ncq", w, v).reshape(B, C3 // 3, T).backward(da.double())
    return qr.grad


@pytest.mark.parametrize("B,heads,T", [(2, 2, 256), (1, 4, 1024), (3, 1, 128), (1, 2, 384), (2, 8, 256), (5, 4, 64), (2, 2, 192)])
@pytest.mark.parametrize("gscale", [1.0, 3e-7])
@pytest.mark.parametrize("qscale", [1.0, 3.0])
def test_f16x2_backward_against_fp64_and_fp32_kernels(B, heads, T, gscale, qscale):
    from imageni2r_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + heads * 10 + T)
    C = 64 * heads
    qkv = torch.randn(B, 3 * C, T, device="cuda", generator=g) * qscale
    da = torch.randn(B, C, T, device="cuda", generator=g) * gscale
    ref = _reference(qkv, da, heads)
    a, lse = ops.attention_fwd(qkv, heads, save_lse=True)
    assert ops.attn_bwd_h2(T)
    got = ops.attention_bwd(qkv, a, lse, da, heads)
    ops.set_attn_bwd_h2(False)
    try:
        exact = ops.attention_bwd(qkv, a, lse, da, heads)
    finally:
        ops.set_attn_bwd_h2(True)
    r5, g5, e5 = (t.reshape(B * heads, 64, 3, T) for t in (ref, got, exact))
    for sel, name in enumerate("qkv"):
        e = _err(g5[:, :, sel], r5[:, :, sel])
        assert e < TOL, (name, e)
        assert _err(g5[:, :, sel], e5[:, :, sel]) < TOL, name
    assert torch.isfinite(got).all()


def test_dispatch_keeps_short_sequences_and_the_off_switch_on_the_fp32_kernels():
    from imageni2r_amd import ops
    assert not ops.attn_bwd_h2(32) and not ops.attn_bwd_h2(16) and not ops.attn_bwd_h2(96) and ops.attn_bwd_h2(256)
    prev = ops._TRAIN_FWD_H2[0]
    ops.set_train_fwd_h2("0")  # the range fall-back of GraphedTrainStep.check_range: everything exact again
    try:
        assert not ops.attn_bwd_h2(256)
    finally:
        ops.set_train_fwd_h2(prev)


def test_c_abi_rejects_what_the_kernel_does_not_cover():
    from imageni2r_amd import _lib
    x = torch.zeros(3 * 64 * 64, device="cuda")
    p = lambda t: t.data_ptr()  # noqa: E731
    with pytest.raises(_lib.I2RError):
        _lib.call("i2r_attention_bwd_h2", p(x), p(x), p(x), p(x), p(x), p(x), p(x), 1, 1, 32, None)
    with pytest.raises(_lib.I2RError):
        _lib.call("i2r_attention_bwd_h2", p(x), p(x), p(x), p(x), p(x), p(x), None, 1, 1, 128, None)


def test_values_beyond_fp16_range_never_give_a_finite_wrong_gradient():
    """The range contract of the f16x2 kernels: v of magnitude 3e4 drives dS' = P (dP' - delta') 2^-10 past
    65504.  Every gradient element is then either non-finite -- which the optimiser's skip and
    GraphedTrainStep.check_range act on -- or right; none is finite and wrong."""
    from imageni2r_amd import ops
    B, heads, T = 2, 2, 128
    g = torch.Generator(device="cuda").manual_seed(9)
    qkv = torch.randn(B, 3 * 64 * heads, T, device="cuda", generator=g)
    qkv.view(B * heads, 64, 3, T)[:, :, 2] *= 3.0e4
    da = torch.randn(B, 64 * heads, T, device="cuda", generator=g)
    ref = _reference(qkv, da, heads)
    a, lse = ops.attention_fwd(qkv, heads, save_lse=True)
    got = ops.attention_bwd(qkv, a, lse, da, heads)
    assert not torch.isfinite(got).all(), "this input is meant to leave fp16's range"
    fin = torch.isfinite(got)
    scale = ref.abs().reshape(B * heads, 64, 3, T).amax(dim=(0, 1, 3)).repeat(B * heads, 64, 1).unsqueeze(-1)
    err = ((got.double() - ref).abs().reshape(B * heads, 64, 3, T) / scale)[fin.reshape(B * heads, 64, 3, T)]
    assert err.numel() == 0 or float(err.max()) < 1e-3
    # and the exact kernels (forward and backward: what check_range falls back to) take the same input in their stride
    prev = ops._TRAIN_FWD_H2[0]
    ops.set_train_fwd_h2("0")
    try:
        a32, lse32 = ops.attention_fwd(qkv, heads, save_lse=True)
        exact = ops.attention_bwd(qkv, a32, lse32, da, heads)
    finally:
        ops.set_train_fwd_h2(prev)
    assert torch.isfinite(exact).all()
    e5, r5 = exact.reshape(B * heads, 64, 3, T), ref.reshape(B * heads, 64, 3, T)
    for sel in range(3):
        assert _err(e5[:, :, sel], r5[:, :, sel]) < TOL
