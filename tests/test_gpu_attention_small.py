"""attn_fwd_small_kernel -- the fp32 attention of the levels with at most 32 tokens (the 4x4, 2x2 and 1x1 images of
the bottom of the UNet, models/networks.py:181-186).  Token counts that are a multiple of 4 take the form with
16-byte accesses (T = 4 and 16 with every loop unrolled, the others with run-time trip counts), the rest the
one-element form; all are checked through the product's entry point against the fp64 softmax attention of the same
operands (2e-6 of the largest value; log-sum-exp rows 1e-5), inside guard bands, with many more (sample, head) pairs
than compute units so that every workgroup slot is used several times."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GUARD, SENT = 1 << 14, 4321.0


@pytest.mark.parametrize("H,W", [(1, 1), (2, 2), (2, 4), (3, 3), (3, 4), (4, 4), (4, 5), (5, 5), (4, 7), (4, 8)])
@pytest.mark.parametrize("B,heads", [(700, 2), (3, 4)])
def test_small_attention_against_fp64(B, heads, H, W):
    from imageni2r_amd import ops
    T, C = H * W, 64 * heads
    gen = torch.Generator().manual_seed(T + B)
    host = torch.randn(B, 3 * C, H, W, generator=gen) * 1.5
    big = torch.full((host.numel() + 2 * GUARD,), SENT, device=DEV)
    qkv = big[GUARD: GUARD + host.numel()].view(host.shape)
    qkv.copy_(host)
    a, lse = ops.attention_fwd(qkv, heads, save_lse=True)
    a2, _ = ops.attention_fwd(qkv, heads)   # (without the log-sum-exp rows: the sampler's call)
    torch.cuda.synchronize()
    assert bool((big[:GUARD] == SENT).all()) and bool((big[GUARD + host.numel():] == SENT).all())
    assert torch.equal(qkv.cpu(), host)
    assert torch.equal(a, a2)
    r = qkv.double().view(B, heads, 64, 3, T)
    s = torch.einsum("bhdq,bhdk->bhqk", r[:, :, :, 0], r[:, :, :, 1]) * 0.125
    ref = torch.einsum("bhqk,bhdk->bhdq", torch.softmax(s, dim=-1), r[:, :, :, 2]).reshape(B, C, H, W)
    err = float((a.double() - ref).abs().max() / ref.abs().max())
    err_lse = float((lse.double().view(B * heads, T) - torch.logsumexp(s, dim=-1).reshape(B * heads, T)).abs().max())
    assert err < 2e-6, err
    assert err_lse < 1e-5, err_lse


def test_small_attention_is_reproducible_beside_another_streams_kernels():
    """The T = 16 form is fully unrolled; the next item's accumulator clear sat right behind a 16-byte store of the same
    registers, and on gfx950 nothing keeps a VALU write away from the data of a store that is issued but has not read
    its registers yet.  Alone the kernel was always right; with the vector-memory path backed up by another stream's
    f16x2 attention launches on the same CUs a few workgroups in a thousand stored other values (errors of order 1) --
    which made the two-stream sampler irreproducible.  Same bits beside the loop as alone, for every token count."""
    from imageni2r_amd import ops
    import imageni2r_amd
    imageni2r_amd.set_conv_math("f16x2")
    g = torch.Generator(device=DEV).manual_seed(3)
    noise = torch.randn(2048, 384, 16, 16, device=DEV, generator=g)
    side = torch.cuda.Stream()
    for (H, W, heads, lse) in ((4, 4, 4, False), (4, 4, 4, True), (2, 4, 4, False), (2, 2, 8, False), (3, 3, 4, False)):
        x = torch.randn(2048, 192 * heads, H, W, device=DEV, generator=g)
        ref = ops.attention_fwd(x, heads, save_lse=lse)[0].clone()
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(40):
                ops.attention_fwd(noise, 2)
        outs = [ops.attention_fwd(x, heads, save_lse=lse)[0] for _ in range(6)]
        torch.cuda.synchronize()
        assert all(torch.equal(o, ref) for o in outs), (H, W, heads, lse)
