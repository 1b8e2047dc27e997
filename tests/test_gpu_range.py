"""fp16 range contract of the f16x2-split kernels: operands are carried as f16 hi + lo pairs, so
  * up to |x| < 65504 the split is as accurate as anywhere else (22 significand bits): the 1e-4 bar holds for
    un-normalised inputs of 1e3 .. 6e4 through the first 3x3 conv and a 1x1 skip conv (the two places that read the
    raw residual stream, `x_split=False`);
  * at |x| >= 65504 the result is inf / NaN -- never a finite wrong number -- and every caller-facing path notices:
    the sampler re-samples the affected series on the exact fp32 kernels, loss_fn_irregular recomputes on them, and
    the graph-replayed train step skips the poisoned update and then leaves the f16x2 forward.
Reference behaviour to match: fp32 F.conv2d (models/networks.py:87), which has no such limit."""
import warnings

import numpy as np
import pytest
import torch

from oracle import i2r_oracle as O
from tests.helpers import rel_err, rms_rel_err, tiny_args

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from imageni2r_amd import ops as o
    return o


@pytest.mark.parametrize("scale", [1e3, 2e4, 6e4])
@pytest.mark.parametrize("B,Cin,Cout,H,k", [(4, 12, 128, 16, 3), (3, 384, 128, 16, 1), (2, 1024, 512, 2, 1)])
def test_large_unnormalised_activations_keep_the_bar(ops, scale, B, Cin, Cout, H, k):
    gen = torch.Generator().manual_seed(int(scale) + Cin)
    x = (torch.rand(B, Cin, H, H, generator=gen) * 2 - 1) * scale          # |x| up to `scale`, some elements near it
    x.view(-1)[::97] = torch.sign(x.view(-1)[::97]) * scale * 0.999
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, H, generator=gen) * scale
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=k // 2) + res.double()
    y = ops.conv2d_fwd_h2(cu(x.numpy()), tuple(w.shape), ops.conv2d_pack_h2(cu(w.numpy())), cu(b.numpy()),
                          cu(res.numpy()))
    assert torch.isfinite(y).all()
    e, r = rel_err(y.cpu().numpy(), ref.numpy()), rms_rel_err(y.cpu().numpy(), ref.numpy())
    assert e < 5e-6 and r < 5e-6, (e, r)  # far inside the path's 1e-4; same grade as O(1) inputs


def test_at_fp16_max_the_result_is_never_a_finite_wrong_number(ops):
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 8, 8, generator=gen)
    x[0, 3, 2, 5] = 7.0e4     # one element past fp16's 65504
    x[1, 0, 0, 0] = -1.0e6
    w = torch.randn(32, 16, 3, 3, generator=gen) / 12
    y = ops.conv2d_fwd_h2(cu(x.numpy()), tuple(w.shape), ops.conv2d_pack_h2(cu(w.numpy())))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    yc = y.cpu().double()
    touched = torch.zeros_like(ref, dtype=torch.bool)  # outputs whose 3x3 window holds an out-of-range input
    touched[0, :, 1:4, 4:7] = True
    touched[1, :, 0:2, 0:2] = True
    assert not torch.isfinite(yc[touched]).any()                          # loud: inf / NaN, not garbage
    assert rel_err(yc[~touched].numpy(), ref[~touched].numpy()) < 5e-6    # everything else untouched
    # the exact fp32 kernel (what the fallbacks switch to) takes the same input in its stride
    y32 = ops.conv2d_fwd(cu(x.numpy()), cu(w.numpy()))
    assert rel_err(y32.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.fixture()
def tiny():
    from imageni2r_amd import TS2img_Karras
    import imageni2r_amd
    imageni2r_amd.set_conv_math("f16x2")
    torch.manual_seed(0)
    m = TS2img_Karras(tiny_args(device=DEV), DEV)
    O.perturb_zero_init(m.net.state_dict(), seed=1234)
    yield m.to(DEV)
    imageni2r_amd.set_conv_math("f16x2")


def test_sampler_resamples_overflowed_series_on_the_exact_kernels(tiny):
    import imageni2r_amd
    from imageni2r_amd import DiffusionProcess
    m = tiny.eval()
    args = tiny_args(device=DEV)
    gen = torch.Generator().manual_seed(1)
    xT = torch.randn(4, 3, 8, 8, generator=gen)
    xT[2] *= 2.0e5          # c_in(80) * 80 * xT ~ xT: the first conv of series 2 sees ~6e5 >= 65504
    proc = DiffusionProcess(args, m.net, (3, 8, 8))
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        out = proc.sample(xT)
    assert torch.isfinite(out).all()
    imageni2r_amd.set_conv_math("f32")
    try:
        exact = DiffusionProcess(args, m.net, (3, 8, 8), use_graph=False).sample(xT)
    finally:
        imageni2r_amd.set_conv_math("f16x2")
    assert rel_err(out[2].cpu().numpy(), exact[2].cpu().numpy()) < 1e-6   # series 2: the fp32 kernels' own result
    assert rel_err(out.cpu().numpy(), exact.cpu().numpy()) < 1e-4        # the others: f16x2, inside the bar
    proc.range_check = False                                             # the guard off: the overflow is visible
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        raw = proc.sample(xT)
    # (its neighbours in the batch may be non-finite too: where Cin is not a multiple of 16 the first conv's padding
    # channels read the next sample's values against zero weights, and inf * 0 = NaN -- the guard re-samples every
    # non-finite series, whatever made it so)
    assert not torch.isfinite(raw[2]).all()
    with warnings.catch_warnings():                                      # in-range latents: no warning, no re-run
        warnings.simplefilter("error")
        proc.range_check = True
        assert torch.isfinite(proc.sample(xT[:2])).all()


def test_loss_fn_recomputes_on_the_exact_kernels_when_the_f16x2_forward_overflows(tiny):
    from imageni2r_amd import ops
    m = tiny.train()
    before = ops._TRAIN_FWD_H2[0]
    ops.set_train_fwd_h2("1")
    ops.new_pack_epoch()
    try:
        gen = torch.Generator().manual_seed(2)
        x = torch.rand(4, 3, 8, 8, generator=gen).to(DEV) * 3.0e5
        mask = torch.ones(4, 3, 8, 8, device=DEV)
        with pytest.warns(RuntimeWarning, match="exact fp32"):
            loss, log = m.loss_fn_irregular(x, mask)
        assert np.isfinite(log["karras loss"]) and ops._TRAIN_FWD_H2[0] == "0"
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in m.net.parameters())
    finally:
        ops.set_train_fwd_h2(before)
        ops.new_pack_epoch()
        m.zero_grad()


def test_graphed_step_skips_a_poisoned_update_then_leaves_the_f16x2_forward(tiny):
    from imageni2r_amd import FusedAdamW, GraphedTrainStep, ops
    m = tiny.train()
    before = ops._TRAIN_FWD_H2[0]
    ops.set_train_fwd_h2("1")
    ops.new_pack_epoch()
    try:
        opt = FusedAdamW(m.net.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0, ema=(m.model_ema, m.net))
        step = GraphedTrainStep(m, opt)
        gen = torch.Generator().manual_seed(4)
        mask = torch.ones(4, 3, 8, 8, device=DEV)
        good = torch.rand(4, 3, 8, 8, generator=gen).to(DEV)
        huge = good * 3.0e5
        assert np.isfinite(step(good, mask).item())
        w0 = [p.detach().clone() for p in m.net.parameters()]
        n0 = int(m.model_ema.num_updates.item())
        assert not np.isfinite(step(huge, mask).item())                   # f16x2 forward: inf / NaN loss ...
        assert all(torch.equal(a, b) for a, b in zip(w0, m.net.parameters()))  # ... and the update was skipped
        assert int(m.model_ema.num_updates.item()) == n0 and opt.skipped_steps() == 1
        with pytest.warns(RuntimeWarning, match="skipped"):
            assert step.check_range() == 1
        assert ops._TRAIN_FWD_H2[0] == "0" and not step._entries          # exact forward from here on, re-capture
        assert np.isfinite(step(huge, mask).item())
        assert any(not torch.equal(a, b) for a, b in zip(w0, m.net.parameters()))
        assert opt.skipped_steps() == 1 and step.check_range() == 0
    finally:
        ops.set_train_fwd_h2(before)
        ops.new_pack_epoch()
