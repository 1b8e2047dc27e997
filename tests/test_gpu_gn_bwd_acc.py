"""GroupNorm backward with dgamma / dbeta accumulated into the training step's gradient slots
(i2r_groupnorm_bwd_acc_f32; models/networks.py:166-187's GroupNorm under autograd): the channel sums of a workgroup
(its samples of one group in the 64-lane-and-narrower kernel forms, its one sample in the whole-workgroup forms) meet
in LDS and leave as one atomic request per workgroup.  Checked against the per-sample-partials form + ordered row
reduction (i2r_groupnorm_bwd_f32 + i2r_reduce_rows_f32): dx and dscale / dshift bit for bit, the parameter gradients to
2e-6 of their largest value; twice in a row into the same slots, and with the switch that puts one atomic per sample
and channel back (I2R_GN_BWD_LDS=0)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (B, C, H): kernel forms <256,1> <256,2> <256,4> <256,8> (few groups) | <64,4> (many groups) | <64,2> <32,1> <16,1>
CASES = [(32, 128, 16), (32, 256, 16), (7, 384, 16), (3, 256, 32), (1, 128, 16), (80, 128, 16), (32, 256, 8), (33, 256, 4),
         (32, 512, 2), (5, 64, 8)]


@pytest.fixture
def gn_env():
    names = ("I2R_GN_BWD_LDS",)
    prev = {n: os.environ.get(n) for n in names}
    yield
    for n, v in prev.items():
        if v is None:
            os.environ.pop(n, None)
        else:
            os.environ[n] = v


@pytest.mark.parametrize("B,C,H", CASES)
@pytest.mark.parametrize("with_ss", [True, False])
def test_gn_bwd_slot_accumulation_against_the_partials_form(gn_env, B, C, H, with_ss):
    from imageni2r_amd import ops
    gen = torch.Generator().manual_seed(B * 1000 + C + H)
    x = (torch.randn(B, C, H, H, generator=gen) * 1.7 + 0.3).to(DEV)
    dy = torch.randn(B, C, H, H, generator=gen).to(DEV)
    add = torch.randn(B, C, H, H, generator=gen).to(DEV)
    w, b = torch.randn(C, generator=gen).to(DEV), torch.randn(C, generator=gen).to(DEV)
    ss = (torch.randn(B, 2 * C, generator=gen) * 0.3).to(DEV) if with_ss else None
    _, mean, rstd = ops.groupnorm_fwd(x, w, b, 32, 1e-5, ss=ss, act=True, save_stats=True)
    prev = ops.slots_zeroed()
    try:
        ops.set_slots_zeroed(False)
        dx_ref, dg_ref, db_ref, dss_ref = ops.groupnorm_bwd(x, w, b, ss, mean, rstd, dy, 32, True, add=add)
        ops.set_slots_zeroed(True)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        for lds in ("1", "1", "0"):
            os.environ["I2R_GN_BWD_LDS"] = lds
            dg.zero_()
            db.zero_()
            dx, og, ob, dss = ops.groupnorm_bwd(x, w, b, ss, mean, rstd, dy, 32, True, dgamma=dg, dbeta=db, add=add)
            torch.cuda.synchronize()
            assert og.data_ptr() == dg.data_ptr() and ob.data_ptr() == db.data_ptr()
            assert torch.equal(dx, dx_ref)
            if with_ss:
                assert torch.equal(dss, dss_ref)
            for got, ref in ((og, dg_ref), (ob, db_ref)):
                err = float((got - ref).abs().max() / ref.abs().max())
                assert err < 2e-6, (lds, err)
    finally:
        ops.set_slots_zeroed(prev)
