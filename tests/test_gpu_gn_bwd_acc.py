"""GroupNorm backward with dgamma / dbeta accumulated into the training step's gradient slots
(i2r_groupnorm_bwd_acc_f32; models/networks.py:166-187's GroupNorm under autograd): the channel sums of a workgroup
(its samples of one group in the 64-lane-and-narrower kernel forms, its one sample in the whole-workgroup forms) meet
in LDS and leave as one atomic request per workgroup.  Checked against the per-sample-partials form + ordered row
reduction (i2r_groupnorm_bwd_f32 + i2r_reduce_rows_f32): dx and dscale / dshift bit for bit, the parameter gradients to
2e-6 of their largest value; twice in a row into the same slots, and with the switch that puts one atomic per sample
and channel back (I2R_GN_BWD_LDS=0).  1x1 images take gn_bwd_hw1_kernel (one lane per channel); checked against the
generic kernel it replaces as well (I2R_GN_HW1=0)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (B, C, H): kernel forms <256,1> <256,2> <256,4> <256,8> (few groups) | <64,4> (many groups) | <64,2> <32,1> <16,1>
CASES = [(32, 128, 16), (32, 256, 16), (7, 384, 16), (3, 256, 32), (1, 128, 16), (80, 128, 16), (32, 256, 8), (33, 256, 4),
         (32, 512, 2), (5, 64, 8),
         (128, 512, 1), (5, 1024, 1), (33, 256, 1), (3, 2048, 1)]   # 1x1 images: gn_bwd_hw1_kernel<16|32|64>


@pytest.fixture
def gn_env():
    names = ("I2R_GN_BWD_LDS", "I2R_GN_HW1")
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
        if H == 1:   # against the generic kernel (one wave per sample and group), which the 1x1 form replaces
            os.environ["I2R_GN_HW1"] = "0"
            ops.set_slots_zeroed(False)
            dx_g, dg_g, db_g, dss_g = ops.groupnorm_bwd(x, w, b, ss, mean, rstd, dy, 32, True, add=add)
            for got, ref in ((dx_ref, dx_g), (dg_ref, dg_g), (db_ref, db_g)) + (((dss_ref, dss_g),) if with_ss else ()):
                err = float((got - ref).abs().max() / ref.abs().max())
                assert err < 2e-6, err
    finally:
        ops.set_slots_zeroed(prev)


@pytest.mark.parametrize("B,C", [(4096, 512), (130, 1024), (7, 256), (3, 2048), (33, 96)])
@pytest.mark.parametrize("variant", ["plain", "ss_act", "ss1_act_split", "act_drop"])
def test_gn_fwd_on_1x1_images_against_the_generic_kernel_and_fp64(gn_env, B, C, variant):
    """gn_fwd_hw1_kernel<16|32|64> (one lane per channel; C = 96: 3 channels per group, 13 idle lanes per slot): output,
    mean and rstd against the generic kernel (I2R_GN_HW1=0; 2e-6, dropped elements at the same places) and the plain
    form against fp64 group normalisation."""
    from imageni2r_amd import ops
    gen = torch.Generator().manual_seed(B + C)
    x = (torch.randn(B, C, 1, 1, generator=gen) * 1.3 + 0.2).to(DEV)
    w, b = torch.randn(C, generator=gen).to(DEV), torch.randn(C, generator=gen).to(DEV)
    kw = {}
    if variant == "ss_act":
        kw = dict(ss=(torch.randn(B, 2 * C, generator=gen) * 0.3).to(DEV), act=True, save_stats=True)
    elif variant == "ss1_act_split":
        kw = dict(ss=(torch.randn(1, 2 * C, generator=gen) * 0.3).to(DEV), act=True, split_out=True)
    elif variant == "act_drop":
        kw = dict(act=True, drop_p=0.25, seed=1234, offset=77, save_stats=True)
    outs = {}
    for sw in ("1", "0"):
        os.environ["I2R_GN_HW1"] = sw
        outs[sw] = ops.groupnorm_fwd(x, w, b, 32, 1e-5, **kw)
    torch.cuda.synchronize()
    y1, y0 = outs["1"][0], outs["0"][0]
    if variant == "ss1_act_split":   # split pairs: compare what they decode to
        dec = lambda t: ((t.view(torch.int32) & 0xFFFF).to(torch.int16).view(torch.float16).double()   # noqa: E731
                         + ((t.view(torch.int32) >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).double())
        y1, y0 = dec(y1), dec(y0)
    assert float((y1 - y0).abs().max() / y0.abs().max()) < 2e-6
    if variant == "act_drop":
        assert torch.equal(y1 == 0, y0 == 0) and 0.15 < float((y1 == 0).float().mean()) < 0.35
    if kw.get("save_stats"):
        for i in (1, 2):
            assert float((outs["1"][i] - outs["0"][i]).abs().max() / outs["0"][i].abs().max()) < 2e-6
    if variant == "plain":
        xd = x.double().view(B, 32, C // 32)
        ref = ((xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-5)).view(B, C, 1, 1)
        ref = ref * w.double().view(1, C, 1, 1) + b.double().view(1, C, 1, 1)
        assert float((y1.double() - ref).abs().max() / ref.abs().max()) < 2e-6
