"""CPU-only: checkpoints in the reference's .pth schema (run_irregular.py:41-87) and the pointer-swapped ema_scope
(models/our.py:87-108).  No kernels run here."""
import os

import numpy as np
import pytest

import torch

from tests.helpers import tiny_args


def _model(seed):
    from imageni2r_amd import TS2img_Karras
    torch.manual_seed(seed)
    m = TS2img_Karras(tiny_args(), "cpu")
    with torch.no_grad():  # make the shadow weights differ from the live ones
        for b in m.model_ema.buffers():
            if b.dtype == torch.float32 and b.ndim > 0:
                b.add_(0.25)
    return m


def test_ema_scope_swaps_storage_and_restores_addresses():
    m = _model(0)
    names = m.model_ema.m_name2s_name
    shadow = dict(m.model_ema.named_buffers())
    live0 = {k: p.detach().clone() for k, p in m.net.named_parameters()}
    ema0 = {k: shadow[names[k]].detach().clone() for k in live0}
    ptr0 = {k: p.data_ptr() for k, p in m.net.named_parameters()}
    sptr0 = {k: shadow[names[k]].data_ptr() for k in live0}
    with m.ema_scope():
        inside = dict(m.model_ema.named_buffers())
        for k, p in m.net.named_parameters():
            assert torch.equal(p, ema0[k]) and p.data_ptr() == sptr0[k], k  # EMA weights, EMA storage: no copy made
            assert torch.equal(inside[names[k]], live0[k])
            assert p.requires_grad and isinstance(p, torch.nn.Parameter)
    after = dict(m.model_ema.named_buffers())
    for k, p in m.net.named_parameters():
        assert torch.equal(p, live0[k]) and p.data_ptr() == ptr0[k], k
        assert torch.equal(after[names[k]], ema0[k]) and after[names[k]].data_ptr() == sptr0[k]


def test_ema_scope_restores_after_an_exception_and_copy_mode_agrees(monkeypatch):
    m = _model(1)
    live0 = [p.detach().clone() for p in m.net.parameters()]
    try:
        with m.ema_scope():
            swapped = [p.detach().clone() for p in m.net.parameters()]
            raise KeyError("boom")
    except KeyError:
        pass
    assert all(torch.equal(a, b) for a, b in zip(live0, m.net.parameters()))
    monkeypatch.setenv("I2R_EMA_SCOPE", "copy")  # the reference's store / copy_to / restore sequence
    ptrs = [p.data_ptr() for p in m.net.parameters()]
    with m.ema_scope():
        assert all(torch.equal(a, b) for a, b in zip(swapped, m.net.parameters()))
        assert ptrs == [p.data_ptr() for p in m.net.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(live0, m.net.parameters()))
    m.use_ema = False
    with m.ema_scope():
        assert all(torch.equal(a, b) for a, b in zip(live0, m.net.parameters()))


def test_checkpoint_round_trip_in_reference_schema(tmp_path):
    from imageni2r_amd import checkpoint as ck
    args = tiny_args(model_save_path=str(tmp_path), seq_len=24, dataset="sine", missing_rate=0.3)
    m = _model(2)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)  # run_irregular.py:117
    for p in m.net.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    enc, dec = torch.nn.Linear(3, 4), torch.nn.Linear(4, 3)
    topt = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()))
    stale = os.path.join(ck.checkpoint_dir(args), "disc_score_9.pth")
    os.makedirs(os.path.dirname(stale))
    open(stale, "w").write("old")
    path = ck.save_checkpoint(args, m, opt, m.model_ema, enc, dec, topt, 0.125, pred_score=0.5)
    assert path.endswith("seq_len_24/sine/missing_rate_30/"
                         "disc_score_0.125_pred_score_0.5_fid_score_None_correlation_score_None.pth")
    assert os.listdir(os.path.dirname(path)) == [os.path.basename(path)]  # older files are removed (:55-62)
    raw = torch.load(path, weights_only=True)
    assert tuple(raw.keys()) == ck.CHECKPOINT_KEYS
    assert list(raw["our_model_state_dict"].keys()) == list(m.state_dict().keys())
    assert raw["args"]["dataset"] == "sine"

    m2 = _model(3)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-4, weight_decay=1e-5)
    enc2, dec2 = torch.nn.Linear(3, 4), torch.nn.Linear(4, 3)
    state = {"model": m2, "optimizer": opt2, "encoder": enc2, "decoder": dec2, "epoch": 7}
    out = ck.restore_checkpoint(str(tmp_path), state, ema_model=m2.model_ema)
    assert out is state and state["epoch"] == 7 and state["disc_score"] == 0.125 and state["path"] == path
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(enc.weight, enc2.weight) and torch.equal(dec.bias, dec2.bias)
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[i]["exp_avg_sq"], s2[i]["exp_avg_sq"]) for i in s1)


def test_save_checkpoint_reports_failure_instead_of_raising(tmp_path, capsys):
    from imageni2r_amd import checkpoint as ck
    blocker = tmp_path / "file"
    blocker.write_text("x")
    args = tiny_args(model_save_path=str(blocker), seq_len=24, dataset="sine", missing_rate=0.3)
    assert ck.save_checkpoint(args, torch.nn.Linear(1, 1), torch.optim.SGD([torch.zeros(1, requires_grad=True)], 0.1),
                              None, None, None, None, 1.0) is None
    assert "Failed to save checkpoint" in capsys.readouterr().out


def test_checkpoint_with_numpy_scores_as_the_metrics_stage_returns_them(tmp_path):
    """metrics/__init__.py:50 hands back np.round(np.mean(..), 4) -- numpy.float64 scalars.  Ours are stored as plain
    numbers (same file name); a file that holds numpy scalars, as the reference writes them, still restores."""
    from imageni2r_amd import checkpoint as ck
    args = tiny_args(model_save_path=str(tmp_path), seq_len=24, dataset="sine", missing_rate=0.3)
    m = _model(4)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5)
    disc = np.round(np.mean([0.12345, 0.2]), 4)
    pred = np.round(np.mean([0.5, 0.25]), 4)
    assert isinstance(disc, np.float64)
    path = ck.save_checkpoint(args, m, opt, m.model_ema, None, None, None, disc, pred_score=pred)
    assert os.path.basename(path) == f"disc_score_{disc}_pred_score_{pred}_fid_score_None_correlation_score_None.pth"
    raw = torch.load(path, weights_only=True)  # nothing numpy in the pickle
    assert type(raw["disc_score"]) is float and raw["disc_score"] == float(disc)
    state = ck.restore_checkpoint(str(tmp_path), {"model": _model(5)})
    assert state["disc_score"] == float(disc) and state["pred_score"] == float(pred)
    # a reference-written file: the scores are pickled numpy scalars
    ref_style = dict(raw)
    ref_style["disc_score"], ref_style["pred_score"] = disc, pred
    torch.save(ref_style, path)
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)
    m6 = _model(6)
    state = ck.restore_checkpoint(str(tmp_path), {"model": m6})
    assert state["disc_score"] == disc and isinstance(state["disc_score"], np.float64)
    for (k, a), (_, b) in zip(m.state_dict().items(), m6.state_dict().items()):
        assert torch.equal(a, b), k
