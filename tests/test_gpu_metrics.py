"""The evaluation scores with the series ON THE GPU: same code path as the CPU tests that pin them
to the reference bit for bit; here the device's own kernels (MIOpen GRU, rocBLAS, conv1d) train the scoring networks,
so the values are compared with the reference's CPU scores on the same inputs only as statistics -- a different
summation order moves a 200 / 2000-step training trajectory."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN
from tests.test_metrics_cpu import metric_inputs

pytestmark = pytest.mark.gpu
G8 = json.load(open(os.path.join(GOLDEN, "g8_metrics.json")))
G9 = json.load(open(os.path.join(GOLDEN, "g9_context_fid.json")))


def test_scores_on_device_tensors():
    from imageni2r_amd import context_fid as CF
    from imageni2r_amd import metrics as M
    real, gen = metric_inputs()
    real_d, gen_d = torch.from_numpy(real).cuda(), torch.from_numpy(gen).cuda()
    args = SimpleNamespace(input_size=real.shape[2], device="cuda")
    np.random.seed(0); torch.manual_seed(0)
    pred = M.predictive_score_metrics(real_d, gen_d, device="cuda")
    ref = G8["cases"][0]
    assert abs(pred - ref["pred"]) < 0.25 * ref["pred"], (pred, ref["pred"])
    np.random.seed(0); torch.manual_seed(0)
    corr = M.calculate_pearson_correlation(real_d, gen_d, device="cuda")
    assert abs(corr[0] - ref["corr_mean"]) <= 1e-3, (corr, ref["corr_mean"])   # no training: only the resample draw
    np.random.seed(0); torch.manual_seed(0)
    fid = float(CF.context_fid(real_d, gen_d, device="cuda"))
    want = G9["cases"][0]["fid"]
    assert np.isfinite(fid) and 0.4 * want < fid < 2.5 * want, (fid, want)
    # the encoder separates the two sets: the real set against a reshuffled copy of itself scores far lower
    np.random.seed(0); torch.manual_seed(0)
    same = float(CF.context_fid(real_d, real_d[torch.randperm(real_d.shape[0], device="cuda")], device="cuda"))
    assert abs(same) < 0.05 * fid, (same, fid)
    np.random.seed(0); torch.manual_seed(0)
    disc = float(M.discriminative_score_metrics(real_d, gen_d, args, iterations=300))
    assert 0.0 <= disc <= 0.5


def test_discriminator_loop_as_a_replayed_graph_equals_the_eager_loop(monkeypatch):
    """On the GPU the discriminator's mini-batch loop runs as one captured HIP graph per iteration (metrics._train_graphed):
    same draws, same algorithm (Adam in its capturable form) -- the score of a short run equals the eager loop's."""
    from imageni2r_amd import metrics as M
    real, gen = metric_inputs()
    real_d, gen_d = torch.from_numpy(real).cuda(), torch.from_numpy(gen).cuda()
    args = SimpleNamespace(input_size=real.shape[2], device="cuda")
    got = {}
    calls = []
    orig = M._train_graphed
    monkeypatch.setattr(M, "_train_graphed", lambda *a, **k: calls.append(1) or orig(*a, **k))
    for mode in ("0", "1"):
        monkeypatch.setenv("I2R_METRICS_GRAPH", mode)
        np.random.seed(3); torch.manual_seed(3)
        got[mode] = float(M.discriminative_score_metrics(real_d, gen_d, args, iterations=150))
        state = np.random.get_state()[1][:4].tolist()   # both loops leave numpy's generator in the same state
        got[mode + "_rng"] = state
    assert len(calls) == 1                               # (the graph path was taken exactly when enabled)
    assert abs(got["0"] - got["1"]) <= 1.0 / 32 + 1e-9, got   # accuracy of 64 test series: at most one flips
    assert got["0_rng"] == got["1_rng"]
