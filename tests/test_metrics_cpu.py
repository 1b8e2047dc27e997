"""The evaluation metrics (imageni2r_amd/metrics.py) against scores the REFERENCE's own metric
functions produced on the same seeded inputs (tests/golden/g8_metrics.json, made by tests/golden/make_golden.py g8 from
the imported reference, CPU).  Same seeds, same random-number consumption order, same device: the discriminative
score -- 2000 Adam steps of a GRU classifier -- comes out bit-identical; the predictive score (batched test pass instead
of one call per series) and the correlation score (one matrix product instead of index-chunk loops) to 1e-5."""
import hashlib
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN

G8 = json.load(open(os.path.join(GOLDEN, "g8_metrics.json")))


def metric_inputs(seed=0, n=160, L=24, C=6):
    """Same construction as tests/golden/make_golden.py:metric_inputs (checksums are in the fixture)."""
    rng = np.random.default_rng(seed)
    t = np.arange(L)[None, :, None]
    f = rng.uniform(0.05, 0.3, (n, 1, C)); ph = rng.uniform(0, 2 * np.pi, (n, 1, C))
    real = 0.5 + 0.4 * np.sin(2 * np.pi * f * t + ph) + 0.02 * rng.standard_normal((n, L, C))
    f = rng.uniform(0.05, 0.3, (n, 1, C)); ph = rng.uniform(0, np.pi, (n, 1, C))
    gen = 0.5 + 0.4 * np.sin(2 * np.pi * f * t + ph) + 0.05 * rng.standard_normal((n, L, C))
    return real.astype(np.float32), gen.astype(np.float32)


@pytest.fixture(scope="module")
def data():
    real, gen = metric_inputs()
    assert hashlib.sha256(real.tobytes()).hexdigest()[:16] == G8["real_sha"]
    assert hashlib.sha256(gen.tobytes()).hexdigest()[:16] == G8["gen_sha"]
    return real, gen


@pytest.mark.parametrize("case", G8["cases"], ids=lambda c: f"seed{c['seed']}")
def test_scores_match_the_reference(data, case):
    from imageni2r_amd import metrics as M
    real, gen = data
    args = SimpleNamespace(input_size=real.shape[2], device="cpu")
    seed = case["seed"]
    threads = torch.get_num_threads()
    torch.set_num_threads(8)  # the fixture was made with 8 threads; tiny GEMMs, but keep the reduction splits equal
    try:
        if seed == 0:  # 2000 Adam steps of the GRU classifier: about a minute on the CPU, once is enough
            np.random.seed(seed); torch.manual_seed(seed)
            disc = float(M.discriminative_score_metrics(real, gen, args))
            assert disc == case["disc"], (disc, case["disc"])
        np.random.seed(seed); torch.manual_seed(seed)
        pred = M.predictive_score_metrics(real, gen, device="cpu")
        assert abs(pred - case["pred"]) <= 1e-5 * abs(case["pred"]), (pred, case["pred"])
        np.random.seed(seed); torch.manual_seed(seed)
        corr = M.calculate_pearson_correlation(real, gen, device="cpu")
        assert abs(corr[0] - case["corr_mean"]) <= 1e-4, (corr, case["corr_mean"])
    finally:
        torch.set_num_threads(threads)


def test_dispatcher_keys_and_device_resident_inputs(data, monkeypatch):
    from imageni2r_amd import metrics as M
    real, gen = data
    args = SimpleNamespace(input_size=real.shape[2], device="cpu")
    np.random.seed(1); torch.manual_seed(1)
    full = M.discriminative_score_metrics  # (the 2000-iteration run is what the golden test above exercises)
    monkeypatch.setattr(M, "discriminative_score_metrics", lambda a, b, c: full(a, b, c, iterations=40))
    r = M.evaluate_model_irregular(torch.from_numpy(real[:40]), torch.from_numpy(gen[:40]), args)
    assert set(r) == {"disc_mean", "disc_std"} and 0.0 <= r["disc_mean"] <= 0.5
    r2 = M.evaluate_model_irregular(real[:40], gen[:40], args, calc_other_metrics=True)
    assert set(r2) == {"pred_score_mean", "pred_score_std", "fid_score_mean", "fid_score_std",
                       "fid_score_conf_interval", "correlation_score_mean", "correlation_score_std",
                       "correlation_score_conf_interval"}
    assert r2["fid_score_mean"] > 0 and np.isnan(r2["fid_score_std"]) and r2["pred_score_mean"] > 0
    # identical sets are indistinguishable by correlation
    c = M.calculate_pearson_correlation(real, real, device="cpu")
    assert c[0] < 0.05
