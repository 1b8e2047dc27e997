"""Context-FID (imageni2r_amd/context_fid.py) against the REFERENCE's TS2Vec fit + Frechet distance
on the same seeded inputs (tests/golden/g9_context_fid.json, made by tests/golden/make_golden.py g9 from the imported
reference on the CPU): same seeds, same random-number consumption order, same torch ops in the same order, so the 200
training iterations, both embeddings and the score reproduce -- asserted to 1e-6 relative (they come out equal to the
last bit in the build container; the slack is for a different BLAS thread split)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN
from tests.test_metrics_cpu import metric_inputs

G9 = json.load(open(os.path.join(GOLDEN, "g9_context_fid.json")))
TOL = 1e-6


def inputs(tag):
    real, gen = metric_inputs()
    assert hashlib.sha256(real.tobytes()).hexdigest()[:16] == G9["real_sha"]
    assert hashlib.sha256(gen.tobytes()).hexdigest()[:16] == G9["gen_sha"]
    if tag == "ragged":  # (as in make_golden.py g9)
        real, gen = real.copy(), gen.copy()
        real[:, :3] = np.nan; real[::2, -2:] = np.nan
        gen[1::3, 5] = np.nan
    return real, gen


@pytest.mark.parametrize("case", [c for c in G9["cases"] if c["seed"] != 3],  # (seed 3: fixture only; CPU time)
                         ids=lambda c: f"{c['tag']}_seed{c['seed']}")
def test_fit_embeddings_and_score_match_the_reference(case):
    from imageni2r_amd import context_fid as CF
    real, gen = inputs(case["tag"])
    threads = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        np.random.seed(case["seed"]); torch.manual_seed(case["seed"])
        model = CF.TS2VecFit(input_dims=real.shape[-1], device="cpu", batch_size=8, lr=0.001, output_dims=320,
                             max_train_length=3000)
        losses = model.fit(real.copy())
        assert model.n_iters == case["n_iters"] and len(losses) == len(case["epoch_losses"])
        np.testing.assert_allclose(losses, case["epoch_losses"], rtol=TOL)
        r1, r2 = model.encode_full_series(real.copy()), model.encode_full_series(gen.copy())
        assert r1.shape == (real.shape[0], 320) and r1.dtype == np.float32
        np.testing.assert_allclose(r1[0, :8], case["repr_real_head"], rtol=10 * TOL, atol=1e-7)
        np.testing.assert_allclose(r2[-1, :8], case["repr_gen_head"], rtol=10 * TOL, atol=1e-7)
        assert abs(r1.astype(np.float64).sum() - case["repr_real_sum"]) <= 10 * TOL * abs(case["repr_real_sum"])
        assert abs(r2.astype(np.float64).sum() - case["repr_gen_sum"]) <= 10 * TOL * abs(case["repr_gen_sum"])
        order = np.random.permutation(real.shape[0])
        fid = float(CF.frechet_distance(r1[order], r2[order]))
        assert abs(fid - case["fid"]) <= 1e-4 * abs(case["fid"]), (fid, case["fid"])
        if case is G9["cases"][0]:  # the one-call form consumes the generators the same way (one more fit: once)
            np.random.seed(case["seed"]); torch.manual_seed(case["seed"])
            again = float(CF.context_fid(real, gen, device="cpu"))
            assert abs(again - case["fid"]) <= 1e-4 * abs(case["fid"]), (again, case["fid"])
        print(case["tag"], case["seed"], "fid", fid, "reference", case["fid"], "bit-equal:", fid == case["fid"])
    finally:
        torch.set_num_threads(threads)


def test_frechet_distance_and_summary():
    from imageni2r_amd import context_fid as CF
    rng = np.random.default_rng(0)
    a = rng.standard_normal((400, 5))
    assert abs(CF.frechet_distance(a, a)) < 1e-9
    b = a + np.array([1.0, 0, 0, 0, 0])
    assert abs(CF.frechet_distance(a, b) - 1.0) < 1e-9       # equal covariances: the squared mean shift
    c = 2.0 * a                                              # tr(S + 4S - 2 * 2S) = tr(S), plus |mu|^2
    want = np.sum(a.mean(0) ** 2) + np.trace(np.cov(a, rowvar=False))
    assert abs(CF.frechet_distance(a, c) - want) < 1e-8
    m, s, h = CF.summarise_scores([0.25])
    assert m == 0.25 and np.isnan(s) and np.isnan(h)
    m, s, h = CF.summarise_scores([0.2, 0.4])
    assert m == 0.3 and s == round(float(np.std([0.2, 0.4], ddof=1)), 4) and h > 0


def test_long_series_are_cut_and_padded():
    from imageni2r_amd import context_fid as CF
    x = np.arange(2 * 13 * 1, dtype=np.float64).reshape(2, 13, 1)
    y = CF._split_long_series(x, 4)                          # 13 // 4 = 3 sections of 5, 4, 4 -> padded to 5
    assert y.shape == (6, 5, 1)
    assert np.array_equal(y[0, :, 0], x[0, :5, 0]) and np.isnan(y[2, 4, 0]) and y[2, 3, 0] == x[0, 8, 0]
    assert CF._split_long_series(x, 7) is x                  # fewer than two sections: untouched
    z = np.full((1, 6, 1), np.nan); z[0, 0:2, 0] = [1.0, 2.0]  # observed head, 4 missing at the tail
    c = CF._centre_observed_span(z)
    assert np.array_equal(np.isnan(c[0, :, 0]), [True, True, False, False, True, True])
