"""CPU checks against the FULL-SIZE reference goldens (tests/golden/g7_fullsize.npz + g7_init.json, made by
tests/golden/make_golden.py from the imported reference at BASELINE shapes A, B1, C, D and at the reference's other
architecture families F (64-channel mujoco), H (multipliers 1-2-4-8, energy-768), I (weather-96 on 32x32)):

* our constructors draw all 550 tensors of every shape bit-identically to the reference (init order pinned at full
  size, so the fixtures need not store weights);
* the seeded inputs regenerate to the stored checksums;
* the CPU oracle reproduces the reference's D(x, sigma), loss, every parameter gradient summary, the sampled series
  and the delay-embedding bits at every shape -- the oracle is pinned at the sizes the GPU tests use it at.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import i2r_oracle as O
from tests.helpers import (FULLSIZE, FULLSIZE_ORACLE, GOLDEN, as_f32, fullsize_inputs, load_golden, probe_vec, rel_err, rms_rel_err,
                           tensor_sha)

INIT = json.load(open(os.path.join(GOLDEN, "g7_init.json")))


def build_state(name):
    """Our model, reference init order, perturbed like the golden generator; returns (state dict, spec, args, S, d)."""
    from imageni2r_amd import TS2img_Karras
    args, S, d = fullsize_inputs(name)
    torch.manual_seed(0)
    m = TS2img_Karras(args, "cpu")
    sd = m.net.state_dict()
    return m, sd, args, S, d


@pytest.mark.parametrize("name", FULLSIZE)
def test_fullsize_init_order_and_inputs(name):
    m, sd, args, S, d = build_state(name)
    ref = INIT[name]
    assert list(sd.keys()) == list(ref["sha"].keys())
    for k, v in sd.items():
        assert hashlib.sha256(v.numpy().tobytes()).hexdigest()[:16] == ref["sha"][k], k
    assert sum(p.numel() for p in m.net.parameters()) == ref["nparams"]
    assert len(m.state_dict()) == ref["state_len"]
    assert [k for k, _ in m.net.named_parameters()] == ref["grad_names"]
    for k, v in d.items():
        assert tensor_sha(v) == ref["inputs"][k], k


@pytest.mark.parametrize("name", FULLSIZE_ORACLE)
def test_oracle_matches_reference_at_full_size(name):
    g = load_golden("g7_fullsize.npz")
    m, sd, args, S, d = build_state(name)
    O.perturb_zero_init(sd, seed=1234)
    O.perturb_norm_affine(sd, seed=4321)
    spec = O.unet_spec(args.img_resolution, args.input_channels, args.unet_channels, args.ch_mult,
                       args.attn_resolution)
    sdg = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if not k.endswith("resample_filter")}
    with torch.no_grad():
        D = O.edm_precond(sdg, spec, d["x"], d["sigma"])
        assert rel_err(D.numpy(), g[name + "/D_per_sample"]) < 2e-5
        assert rms_rel_err(D.numpy(), g[name + "/D_per_sample"]) < 2e-5
        D2 = O.edm_precond(sdg, spec, d["x"][:1].double(), torch.tensor(1.7, dtype=torch.float64))
        assert rel_err(D2.numpy(), g[name + "/D_scalar_sigma"]) < 2e-5
    # delay embedding bits, mask, loss, gradients
    img = O.ts_to_img(d["ts"].numpy(), args.delay, args.embedding)
    assert np.array_equal(img.view(np.uint32), g[name + "/img_bits"])
    mask = torch.from_numpy(O.observed_mask(img))
    x_tilde = torch.from_numpy(O.ts_to_img(d["ts_tilde"].numpy(), args.delay, args.embedding))
    loss = O.masked_edm_loss(sdg, spec, x_tilde, mask, d["rnd"], d["noise"])
    ref_loss = g[name + "/loss_value"][0]
    assert abs(loss.item() - ref_loss) < 1e-5 * abs(ref_loss)
    loss.backward()
    summ = g[name + "/grad_summary"]
    names = INIT[name]["grad_names"]
    for i, k in enumerate(names):
        gr = sdg[k].grad.double().numpy().ravel()
        got = np.array([gr.sum(), (gr * gr).sum(), (gr * probe_vec(gr.size, 9000 + i)).sum()])
        scale = max(np.sqrt(summ[i][1]) * np.sqrt(gr.size), 1e-12)
        assert abs(got[1] - summ[i][1]) <= 2e-4 * max(summ[i][1], 1e-20), k
        assert abs(got[0] - summ[i][0]) <= 1e-4 * scale, k
        assert abs(got[2] - summ[i][2]) <= 1e-4 * scale, k
    small = g[name + "/grad_small"]
    for i, off, n in g[name + "/grad_small_index"]:
        assert rel_err(sdg[names[i]].grad.numpy().ravel(), small[off:off + n]) < 1e-4, names[i]
    # the sampled series: the oracle's own Heun loop at shapes whose 35 evaluations take seconds on the CPU;
    # img_to_ts bits from the reference's sample at every shape
    if name in ("A_sine24", "B1_weather96", "F_mujoco24"):
        with torch.no_grad():
            out = O.heun_sample(lambda xx, s: O.edm_precond(sdg, spec, xx, s), d["xT"])
        assert out.dtype == torch.float64
        assert rel_err(out.numpy(), g[name + "/sample"]) < 1e-4
        assert rms_rel_err(out.numpy(), g[name + "/sample"]) < 1e-4
    ts = O.img_to_ts(g[name + "/sample"], args.seq_len, args.delay, args.embedding)
    assert np.array_equal(ts.view(np.uint32), g[name + "/sample_ts_bits"])
