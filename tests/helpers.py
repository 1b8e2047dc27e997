"""Shared helpers for the test-suite (golden loading, tolerances, the tiny reference config)."""
import json
import os
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

TINY = dict(img_resolution=8, input_channels=3, ch_mult=[1, 2], unet_channels=64, attn_resolution=[8, 4],
            delay=3, embedding=8, seq_len=24, batch_size=4, ema=True, ema_warmup=2, diffusion_steps=18,
            beta1=1e-5, betaT=1e-2, deterministic=False, device="cpu")


def tiny_args(**kw):
    d = dict(TINY)
    d.update(kw)
    return SimpleNamespace(**d)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_meta(npz, key="meta"):
    return json.loads(bytes(npz[key]).decode())


def as_f32(bits):
    return np.ascontiguousarray(bits).view(np.float32)


def as_f64(bits):
    return np.ascontiguousarray(bits).view(np.float64)


def rel_err(a, b):
    """max |a-b| / max |b| -- how the 'relative error 1e-4' bar of BASELINE.json:north_star is evaluated."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def probe_vec(n, seed):
    """Deterministic probe for the gradient summaries in g4_net.npz (same as tests/golden/make_golden.py)."""
    return np.random.default_rng(seed).standard_normal(n)


def rms_rel_err(a, b):
    """sqrt(mean |a-b|^2) / sqrt(mean |b|^2): the RMS-relative companion of rel_err (a large element cannot hide the
    error of the small ones as it can in max/max)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def chan_rel_err(a, b):
    """Worst per-channel max|a-b| / max|b| over axis 1 of (B, C, ...) arrays: every channel is held to the bar on its
    own scale."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ax = tuple(i for i in range(a.ndim) if i != 1)
    num = np.abs(a - b).max(axis=ax)
    den = np.maximum(np.abs(b).max(axis=ax), 1e-30)
    return float((num / den).max())


# ---------------------------------------------------------------- full-size golden cases (tests/golden/g7_fullsize.npz)
# Small nets with topologies no reference config has (odd channel multipliers -> 3 and 6 attention heads, four levels
# down to a 1x1 image with attention there, a single level without attention, nominal resolution != image side, widths
# that shrink with depth): the plan builder, the tile / head / group dispatch and the sampler against the reference on
# configurations it was never tuned for.
EXTRA_NETS = {
    "R0_c64_112": dict(seq_len=24, delay=3, embedding=8, img_resolution=8, input_channels=2, unet_channels=64,
                       ch_mult=[1, 1, 2], attn_resolution=[4]),
    "R1_c64_21": dict(seq_len=50, delay=7, embedding=16, img_resolution=16, input_channels=4, unet_channels=64,
                      ch_mult=[2, 1], attn_resolution=[16, 8]),
    "R2_c128_13": dict(seq_len=24, delay=3, embedding=8, img_resolution=8, input_channels=9, unet_channels=128,
                       ch_mult=[1, 3], attn_resolution=[8]),
    "R3_c64_1234": dict(seq_len=30, delay=4, embedding=8, img_resolution=8, input_channels=1, unet_channels=64,
                        ch_mult=[1, 2, 3, 4], attn_resolution=[2, 1]),
    "R4_c64_1": dict(seq_len=96, delay=6, embedding=16, img_resolution=16, input_channels=5, unet_channels=64,
                     ch_mult=[1], attn_resolution=[]),
    "R5_c64_12_nominal8": dict(seq_len=96, delay=6, embedding=16, img_resolution=8, input_channels=6, unet_channels=64,
                               ch_mult=[1, 2], attn_resolution=[8]),
}
FULLSIZE = ("A_sine24", "B1_weather96", "C_electricity96", "D_etth1_768", "F_mujoco24", "H_energy768",
            "I_weather96_32") + tuple(EXTRA_NETS)
# the CPU oracle's own forward + backward + sampling is run at these (the rest would take the CPU suite past minutes;
# their goldens are checked against the HIP path on the GPU, and their init order / inputs on the CPU)
FULLSIZE_ORACLE = ("A_sine24", "B1_weather96", "C_electricity96", "D_etth1_768", "F_mujoco24") + tuple(EXTRA_NETS)


def fullsize_args(name, device="cpu"):
    from imageni2r_amd.workloads import WORKLOADS, workload_args
    if name in WORKLOADS:
        return workload_args(name, device=device)
    d = dict(ema=True, ema_warmup=100, diffusion_steps=18, beta1=1e-5, betaT=1e-2, deterministic=False, batch_size=4)
    d.update(EXTRA_NETS[name])
    d["device"] = device
    return SimpleNamespace(**d)


def fullsize_inputs(name):
    """Seeded inputs of the full-size golden case `name` -- regenerated identically by tests/golden/make_golden.py
    (which feeds them to the imported reference) and by the tests (torch's CPU generator is deterministic; the fixture
    stores a checksum of every tensor made here)."""
    import torch
    from imageni2r_amd.workloads import image_side
    args = fullsize_args(name)
    S, C, L = image_side(args), args.input_channels, args.seq_len
    g = torch.Generator().manual_seed(7000 + FULLSIZE.index(name))
    d = {"x": torch.randn(2, C, S, S, generator=g), "sigma": torch.tensor([0.3, 7.0])}
    ts = torch.rand(2, L, C, generator=g)
    for b in range(2):
        ts[b, torch.randperm(L, generator=g)[: int(0.3 * L)], :] = float("nan")
    d["ts"] = ts
    d["ts_tilde"] = torch.rand(2, L, C, generator=g)
    d["rnd"] = torch.randn(2, 1, 1, 1, generator=g)
    d["noise"] = torch.randn(2, C, S, S, generator=g)
    d["xT"] = torch.randn(1, C, S, S, generator=g)
    return args, S, d


def tensor_sha(t):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


def assert_close(a, b, tol, what=""):
    """The path's bar in two measures at once: max|a-b|/max|b| and the RMS-relative error (one large element cannot
    hide the error of the small ones in the second)."""
    e, r = rel_err(a, b), rms_rel_err(a, b)
    assert e < tol and r < tol, (what, "max/max %.3e rms %.3e (bar %.1e)" % (e, r, tol))


def _O():
    from oracle import i2r_oracle
    return i2r_oracle


def inverse_close(y, ref_bits, m):
    """img_to_ts against the reference's bits.  Up to four cells per time step -- every reference config has <= 3 --
    any summation order gives the same bits, and equality is required.  From five cells on ATen's CPU sum is not one
    order: inside one call the first 4*(B*C//4) outputs are summed left to right and the rest as four interleaved
    partial sums (scalar_outer_sum / row_sum in aten/src/ATen/native/cpu/SumKernel.cpp), and from B*C >= the host's
    vector width the split moves with the CPU's SIMD ISA -- the reference's own last bit depends on the machine
    there.  The build sums left to right; those outputs are held to 2 ulp."""
    kmax = max(len(c) for c in _O().inverse_positions(m["L"], m["d"], m["E"]))
    ref = as_f32(ref_bits)
    if kmax <= 4:
        return np.array_equal(y.view(np.uint32), ref_bits)
    return bool(np.all(np.abs(y.astype(np.float64) - ref) <= 2.4e-7 * np.maximum(np.abs(ref), 0.25)))
