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
