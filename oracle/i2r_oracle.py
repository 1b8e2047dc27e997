"""CPU ORACLE for the ImagenI2R masked-diffusion hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (numpy for the index/byte work, torch-CPU fp32/fp64
functional ops for the floating-point work) of the algorithm the reference implements
in models/img_transformations.py, models/networks.py, models/our.py and
models/sampler.py.  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker / the timed CPU baseline.  The product path
(``imageni2r_amd``) never imports it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors.  The oracle
is pinned against outputs of the reference itself, imported on CPU in the build
container by ``tests/golden/make_golden.py``; the resulting small fixtures live in
``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks the oracle against
every one of them.

Everything here is written functionally over a flat ``dict`` of tensors keyed by the
reference's state-dict names (``model.enc.8x8_block0.conv0.weight`` ...), so the same
parameters can be fed to the oracle, to the reference and to the HIP product.
All citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Delay embedding (models/img_transformations.py)
# --------------------------------------------------------------------------------------


def delay_geometry(seq_len: int, delay: int, embedding: int) -> Tuple[int, int]:
    """(cols, side): number of image columns and padded square side.

    Restates the two loops of ``DelayEmbedder.ts_to_img`` (img_transformations.py:87-99):
    full columns while ``i*delay + embedding <= L``; one more partial column iff
    ``i*delay != L`` (the second conjunct at :95 is then always true).
    """
    n_full = 0 if seq_len < embedding else (seq_len - embedding) // delay + 1
    cols = n_full + (1 if n_full * delay != seq_len else 0)
    return cols, max(embedding, cols)


def delay_index_map(seq_len: int, delay: int, embedding: int) -> np.ndarray:
    """int32 (embedding, cols) array: source time index of image cell (row, col), -1 = zero fill.

    ``img[b,c,r,j] = ts[b, j*delay + r, c]`` when ``j*delay + r < L`` (img_transformations.py:88-99).
    """
    cols, _ = delay_geometry(seq_len, delay, embedding)
    r = np.arange(embedding, dtype=np.int64)[:, None]
    j = np.arange(cols, dtype=np.int64)[None, :]
    t = j * delay + r
    return np.where(t < seq_len, t, -1).astype(np.int32)


def ts_to_img(ts: np.ndarray, delay: int, embedding: int, pad_value: float = 0.0) -> np.ndarray:
    """(B, L, C) float32 -> (B, C, S, S) float32, bit-for-bit a copy (NaNs included).

    img_transformations.py:79-108 plus pad_to_square :59-70 (constant ``pad_value`` right/bottom).
    """
    ts = np.asarray(ts, dtype=np.float32)
    B, L, C = ts.shape
    cols, S = delay_geometry(L, delay, embedding)
    idx = delay_index_map(L, delay, embedding)  # (E, cols)
    img = np.full((B, C, S, S), np.float32(pad_value), dtype=np.float32)
    inner = np.zeros((B, C, embedding, cols), dtype=np.float32)
    valid = idx >= 0
    src = ts.transpose(0, 2, 1)  # (B, C, L)
    inner[:, :, valid] = src[:, :, idx[valid]]
    img[:, :, :embedding, :cols] = inner
    return img


def observed_mask(img: np.ndarray) -> np.ndarray:
    """run_irregular.py:216 / our.py:61 -- 1.0 where observed (not NaN), else 0.0."""
    return (~np.isnan(img)).astype(np.float32)


def inverse_positions(seq_len: int, delay: int, embedding: int) -> List[List[Tuple[int, int]]]:
    """For each t, the (row, col) cells holding it, in the reference's row-major scan order
    (img_transformations.py:131-138): ascending row, i.e. descending column."""
    cols, _ = delay_geometry(seq_len, delay, embedding)
    out: List[List[Tuple[int, int]]] = []
    for t in range(seq_len):
        cells = []
        for r in range(embedding):
            rem = t - r
            if rem < 0:
                break
            if rem % delay == 0 and rem // delay < cols:
                cells.append((r, rem // delay))
        out.append(cells)
    return out


def img_to_ts(img: np.ndarray, seq_len: int, delay: int, embedding: int) -> np.ndarray:
    """(B, C, S, S) float32|float64 -> (B, L, C) float32.

    img_transformations.py:140-170: per t, stack the k cells in scan order and ``mean``:
    a left-to-right sum in the input dtype followed by a true division by k (SURVEY §0.7,
    probed), then stored into an fp32 buffer.  t with no cell stays 0.
    """
    img = np.asarray(img)
    assert img.dtype in (np.float32, np.float64)
    B, C = img.shape[:2]
    out = np.zeros((B, seq_len, C), dtype=np.float32)
    for t, cells in enumerate(inverse_positions(seq_len, delay, embedding)):
        if not cells:
            continue
        acc = img[:, :, cells[0][0], cells[0][1]].copy()
        for (r, j) in cells[1:]:
            acc = acc + img[:, :, r, j]
        acc = acc / img.dtype.type(len(cells))
        out[:, t, :] = acc.astype(np.float32)
    return out


# --------------------------------------------------------------------------------------
# ADM UNet topology (models/networks.py:372-433)
# --------------------------------------------------------------------------------------


@dataclass
class BlockSpec:
    name: str  # e.g. "enc.8x8_block0"
    kind: str  # "conv" (plain 3x3 conv entry of the encoder) or "block"
    cin: int
    cout: int
    up: bool = False
    down: bool = False
    heads: int = 0
    skip_conv: bool = False  # 1x1 skip conv present (cin != cout)


@dataclass
class UNetSpec:
    img_resolution: int
    in_channels: int
    out_channels: int
    model_channels: int
    channel_mult: Tuple[int, ...]
    attn_resolutions: Tuple[int, ...]
    num_blocks: int = 3
    emb_channels: int = 0
    enc: List[BlockSpec] = field(default_factory=list)
    dec: List[BlockSpec] = field(default_factory=list)
    final_channels: int = 0


def unet_spec(img_resolution: int, in_channels: int, model_channels: int, channel_mult, attn_resolutions,
              num_blocks: int = 3, channels_per_head: int = 64) -> UNetSpec:
    """Block list of DhariwalUNet.__init__ (networks.py:402-433) with TS2img_Karras's kwargs (our.py:26-27)."""
    spec = UNetSpec(img_resolution, in_channels, in_channels, model_channels, tuple(channel_mult),
                    tuple(attn_resolutions), num_blocks, emb_channels=4 * model_channels)
    cout = in_channels
    for level, mult in enumerate(channel_mult):
        res = img_resolution >> level
        if level == 0:
            cin, cout = cout, model_channels * mult
            spec.enc.append(BlockSpec(f"enc.{res}x{res}_conv", "conv", cin, cout))
        else:
            spec.enc.append(BlockSpec(f"enc.{res}x{res}_down", "block", cout, cout, down=True))
        for idx in range(num_blocks):
            cin, cout = cout, model_channels * mult
            heads = cout // channels_per_head if res in attn_resolutions else 0
            spec.enc.append(BlockSpec(f"enc.{res}x{res}_block{idx}", "block", cin, cout, heads=heads,
                                      skip_conv=cin != cout))
    skips = [b.cout for b in spec.enc]
    for level, mult in reversed(list(enumerate(channel_mult))):
        res = img_resolution >> level
        if level == len(channel_mult) - 1:
            spec.dec.append(BlockSpec(f"dec.{res}x{res}_in0", "block", cout, cout, heads=cout // channels_per_head))
            spec.dec.append(BlockSpec(f"dec.{res}x{res}_in1", "block", cout, cout))
        else:
            spec.dec.append(BlockSpec(f"dec.{res}x{res}_up", "block", cout, cout, up=True))
        for idx in range(num_blocks + 1):
            cin = cout + skips.pop()
            cout = model_channels * mult
            heads = cout // channels_per_head if res in attn_resolutions else 0
            spec.dec.append(BlockSpec(f"dec.{res}x{res}_block{idx}", "block", cin, cout, heads=heads,
                                      skip_conv=cin != cout))
    spec.final_channels = cout
    return spec


# --------------------------------------------------------------------------------------
# Layers, functional (models/networks.py)
# --------------------------------------------------------------------------------------


def positional_embedding(x: torch.Tensor, num_channels: int, max_positions: int = 10000) -> torch.Tensor:
    """networks.py:200-206 (endpoint=False): [cos(x f), sin(x f)], f_i = (1/max_positions)^(i/(n/2))."""
    half = num_channels // 2
    freqs = torch.arange(half, dtype=torch.float32) / half
    freqs = (1.0 / max_positions) ** freqs
    arg = torch.outer(x.to(torch.float32), freqs)
    return torch.cat([arg.cos(), arg.sin()], dim=1)


def linear(x, w, b=None):
    """networks.py:39-43."""
    y = x @ w.t()
    return y if b is None else y + b


def avg_down2(x):
    """networks.py:84-85 with resample_filter [1,1]: depthwise 2x2 box, stride 2 == avg_pool2d."""
    return F.avg_pool2d(x, 2)


def nearest_up2(x):
    """networks.py:82-83: depthwise conv_transpose with 4f = ones(2,2), stride 2 == nearest x2."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def conv2d(x, w, b, up=False, down=False):
    """networks.py:68-90, non-fused resample path."""
    if up:
        x = nearest_up2(x)
    if down:
        x = avg_down2(x)
    if w is not None:
        x = F.conv2d(x, w, padding=w.shape[-1] // 2)
    if b is not None:
        x = x + b.reshape(1, -1, 1, 1)
    return x


def group_norm(x, w, b, eps=1e-5):
    """networks.py:97-106: groups = min(32, C // 4)."""
    C = x.shape[1]
    return F.group_norm(x, min(32, C // 4), w, b, eps)


def attention(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """networks.py:113-126,182-184.  qkv (B, 3C, H, W) with channel = (head, d, {q,k,v}) interleaved.

    returns a (B, C, H, W)."""
    B, C3, H, W = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B * heads, C // heads, 3, H * W).unbind(2)
    w = torch.einsum('ncq,nck->nqk', q, k / math.sqrt(k.shape[1])).softmax(dim=2)
    a = torch.einsum('nqk,nck->ncq', w, v)
    return a.reshape(B, C, H, W)


def unet_block(sd: Dict[str, torch.Tensor], pfx: str, bs: BlockSpec, x, emb, dropout_mask=None):
    """networks.py:166-187 with the DhariwalUNet kwargs (adaptive_scale, skip_scale=1, eps=1e-5).

    ``dropout_mask``: optional already-scaled keep mask (keep/(1-p)) multiplied where the reference
    applies F.dropout (:177); None == eval mode.
    """
    g = lambda n: sd.get(f"{pfx}.{n}")
    orig = x
    x = conv2d(F.silu(group_norm(x, g("norm0.weight"), g("norm0.bias"))), g("conv0.weight"), g("conv0.bias"),
               up=bs.up, down=bs.down)
    params = linear(emb, g("affine.weight"), g("affine.bias"))[:, :, None, None]
    scale, shift = params.chunk(2, dim=1)
    x = F.silu(shift + group_norm(x, g("norm1.weight"), g("norm1.bias")) * (scale + 1))
    if dropout_mask is not None:
        x = x * dropout_mask
    x = conv2d(x, g("conv1.weight"), g("conv1.bias"))
    if bs.skip_conv or bs.up or bs.down:
        x = x + conv2d(orig, g("skip.weight"), g("skip.bias"), up=bs.up, down=bs.down)
    else:
        x = x + orig
    if bs.heads:
        qkv = conv2d(group_norm(x, g("norm2.weight"), g("norm2.bias")), g("qkv.weight"), g("qkv.bias"))
        a = attention(qkv, bs.heads)
        x = conv2d(a, g("proj.weight"), g("proj.bias")) + x
    return x


def unet_forward(sd, spec: UNetSpec, x, noise_labels, prefix="model.", dropout_masks=None):
    """DhariwalUNet.forward (networks.py:435-461), label_dim = augment_dim = 0."""
    P = prefix
    emb = positional_embedding(noise_labels, spec.model_channels)
    emb = F.silu(linear(emb, sd[P + "map_layer0.weight"], sd[P + "map_layer0.bias"]))
    emb = F.silu(linear(emb, sd[P + "map_layer1.weight"], sd[P + "map_layer1.bias"]))
    skips = []
    dm = dropout_masks or {}
    for bs in spec.enc:
        if bs.kind == "conv":
            x = conv2d(x, sd[P + bs.name + ".weight"], sd[P + bs.name + ".bias"])
        else:
            x = unet_block(sd, P + bs.name, bs, x, emb, dm.get(bs.name))
        skips.append(x)
    for bs in spec.dec:
        if x.shape[1] != bs.cin:
            x = torch.cat([x, skips.pop()], dim=1)
        x = unet_block(sd, P + bs.name, bs, x, emb, dm.get(bs.name))
    x = F.silu(group_norm(x, sd[P + "out_norm.weight"], sd[P + "out_norm.bias"]))
    return conv2d(x, sd[P + "out_conv.weight"], sd[P + "out_conv.bias"])


def edm_precond(sd, spec: UNetSpec, x, sigma, sigma_data: float = 0.5, dropout_masks=None):
    """EDMPrecond.forward (networks.py:654-668)."""
    x = x.to(torch.float32)
    sigma = sigma.to(torch.float32).reshape(-1, 1, 1, 1)
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2).sqrt()
    c_in = 1 / (sigma_data ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    f = unet_forward(sd, spec, c_in * x, c_noise.flatten(), dropout_masks=dropout_masks)
    return c_skip * x + c_out * f


# --------------------------------------------------------------------------------------
# Masked EDM loss (models/our.py:53-78)
# --------------------------------------------------------------------------------------


def masked_edm_loss(sd, spec, x, mask, rnd_normal, noise, P_mean=-1.2, P_std=1.2, sigma_data=0.5,
                    dropout_masks=None):
    """loss_fn_irregular + forward_irregular with the two random draws injected.

    rnd_normal (B,1,1,1) ~ N(0,1) (our.py:71); noise (B,C,S,S) ~ N(0,1) (our.py:75, before *sigma).
    If mask is None it is derived from NaNs and x is nan_to_num'ed (our.py:60-62).
    """
    if mask is None:
        mask = (~torch.isnan(x)).to(torch.float32)
        x = torch.nan_to_num(x, nan=0.0)
    sigma = (rnd_normal * P_std + P_mean).exp()
    weight = (sigma ** 2 + sigma_data ** 2) / (sigma * sigma_data) ** 2
    n = noise * sigma
    D = edm_precond(sd, spec, x + n * mask, sigma, sigma_data, dropout_masks=dropout_masks)
    loss = (weight * (D * mask - x * mask).square()).mean()
    return loss


# --------------------------------------------------------------------------------------
# Heun sampler (models/sampler.py:32-65)
# --------------------------------------------------------------------------------------


def karras_sigmas(num_steps: int, sigma_min=0.002, sigma_max=80.0, rho=7.0) -> torch.Tensor:
    """sampler.py:39-42: fp64 schedule with t_N = 0 appended."""
    i = torch.arange(num_steps, dtype=torch.float64)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros(1, dtype=torch.float64)])


def heun_sample(denoise, latents: torch.Tensor, num_steps: int = 18) -> torch.Tensor:
    """sampler.py:44-65 with S_churn = 0 (gamma = 0, t_hat = t_cur, x_hat = x_cur).

    ``denoise(x_fp64, sigma_fp64_scalar) -> fp32 D``; state is fp64."""
    t_steps = karras_sigmas(num_steps)
    x_next = latents.to(torch.float64) * t_steps[0]
    for i in range(num_steps):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        x_hat = x_next
        d_cur = (x_hat - denoise(x_hat, t_cur).to(torch.float64)) / t_cur
        x_next = x_hat + (t_next - t_cur) * d_cur
        if i < num_steps - 1:
            d_prime = (x_next - denoise(x_next, t_next).to(torch.float64)) / t_next
            x_next = x_hat + (t_next - t_cur) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next


# --------------------------------------------------------------------------------------
# Parameter helpers shared by tests / golden generator / bench
# --------------------------------------------------------------------------------------

ZERO_INIT_SUFFIXES = ("conv1.weight", "proj.weight", "out_conv.weight")


def perturb_zero_init(sd: Dict[str, torch.Tensor], seed: int = 1234, scale_mult: float = 1.0) -> None:
    """In place: give the zero-initialised layers (networks.py:392,154,164,433) small random weights so that
    D(x) != c_skip*x and parity tests are not vacuous.  Deterministic in key order."""
    gen = torch.Generator().manual_seed(seed)
    for key in sorted(sd.keys()):
        if key.endswith(ZERO_INIT_SUFFIXES):
            w = sd[key]
            fan_in = w[0].numel()
            with torch.no_grad():
                w.copy_(torch.randn(w.shape, generator=gen) * (scale_mult * math.sqrt(1.0 / fan_in)))


def perturb_norm_affine(sd: Dict[str, torch.Tensor], seed: int = 4321, spread: float = 0.3) -> None:
    """In place: GroupNorm gamma <- 1 + spread*N(0,1), beta <- spread*N(0,1) (they are initialised to exactly 1 / 0,
    networks.py:100-101, which would hide a wrong per-channel gamma/beta index).  Deterministic in key order."""
    gen = torch.Generator().manual_seed(seed)
    for key in sorted(sd.keys()):
        if ".norm" in key or key.startswith(("model.out_norm", "out_norm")):
            w = sd[key]
            with torch.no_grad():
                w.copy_(torch.randn(w.shape, generator=gen) * spread + (1.0 if key.endswith("weight") else 0.0))


def ema_decay(num_updates: int, decay: float = 0.9999) -> float:
    """ema.py:29-31 after the increment."""
    return min(decay, (1 + num_updates) / (10 + num_updates))
