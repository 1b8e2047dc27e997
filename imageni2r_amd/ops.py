"""Tensor-level wrappers over the C ABI (include/i2r_hip.h) and their torch.autograd.Function shells.

PyTorch is used here for device memory, streams and the autograd tape only; every arithmetic step is a HIP kernel of
libi2r_hip.so.  All wrappers validate on the host (device, dtype, contiguity, shapes) before a pointer is handed over.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib

_vp = ctypes.c_void_p


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current device's current HIP stream as a C pointer.  (torch.cuda.current_stream() builds a Stream object,
    ~9 us a call: with ~460 launches per eager training step that alone was 4 ms of a 31 ms step.)"""
    if _raw_stream is not None:
        return _vp(_raw_stream(torch.cuda.current_device()))
    return _vp(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.I2RError(f"{name}: tensor is on {t.device}; the HIP path needs a 'cuda' (HIP) tensor and there is "
                            "no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]):
    return None if t is None else _vp(t.data_ptr())


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


_PACK_EPOCH = [0]
_CONV_MATH = ["f16x2"]  # math of INFERENCE (no-grad) conv forwards: "f16x2" (split-f16 MFMA), "f32" (exact fp32 MFMA)
                        # or "f16" (f16x2 with the 3x3 convs reduced to the plain f16 product; optional, 1e-2 class)


def conv_math() -> str:
    return _CONV_MATH[0]


def h2_math() -> bool:
    """True when inference convs / attention run on the half-precision matrix cores ("f16x2" or "f16")."""
    return _CONV_MATH[0] in ("f16x2", "f16")


def set_conv_math(mode: str) -> None:
    """Select the inference conv kernel: "f16x2" (default; ~1e-6 rel. error, 3-5x faster than fp32 MFMA), "f32" (exact
    fp32 MFMA) or "f16" -- the optional reduced-precision path (north_star's "bf16-MFMA path", built on f16 for its three
    extra significand bits at the same matrix rate): as "f16x2", but the 3x3 convs multiply the f16 high halves only
    (one MFMA instead of three, ~5e-4 per layer; the whole denoiser stays inside 1e-2).
    Training: the backward always uses the exact fp32 kernels; the forward of convs with >= 2048 output positions runs
    on the f16x2 kernel unless the mode is "f32" (set_train_fwd_h2 / I2R_TRAIN_FWD_H2; never the "f16" product)."""
    if mode not in ("f16x2", "f32", "f16"):
        raise ValueError(mode)
    _CONV_MATH[0] = mode


# Kernel-selection knobs, read from the environment ONCE at import (A/B switches for measurements; the defaults are the
# product path).  They are part of every captured graph's cache key (kernel_mode_key).
_KNOBS = {
    "no_fuse_up": bool(os.environ.get("I2R_NO_FUSE_UP")),            # materialise the decoder's x2 upsample
    "no_gn_bypass": bool(os.environ.get("I2R_NO_GN_BYPASS")),        # training: residual branch outside GroupNorm
    "no_qkv_presplit": bool(os.environ.get("I2R_NO_QKV_PRESPLIT")),  # attention splits q,k,v itself
    # qkv bytes per attention-tail chunk.  256 MB (one infinity cache) until round 3; measured then: 256 -> 1024 MB / off
    # B1 1672 -> 1697 series/s, D 383 -> 390 (the chunks' small launches cost more than the cache returns); 1024 MB keeps
    # every chunk's qkv tensor under the kernels' 2 GiB addressing limit whatever the batch.
    "attn_chunk_bytes": float(os.environ.get("I2R_ATTN_CHUNK_MB", 1024)) * 1e6,
}


def knob(name):
    return _KNOBS[name]


def set_knob(name, value) -> None:
    if name not in _KNOBS:
        raise KeyError(name)
    _KNOBS[name] = value


def kernel_mode_key():
    """Everything that changes WHICH kernels an inference pass launches (for graph cache keys)."""
    return (_CONV_MATH[0],) + tuple(sorted(_KNOBS.items()))


# Training forwards of the convs on the f16x2-split kernels (backward always uses the exact fp32 kernels):
# "auto" = convs with at least _TRAIN_FWD_H2_MINPOS output positions (below that the per-layer weight split costs more
# than the faster kernel saves), "1" = all, "0" = none.
_TRAIN_FWD_H2 = [os.environ.get("I2R_TRAIN_FWD_H2", "auto")]
_TRAIN_FWD_H2_MINPOS = [int(os.environ.get("I2R_TRAIN_FWD_H2_MINPOS", 2048))]


def train_fwd_h2(npos=None) -> bool:
    """Does a training conv forward with ``npos`` output positions (None: "every conv?") run on the f16x2 kernel?"""
    mode = _TRAIN_FWD_H2[0]
    if not h2_math() or mode == "0":
        return False
    if mode == "1":
        return True
    return npos is not None and npos >= _TRAIN_FWD_H2_MINPOS[0]


# Training DATA GRADIENTS of the convs on the f16x2-split kernel (i2r_conv2d_dgrad_h2: the forward kernel on transposed,
# tap-flipped weights, dy scaled by a per-tensor power of two so that it splits at full precision).  "auto": convs with
# at least _TRAIN_BWD_H2_MINPOS positions, or at least _MIDPOS positions and _MIDMACS multiply-adds per position -- each
# such layer pays ~3 small launches (weight split, abs-max, scale) and the kernel splits dy itself, so it wins only
# where the exact fp32-MFMA kernel is clearly matrix-pipe bound (measured, graph-mode step, after the abs-max kernel
# stopped queueing on one atomic: positions alone >= 8192 still loses 0.7 % on B1/C/A while gaining 3.7 % on D; with the
# work criterion every shape gains -- B1 15.72 -> 15.50 ms, C 16.13 -> 15.85, A 14.87 -> 14.61, D 33.1 -> 31.8); "0":
# never.  Weight gradients stay exact fp32.
_TRAIN_BWD_H2 = [os.environ.get("I2R_TRAIN_BWD_H2", "auto")]
_TRAIN_BWD_H2_MINPOS = [int(os.environ.get("I2R_TRAIN_BWD_H2_MINPOS", 32768))]


_TRAIN_BWD_H2_MIDPOS = [int(os.environ.get("I2R_TRAIN_BWD_H2_MIDPOS", 2048))]
_TRAIN_BWD_H2_MIDMACS = [int(os.environ.get("I2R_TRAIN_BWD_H2_MIDMACS", 250000))]


def train_bwd_h2(npos, macs_per_pos=0) -> bool:
    """``macs_per_pos`` = Cin * Cout * k * k.  Besides the position threshold, layers from _MIDPOS positions on qualify
    when they are heavy per position (the 256-channel 3x3 convs of the 16x16 level of the seq_len-768 shapes)."""
    if not (h2_math() and _TRAIN_BWD_H2[0] != "0" and _TRAIN_FWD_H2[0] != "0"):
        return False
    return npos >= _TRAIN_BWD_H2_MINPOS[0] or (npos >= _TRAIN_BWD_H2_MIDPOS[0]
                                               and macs_per_pos >= _TRAIN_BWD_H2_MIDMACS[0])


def set_train_bwd_h2(mode, minpos=None) -> None:
    mode = {True: "auto", False: "0"}.get(mode, mode)
    if mode not in ("auto", "0"):
        raise ValueError(mode)
    _TRAIN_BWD_H2[0] = mode
    if minpos is not None:
        _TRAIN_BWD_H2_MINPOS[0] = int(minpos)


def train_fwd_h2_enabled() -> bool:
    """May ANY training conv forward run on the f16x2 kernel (whose operands must stay below fp16's 65504)?"""
    return h2_math() and _TRAIN_FWD_H2[0] != "0"


def set_train_fwd_h2(mode) -> None:
    """"auto" (default), True / "1" (every conv) or False / "0" (exact fp32 forward everywhere)."""
    mode = {True: "1", False: "0"}.get(mode, mode)
    if mode not in ("auto", "0", "1"):
        raise ValueError(mode)
    _TRAIN_FWD_H2[0] = mode


_DETERMINISTIC = [bool(os.environ.get("I2R_DETERMINISTIC"))]


def deterministic() -> bool:
    return _DETERMINISTIC[0]


def set_deterministic(flag: bool) -> None:
    """Run-to-run reproducible gradients: the conv weight gradients reduce their position splits through per-split
    slabs added in a fixed order (i2r_conv2d_wgrad_det_f32) instead of fp32 atomics, and GroupNorm's dgamma / dbeta go
    through per-sample partials and an ordered row reduction instead of atomic accumulation.  Everything else on the
    path already has a fixed summation order.  A few per cent slower; a graph-replayed step is then bit-identical to
    the same step run eagerly.  (Env: I2R_DETERMINISTIC=1.)  Captured graphs keep the mode they were captured in."""
    _DETERMINISTIC[0] = bool(flag)


_RNG_STEP = [None]  # device uint64 step counter of the training RNG streams (set by GraphedTrainStep), or None


def rng_step():
    return _RNG_STEP[0]


def set_rng_step(counter) -> None:
    """Route every training RNG stream (dropout masks, sigma / noise draws) through a device-resident step counter:
    Philox offset = the call site's fixed id + (counter << 20).  ``None`` restores host-side offsets."""
    if counter is not None:
        if counter.dtype != torch.int64 or counter.numel() != 1 or not counter.is_cuda:
            raise TypeError("rng step counter must be a 1-element int64 HIP tensor")
    _RNG_STEP[0] = counter


_RNG_RANK = [None]  # override of the data-parallel rank folded into the training RNG seed (tests); None = ask torch.distributed


def set_rng_rank(rank) -> None:
    _RNG_RANK[0] = None if rank is None else int(rank)


def rng_seed() -> int:
    """Philox key of the training streams (sigma / noise draws, dropout masks): torch's seed with the data-parallel
    rank folded in.  Data-parallel ranks must seed torch identically to build identical initial weights; without the
    rank in the key every rank would then draw the same sigma, noise field and dropout masks for its local sample b,
    and a global batch would hold each draw world-size times."""
    rank = _RNG_RANK[0]
    if rank is None:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return (torch.initial_seed() ^ (rank * 0x9E3779B97F4A7C15)) & ((1 << 63) - 1)


def counter_add(counter, inc=1):
    _lib.call("i2r_counter_add_u64", _p(counter), int(inc), _stream())


_SLOTS_ZEROED = [False]


def slots_zeroed() -> bool:
    return _SLOTS_ZEROED[0]


def set_slots_zeroed(flag: bool) -> None:
    """Promise that every parameter's ``_grad_slot`` buffer was zero-filled at the start of this step: weight-gradient
    kernels then ADD into the slots (no per-layer memsets) instead of overwriting them."""
    _SLOTS_ZEROED[0] = bool(flag)


def param_slot(param):
    """The gradient slot the backward kernels may WRITE for this parameter, or None.  A parameter that already holds a
    gradient (a second backward without zero_grad: gradient accumulation) gets None: its kernels then return a fresh
    tensor and autograd adds it to ``.grad`` -- writing the slot again would hand autograd an alias of ``.grad`` itself
    and double the last gradient instead of summing the two."""
    if param is None or param.grad is not None:
        return None
    return getattr(param, "_grad_slot", None)


def grad_slot(param):
    """A fresh alias of the parameter's preassigned gradient buffer (FusedAdamW / GradBucketReducer own it), or None.
    Backward kernels write the gradient there and autograd adopts the alias as ``param.grad`` without a copy, so
    gradient addresses never change and nothing is accumulated by separate add kernels."""
    slot = getattr(param, "_grad_slot", None)
    return None if slot is None else slot.detach()


def pack_epoch() -> int:
    return _PACK_EPOCH[0]


def new_pack_epoch() -> int:
    """Invalidate every cached packed weight (used at the start of a hipGraph capture)."""
    _PACK_EPOCH[0] += 1
    return _PACK_EPOCH[0]


# ======================================================================================================= raw ops
def ts_to_img(ts, delay, embedding, cols, side, pad_value=0.0, want_mask=False):
    _chk(ts, "ts")
    B, L, C = ts.shape
    img = torch.empty((B, C, side, side), dtype=torch.float32, device=ts.device)
    mask = torch.empty_like(img) if want_mask else None
    _lib.call("i2r_ts_to_img_f32", _p(ts), _p(img), _p(mask), B, L, C, delay, embedding, cols, side, float(pad_value),
              _stream())
    return (img, mask) if want_mask else img


def ts_to_img_bwd(dimg, L, delay, embedding, cols):
    _chk(dimg, "dimg")
    B, C, S, _ = dimg.shape
    dts = torch.empty((B, L, C), dtype=torch.float32, device=dimg.device)
    _lib.call("i2r_ts_to_img_bwd_f32", _p(dimg), _p(dts), B, L, C, delay, embedding, cols, S, _stream())
    return dts


def img_to_ts(img, L, delay, embedding, cols):
    if img.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"img_to_ts: expected float32/float64, got {img.dtype}")
    _chk(img, "img", img.dtype)
    B, C, S, S2 = img.shape
    if S != S2:
        raise ValueError("img_to_ts: image must be square")
    ts = torch.empty((B, L, C), dtype=torch.float32, device=img.device)
    fn = "i2r_img_to_ts_f64" if img.dtype == torch.float64 else "i2r_img_to_ts_f32"
    _lib.call(fn, _p(img), _p(ts), B, L, C, delay, embedding, cols, S, _stream())
    return ts


def conv2d_pack(w, dgrad=False):
    """OIHW weights -> the K-major packed layout the conv kernels read (transposed + flipped for dgrad)."""
    _chk(w, "w")
    Cout, Cin, k, k2 = w.shape
    if k != k2 or k not in (1, 3):
        raise ValueError(f"conv2d: kernel {k}x{k2} unsupported")
    n = _lib.query("i2r_conv2d_packed_floats", Cout, Cin, k, int(dgrad))
    wp = torch.empty((n,), dtype=torch.float32, device=w.device)
    _lib.call("i2r_conv2d_pack_weight_f32", _p(w), _p(wp), Cout, Cin, k, int(dgrad), _stream())
    return wp


def pack_multi(table, blockmap, nblocks):
    _lib.call("i2r_conv2d_pack_multi_f32", _p(table), _p(blockmap), int(nblocks), _stream())


def pack_h2_multi(table, blockmap, nblocks):
    _lib.call("i2r_conv2d_pack_h2_multi", _p(table), _p(blockmap), int(nblocks), _stream())


def _conv_packed(x, wp, bias, residual, Cout, k):
    B, Cin, H, W = x.shape
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    if residual is not None:
        _chk(residual, "residual")
        if residual.shape != out.shape:
            raise ValueError("conv2d: residual shape mismatch")
    nsk = _lib.query("i2r_conv2d_splitk_floats", B, Cin, H, W, Cout, k)
    ws = torch.empty((nsk,), dtype=torch.float32, device=x.device) if nsk > 0 else None
    _lib.call("i2r_conv2d_fwd_packed_f32", _p(x), _p(wp), _p(bias), _p(residual), _p(out), _p(ws), nsk, B, Cin, H, W,
              Cout, k, _stream())
    return out


def conv2d_fwd(x, w, bias=None, residual=None, wp=None):
    """conv2d(x, w, pad=k//2) + bias + residual.  ``wp``: weights already packed by conv2d_pack (cached by Conv2d)."""
    _chk(x, "x"); _chk(w, "w")
    Cout, Cin2, k, k2 = w.shape
    if Cin2 != x.shape[1] or k != k2:
        raise ValueError(f"conv2d: weight {tuple(w.shape)} does not match input {tuple(x.shape)}")
    if bias is not None:
        _chk(bias, "bias")
    if wp is None:
        wp = conv2d_pack(w)
    return _conv_packed(x, wp, bias, residual, Cout, k)


def conv2d_pack_h2(w):
    """OIHW fp32 weights -> split-f16 (hi | lo, scaled 2^6) packed layout of the f16x2 inference conv kernel."""
    _chk(w, "w")
    Cout, Cin, k, k2 = w.shape
    if k != k2 or k not in (1, 3):
        raise ValueError(f"conv2d: kernel {k}x{k2} unsupported")
    n = _lib.query("i2r_conv2d_packed_h2_bytes", Cout, Cin, k)
    wp = torch.empty((n,), dtype=torch.uint8, device=w.device)
    _lib.call("i2r_conv2d_pack_weight_h2", _p(w), _p(wp), Cout, Cin, k, _stream())
    return wp


def conv2d_pack_h2_dgrad(w):
    """OIHW fp32 weights -> split-f16 packed layout of the data-gradient problem (transposed, taps flipped)."""
    _chk(w, "w")
    Cout, Cin, k, k2 = w.shape
    if k != k2 or k not in (1, 3):
        raise ValueError(f"conv2d: kernel {k}x{k2} unsupported")
    n = _lib.query("i2r_conv2d_packed_h2_bytes", Cin, Cout, k)
    wp = torch.empty((n,), dtype=torch.uint8, device=w.device)
    _lib.call("i2r_conv2d_pack_weight_h2_dgrad", _p(w), _p(wp), Cout, Cin, k, _stream())
    return wp


_SCALE_SCRATCH = {}


def _scale_scratch(device):
    """Two zeroed uint32 per device for i2r_absmax_pow2_scale_f32 (the kernel leaves them zero).  Made outside any
    stream capture: the first data-gradient call of a process is always an eager one (graph warm-up)."""
    t = _SCALE_SCRATCH.get(device)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.I2RError("conv2d_dgrad_h2: first use inside a stream capture; run one eager step first")
        t = _SCALE_SCRATCH[device] = torch.zeros(2, dtype=torch.int32, device=device)
    return t


def absmax_pow2_scale(x):
    """Device float[2] {s, 1/s}: s the power of two that puts max|x| into [2^13, 2^14)."""
    _chk(x, "x")
    scale = torch.empty((2,), dtype=torch.float32, device=x.device)
    _lib.call("i2r_absmax_pow2_scale_f32", _p(x), x.numel(), _p(_scale_scratch(x.device)), _p(scale), _stream())
    return scale


def conv2d_dgrad_h2(dy, w_shape, wpd):
    """dx = conv2d(dy, flip(w)^T) on the f16x2-split kernel at fp32-grade accuracy; ``wpd`` from conv2d_pack_h2_dgrad."""
    _chk(dy, "dy")
    Cout, Cin, k, _ = w_shape
    B, Cout2, H, W = dy.shape
    if Cout2 != Cout:
        raise ValueError("conv2d_dgrad_h2: channel mismatch")
    dx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dy.device)
    scale = absmax_pow2_scale(dy)
    nsk = _lib.query("i2r_conv2d_splitk_h2_floats", B, Cout, H, W, Cin, k)
    ws = torch.empty((nsk,), dtype=torch.float32, device=dy.device) if nsk > 0 else None
    _lib.call("i2r_conv2d_dgrad_h2", _p(dy), _p(wpd), _p(dx), _p(scale), _p(ws), nsk, B, Cin, H, W, Cout, k, _stream())
    return dx


def conv2d_fwd_h2(x, w_shape, wp, bias=None, residual=None, x_split=False, out=None, qkv_out=False, up_in=False,
                  up_res=False):
    """Inference conv on the f16 matrix cores with fp32-grade accuracy (csrc/conv_fwd_h2.hip).  ``x_split``: x holds
    pre-split f16 pairs (groupnorm_fwd(..., split_out=True)).  ``out``: write into this (B, Cout, H, W) tensor.
    ``qkv_out``: store the result as split pairs with the q rows pre-scaled, for attention_fwd(presplit=True).
    ``up_in`` / ``up_res``: x / the residual are at half resolution and are nearest-upsampled x2 on the fly (the
    ``up=True`` convs and the weight-less skip of the decoder's up blocks, networks.py:79-81, :156-159)."""
    _chk(x, "x")
    Cout, Cin, k, _ = w_shape
    B, Cin2, H, W = x.shape
    if up_in:
        H, W = 2 * H, 2 * W
    if Cin2 != Cin:
        raise ValueError("conv2d_fwd_h2: channel mismatch")
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    elif tuple(_chk(out, "out").shape) != (B, Cout, H, W):
        raise ValueError("conv2d_fwd_h2: out has the wrong shape")
    if residual is not None:
        _chk(residual, "residual")
        want = (B, Cout, H // 2, W // 2) if up_res else (B, Cout, H, W)
        if tuple(residual.shape) != want:
            raise ValueError(f"conv2d_fwd_h2: residual shape {tuple(residual.shape)} != {want}")
    nsk = _lib.query("i2r_conv2d_splitk_h2_floats", B, Cin, H, W, Cout, k)
    ws = torch.empty((nsk,), dtype=torch.float32, device=x.device) if nsk > 0 else None
    flags = (int(bool(x_split)) | (2 if qkv_out else 0) | (4 if up_in else 0) | (8 if up_res else 0)
             | (16 if _CONV_MATH[0] == "f16" else 0))
    _lib.call("i2r_conv2d_fwd_h2", _p(x), _p(wp), _p(bias), _p(residual), _p(out), _p(ws), nsk, B, Cin, H, W, Cout, k,
              flags, _stream())
    return out


def conv1x1_cat_fwd_h2(x1, x2, w_shape, wp, bias=None, residual=None):
    """1x1 conv over torch.cat([x1, x2], 1) read in place (f16x2 inference kernel)."""
    _chk(x1, "x1"); _chk(x2, "x2")
    Cout, Cin, k, _ = w_shape
    B, C1, H, W = x1.shape
    C2 = x2.shape[1]
    if k != 1 or C1 + C2 != Cin or x2.shape[0] != B or tuple(x2.shape[2:]) != (H, W):
        raise ValueError("conv1x1_cat_fwd_h2: shape mismatch")
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x1.device)
    if residual is not None:
        _chk(residual, "residual")
    nsk = _lib.query("i2r_conv2d_splitk_h2_floats", B, Cin, H, W, Cout, 1)
    ws = torch.empty((nsk,), dtype=torch.float32, device=x1.device) if nsk > 0 else None
    _lib.call("i2r_conv1x1_cat_fwd_h2", _p(x1), _p(x2), C1, C2, _p(wp), _p(bias), _p(residual), _p(out), _p(ws), nsk,
              B, H, W, Cout, _stream())
    return out


def groupnorm_cat_fwd(x1, x2, gamma, beta, groups, eps, ss=None, act=False, split_out=False):
    """GroupNorm (+modulation, SiLU, split output) over torch.cat([x1, x2], 1) read in place."""
    _chk(x1, "x1"); _chk(x2, "x2"); _chk(gamma, "gamma"); _chk(beta, "beta")
    B, C1 = x1.shape[0], x1.shape[1]
    C2 = x2.shape[1]
    HW = x1.numel() // (B * C1)
    ss_batch = 0
    if ss is not None:
        _chk(ss, "scale_shift")
        ss_batch = ss.shape[0]
    y = torch.empty((B, C1 + C2) + tuple(x1.shape[2:]), dtype=torch.float32, device=x1.device)
    _lib.call("i2r_groupnorm_cat_fwd_f32", _p(x1), _p(x2), C1, C2, _p(gamma), _p(beta), _p(ss), ss_batch, _p(y), B, HW,
              groups, float(eps), int(bool(act)), int(bool(split_out)), _stream())
    return y


def conv2d_dgrad(dy, w, wpd=None):
    _chk(dy, "dy"); _chk(w, "w")
    Cout, Cin, k, _ = w.shape
    assert dy.shape[1] == Cout
    return _conv_packed(dy, wpd if wpd is not None else conv2d_pack(w, dgrad=True), None, None, Cin, k)


def conv2d_wgrad(x, dy, ksize, want_bias=True, dw=None, db=None, accumulate=False):
    """dw / db: optional preallocated outputs (gradient slots); accumulate: add into them (they were zeroed)."""
    _chk(x, "x"); _chk(dy, "dy")
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if dw is None:
        dw = torch.empty((Cout, Cin, ksize, ksize), dtype=torch.float32, device=x.device)
        accumulate = False
    if want_bias and db is None:
        if accumulate:
            db = torch.zeros((Cout,), dtype=torch.float32, device=x.device)
        else:
            db = torch.empty((Cout,), dtype=torch.float32, device=x.device)
    if deterministic():
        n = _lib.query("i2r_conv2d_wgrad_det_floats", B, Cin, H, W, Cout, ksize)
        ws = torch.empty((n,), dtype=torch.float32, device=x.device)
        _lib.call("i2r_conv2d_wgrad_det_f32", _p(x), _p(dy), _p(dw), _p(db) if want_bias else None, _p(ws), n, B, Cin,
                  H, W, Cout, ksize, int(bool(accumulate)), _stream())
        return dw, (db if want_bias else None)
    _lib.call("i2r_conv2d_wgrad_f32", _p(x), _p(dy), _p(dw), _p(db) if want_bias else None, B, Cin, H, W, Cout,
              ksize, int(bool(accumulate)), _stream())
    return dw, (db if want_bias else None)


def down2(x, scale=0.25):
    _chk(x, "x")
    B, C, H, W = x.shape
    y = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.call("i2r_down2_f32", _p(x), _p(y), B * C, H, W, float(scale), _stream())
    return y


def up2(x, scale=1.0):
    _chk(x, "x")
    B, C, H, W = x.shape
    y = torch.empty((B, C, H * 2, W * 2), dtype=torch.float32, device=x.device)
    _lib.call("i2r_up2_f32", _p(x), _p(y), B * C, H, W, float(scale), _stream())
    return y


def concat_channels(a, c):
    _chk(a, "a"); _chk(c, "c")
    B, C1, H, W = a.shape
    C2 = c.shape[1]
    out = torch.empty((B, C1 + C2, H, W), dtype=torch.float32, device=a.device)
    _lib.call("i2r_concat_channels_f32", _p(a), _p(c), _p(out), B, C1 * H * W, C2 * H * W, _stream())
    return out


def split_channels(x, C1):
    _chk(x, "x")
    B, C, H, W = x.shape
    a = torch.empty((B, C1, H, W), dtype=torch.float32, device=x.device)
    c = torch.empty((B, C - C1, H, W), dtype=torch.float32, device=x.device)
    _lib.call("i2r_split_channels_f32", _p(x), _p(a), _p(c), B, C1 * H * W, (C - C1) * H * W, 0, _stream())
    return a, c


def groupnorm_fwd(x, gamma, beta, groups, eps, ss=None, act=False, save_stats=False, drop_p=0.0, seed=0, offset=0,
                  step=None, split_out=False):
    _chk(x, "x"); _chk(gamma, "gamma"); _chk(beta, "beta")
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    ss_batch = 0
    if ss is not None:
        _chk(ss, "scale_shift")
        if ss.shape[-1] != 2 * C:
            raise ValueError("groupnorm: scale/shift must have 2*C columns")
        ss_batch = ss.shape[0]
    y = torch.empty_like(x)
    mean = rstd = None
    if save_stats:
        mean = torch.empty((B * groups,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    _lib.call("i2r_groupnorm_fwd_f32", _p(x), _p(gamma), _p(beta), _p(ss), ss_batch, _p(y), _p(mean), _p(rstd), B, C,
              HW, groups, float(eps), int(bool(act)), float(drop_p), int(seed), int(offset), _p(step),
              int(bool(split_out)), _stream())
    return y, mean, rstd


def groupnorm_bwd(x, gamma, beta, ss, mean, rstd, dy, groups, act, drop_p=0.0, seed=0, offset=0, step=None,
                  dgamma=None, dbeta=None, dss=None, add=None):
    """``add``: a (B, C, H, W) gradient added to dx in the same pass (the block's bypass of x)."""
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    _chk(dy, "dy")
    if add is not None and (tuple(_chk(add, "add").shape) != tuple(x.shape)):
        raise ValueError("groupnorm_bwd: add has the wrong shape")
    dx = torch.empty_like(x)
    if ss is None:
        dss = None
    elif dss is None or tuple(dss.shape) != (B, 2 * C) or not dss.is_contiguous():
        dss = torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)
    if dgamma is not None and dbeta is not None and slots_zeroed() and not deterministic():
        # the step's gradient slots were zero-filled: accumulate over the batch in place (no partials, no reduction)
        _lib.call("i2r_groupnorm_bwd_acc_f32", _p(x), _p(gamma), _p(beta), _p(ss), 0 if ss is None else ss.shape[0],
                  _p(mean), _p(rstd), _p(dy), _p(dx), _p(dgamma), _p(dbeta), _p(dss), _p(add), B, C, HW, groups,
                  int(bool(act)), float(drop_p), int(seed), int(offset), _p(step), _stream())
        return dx, dgamma, dbeta, dss
    part = torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)  # per-sample [dgamma | dbeta]
    _lib.call("i2r_groupnorm_bwd_f32", _p(x), _p(gamma), _p(beta), _p(ss), 0 if ss is None else ss.shape[0], _p(mean),
              _p(rstd), _p(dy), _p(dx), _p(part), _p(dss), _p(add), B, C, HW, groups, int(bool(act)),
              float(drop_p), int(seed), int(offset), _p(step), _stream())
    if dgamma is not None and dbeta is not None:  # straight into the parameters' gradient slots
        _lib.call("i2r_reduce_rows2_f32", _p(part), _p(dgamma), _p(dbeta), B, C, C, _stream())
        return dx, dgamma, dbeta, dss
    dgb = torch.empty((2 * C,), dtype=torch.float32, device=x.device)
    _lib.call("i2r_reduce_rows_f32", _p(part), _p(dgb), B, 2 * C, _stream())
    return dx, dgb[:C], dgb[C:], dss


def attention_h2_ok(T) -> bool:
    """Shapes the f16x2 inference attention kernel covers (others take the fp32 kernels)."""
    return h2_math() and T > 32 and T % 32 == 0


def attention_fwd(qkv, heads, save_lse=False, presplit=False):
    _chk(qkv, "qkv")
    B, C3 = qkv.shape[0], qkv.shape[1]
    C = C3 // 3
    if C != heads * 64:
        raise ValueError(f"attention: channels {C} != heads*64 ({heads}*64); head dim is fixed at 64")
    T = qkv.numel() // (B * C3)
    out = torch.empty((B, C) + tuple(qkv.shape[2:]), dtype=torch.float32, device=qkv.device)
    if not save_lse and attention_h2_ok(T):
        _lib.call("i2r_attention_fwd_h2", _p(qkv), _p(out), B, heads, T, int(bool(presplit)), _stream())
        return out, None  # inference: f16x2-split MFMA
    if presplit:
        raise ValueError("attention_fwd: presplit qkv is only understood by the f16x2 inference kernel")
    lse = torch.empty((B * heads * T,), dtype=torch.float32, device=qkv.device) if save_lse else None
    if save_lse and T > 32 and T % 32 == 0 and train_fwd_h2_enabled():
        # training forward on the f16x2 kernel (like the conv forwards; the backward stays on the exact fp32 kernels)
        _lib.call("i2r_attention_fwd_h2_lse", _p(qkv), _p(out), _p(lse), B, heads, T, _stream())
        return out, lse
    _lib.call("i2r_attention_fwd_f32", _p(qkv), _p(out), _p(lse), B, heads, T, _stream())
    return out, lse


# Attention BACKWARD on the f16x2-split kernels (i2r_attention_bwd_h2): sequences of 64 positions and more (T % 64
# == 0), whenever training forwards may run on f16x2 kernels at all -- a non-finite gradient switches both back to the
# exact fp32 kernels (GraphedTrainStep.check_range).  I2R_ATTN_BWD_H2=0: always the fp32 kernels.
_ATTN_BWD_H2 = [os.environ.get("I2R_ATTN_BWD_H2", "auto")]
_ATTN_BWD_H2_MINT = int(os.environ.get("I2R_ATTN_BWD_H2_MINT", 64))  # A/B: 128 keeps T = 64 on the fp32 kernels


def set_attn_bwd_h2(mode) -> None:
    mode = {True: "auto", False: "0"}.get(mode, mode)
    if mode not in ("auto", "0"):
        raise ValueError(mode)
    _ATTN_BWD_H2[0] = mode


def attn_bwd_h2(T) -> bool:
    return _ATTN_BWD_H2[0] != "0" and train_fwd_h2_enabled() and T >= _ATTN_BWD_H2_MINT and T % 64 == 0


def attention_bwd(qkv, out, lse, dout, heads):
    _chk(dout, "dout")
    B, C3 = qkv.shape[0], qkv.shape[1]
    T = qkv.numel() // (B * C3)
    dqkv = torch.empty_like(qkv)
    ws = torch.empty((B * heads * T,), dtype=torch.float32, device=qkv.device)
    if attn_bwd_h2(T) and qkv.data_ptr() % 16 == 0 and dout.data_ptr() % 16 == 0:
        _lib.call("i2r_attention_bwd_h2", _p(qkv), _p(out), _p(lse), _p(dout), _p(dqkv), _p(ws),
                  _p(absmax_pow2_scale(dout)), B, heads, T, _stream())
        return dqkv
    _lib.call("i2r_attention_bwd_f32", _p(qkv), _p(out), _p(lse), _p(dout), _p(dqkv), _p(ws), B, heads, T, _stream())
    return dqkv


def pos_embedding(x, num_channels, max_positions=10000):
    _chk(x, "noise_labels")
    out = torch.empty((x.numel(), num_channels), dtype=torch.float32, device=x.device)
    _lib.call("i2r_pos_embedding_f32", _p(x), _p(out), x.numel(), num_channels, max_positions, _stream())
    return out


def linear_fwd(x, w, bias=None, act=False, save_pre=False):
    _chk(x, "x"); _chk(w, "w")
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError("linear: shape mismatch")
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if (save_pre and act) else None
    _lib.call("i2r_linear_fwd_f32", _p(x), _p(w), _p(bias), _p(y), _p(pre), M, K, N, int(bool(act)), _stream())
    return y, pre


def linear_bwd(x, w, dy, need_dx=True, need_dw=True, need_db=False, dw=None, db=None):
    M, K = x.shape
    N = w.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    if dw is None:
        dw = torch.empty_like(w) if (need_dw or need_db) else None
    if db is None:
        db = torch.empty((N,), dtype=torch.float32, device=x.device) if need_db else None
    _lib.call("i2r_linear_bwd_f32", _p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), M, K, N, _stream())
    return dx, dw, db


def silu_bwd(pre, dy):
    dx = torch.empty_like(dy)
    _lib.call("i2r_silu_bwd_f32", _p(pre), _p(dy), _p(dx), dy.numel(), _stream())
    return dx


def reduce_rows(x2d):
    _chk(x2d, "x")
    out = torch.empty((x2d.shape[1],), dtype=torch.float32, device=x2d.device)
    _lib.call("i2r_reduce_rows_f32", _p(x2d), _p(out), x2d.shape[0], x2d.shape[1], _stream())
    return out


def add(a, b):
    _chk(a, "a"); _chk(b, "b")
    y = torch.empty_like(a)
    _lib.call("i2r_add_f32", _p(a), _p(b), _p(y), a.numel(), _stream())
    return y


def _sigma_vec(sigma, device):
    s = sigma.to(device=device, dtype=torch.float32).reshape(-1)
    return _c(s)


def edm_scale_input(x, sigma, sigma_data):
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float32)
    x = _c(x)
    _chk(x, "x", x.dtype)
    B = x.shape[0]
    chw = x.numel() // B
    xin = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    c_noise = torch.empty((sigma.numel(),), dtype=torch.float32, device=x.device)
    _lib.call("i2r_edm_scale_input", _p(x), int(x.dtype == torch.float64), _p(sigma), sigma.numel(), _p(xin),
              _p(c_noise), B, chw, float(sigma_data), _stream())
    return xin, c_noise


def edm_combine(x, F, sigma, sigma_data):
    _chk(F, "F")
    B = x.shape[0]
    chw = x.numel() // B
    D = torch.empty_like(F)
    _lib.call("i2r_edm_combine", _p(x), int(x.dtype == torch.float64), _p(F), _p(sigma), sigma.numel(), _p(D), B, chw,
              float(sigma_data), _stream())
    return D


def edm_combine_bwd(dD, sigma, sigma_data):
    _chk(dD, "dD")
    B = dD.shape[0]
    dF = torch.empty_like(dD)
    _lib.call("i2r_edm_combine_bwd", _p(dD), _p(sigma), sigma.numel(), _p(dF), B, dD.numel() // B, float(sigma_data),
              _stream())
    return dF


def edm_noise(x, mask, P_mean, P_std, sigma_data, rnd=None, noise=None, seed=0, offset=0, step=None):
    _chk(x, "x"); _chk(mask, "mask")
    B = x.shape[0]
    chw = x.numel() // B
    xn = torch.empty_like(x)
    sigma = torch.empty((B,), dtype=torch.float32, device=x.device)
    weight = torch.empty_like(sigma)
    if rnd is not None:
        rnd = _c(rnd.to(device=x.device, dtype=torch.float32).reshape(-1))
        assert rnd.numel() == B
    if noise is not None:
        noise = _chk(_c(noise.to(device=x.device, dtype=torch.float32)), "noise")
        assert noise.numel() == x.numel()
    _lib.call("i2r_edm_noise", _p(x), _p(mask), _p(rnd), _p(noise), _p(xn), _p(sigma), _p(weight), B, chw,
              float(P_mean), float(P_std), float(sigma_data), int(seed), int(offset), _p(step), _stream())
    return xn, sigma, weight


def nan_mask(x):
    _chk(x, "x")
    xc = torch.empty_like(x)
    mask = torch.empty_like(x)
    _lib.call("i2r_nan_mask_f32", _p(x), _p(xc), _p(mask), x.numel(), _stream())
    return xc, mask


def masked_edm_loss(D, x, mask, weight, want_grad=True):
    _chk(D, "D"); _chk(x, "x"); _chk(mask, "mask"); _chk(weight, "weight")
    B = D.shape[0]
    loss = torch.empty((), dtype=torch.float32, device=D.device)
    dD = torch.empty_like(D) if want_grad else None
    partials = torch.empty((1024,), dtype=torch.float32, device=D.device)
    _lib.call("i2r_masked_edm_loss", _p(D), _p(x), _p(mask), _p(weight), _p(loss), _p(dD), _p(partials), 1024, B,
              D.numel() // B, _stream())
    return loss, dD


def heun_init(latents, t0):
    """x_0 = latents.double() * t_0 (sampler.py:45); fp64 latents are multiplied unrounded."""
    if latents.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"heun_init: expected float32/float64 latents, got {latents.dtype}")
    _chk(latents, "latents", latents.dtype)
    x = torch.empty(latents.shape, dtype=torch.float64, device=latents.device)
    fn = "i2r_heun_init_f64_from_f64" if latents.dtype == torch.float64 else "i2r_heun_init_f64"
    _lib.call(fn, _p(latents), _p(x), latents.numel(), float(t0), _stream())
    return x


def heun_euler(x_hat, D, d_cur, x_next, t_hat, t_next):
    _lib.call("i2r_heun_euler_f64", _p(x_hat), _p(D), _p(d_cur), _p(x_next), x_hat.numel(), float(t_hat),
              float(t_next), _stream())


def heun_correct(x_hat, x_next, Dp, d_cur, x_out, t_hat, t_next):
    _lib.call("i2r_heun_correct_f64", _p(x_hat), _p(x_next), _p(Dp), _p(d_cur), _p(x_out), x_hat.numel(),
              float(t_hat), float(t_next), _stream())


# ======================================================================================================= autograd
class Conv2dFn(torch.autograd.Function):
    """conv2d + bias + residual (networks.py:87-89 and the x.add_ of :178/:185)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, wp=None, wpd=None, wph=None, wpd_h2=None):
        x = _c(x)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.wpd = wpd  # weights already packed for the data gradient (DhariwalUNet.pack_all_weights), or None
        ctx.wpd_h2 = wpd_h2  # split-f16 data-gradient weights: dx then runs on the f16x2 kernel (large layers)
        ctx.slots = (param_slot(w), param_slot(bias))
        if wph is not None:  # forward on the f16x2-split kernel (~1e-6); the gradients below use the exact kernels
            return conv2d_fwd_h2(x, tuple(w.shape), wph, bias, None if residual is None else _c(residual))
        return conv2d_fwd(x, w, bias, None if residual is None else _c(residual), wp)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_dgrad_h2(dy, tuple(w.shape), ctx.wpd_h2) if ctx.wpd_h2 is not None else \
                conv2d_dgrad(dy, w, ctx.wpd)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            sw, sb = ctx.slots
            use = sw is not None and (sb is not None or not ctx.has_bias)
            dw, db = conv2d_wgrad(x, dy, w.shape[-1], want_bias=ctx.has_bias, dw=sw.detach() if use else None,
                                  db=sb.detach() if (use and ctx.has_bias) else None,
                                  accumulate=use and slots_zeroed())
        dres = dy if ctx.needs_input_grad[3] else None
        return dx, dw, db, dres, None, None, None, None


class Down2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return down2(_c(x), 0.25)

    @staticmethod
    def backward(ctx, dy):
        return up2(_c(dy), 0.25)


class Up2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return up2(_c(x), 1.0)

    @staticmethod
    def backward(ctx, dy):
        return down2(_c(dy), 1.0)


class ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, c):
        ctx.c1 = a.shape[1]
        return concat_channels(_c(a), _c(c))

    @staticmethod
    def backward(ctx, dy):
        return split_channels(_c(dy), ctx.c1)


class GroupNormFn(torch.autograd.Function):
    """gn -> (1+scale)*.+shift -> SiLU -> dropout, one kernel each way."""

    @staticmethod
    def forward(ctx, x, gamma, beta, ss, groups, eps, act, drop_p, seed, offset, bypass=False):
        """``bypass``: also return x itself as a second output.  The block routes its residual / skip branch through
        that alias, so x has ONE consumer in the autograd graph and the two gradients are summed inside the backward
        kernel (``add``) instead of by a separate accumulation kernel."""
        x = _c(x)
        need_bwd = any(ctx.needs_input_grad[:4])
        dss_slot = getattr(ss, "_dss_slot", None)  # AffineAllFn's dy slab piece for this layer, if grouped
        if ss is not None:
            ss = _c(ss)
        step = rng_step() if drop_p > 0.0 else None
        y, mean, rstd = groupnorm_fwd(x, gamma, beta, groups, eps, ss, act, need_bwd, drop_p, seed, offset, step)
        if need_bwd:
            ctx.save_for_backward(x, gamma, beta, ss, mean, rstd)
            ctx.cfg = (groups, act, drop_p, seed, offset, step)
            ctx.slots = (param_slot(gamma), param_slot(beta))
            ctx.dss_slot = dss_slot
        return (y, x.view_as(x)) if bypass else y

    @staticmethod
    def backward(ctx, dy, dbypass=None):
        x, gamma, beta, ss, mean, rstd = ctx.saved_tensors
        groups, act, drop_p, seed, offset, step = ctx.cfg
        if ss is not None and ss.shape[0] != x.shape[0]:
            raise _lib.I2RError("groupnorm backward needs per-sample scale/shift (sigma given per sample)")
        sg, sb = ctx.slots
        use = sg is not None and sb is not None
        dx, dg, db, dss = groupnorm_bwd(x, gamma, beta, ss, mean, rstd, _c(dy), groups, act, drop_p, seed, offset,
                                        step, sg.detach() if use else None, sb.detach() if use else None,
                                        dss=ctx.dss_slot, add=None if dbypass is None else _c(dbypass))
        return dx, dg, db, dss, None, None, None, None, None, None, None


class AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = _c(qkv)
        need_bwd = ctx.needs_input_grad[0]
        out, lse = attention_fwd(qkv, heads, save_lse=need_bwd)
        if need_bwd:
            ctx.save_for_backward(qkv, out, lse)
            ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        return attention_bwd(qkv, out, lse, _c(dout), ctx.heads), None


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, act):
        x = _c(x)
        need_bwd = any(ctx.needs_input_grad[:3])
        y, pre = linear_fwd(x, w, bias, act, save_pre=need_bwd)
        if need_bwd:
            ctx.save_for_backward(x, w, pre)
            ctx.has_bias = bias is not None
            ctx.slots = (param_slot(w), param_slot(bias))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, pre = ctx.saved_tensors
        dy = _c(dy)
        if pre is not None:
            dy = silu_bwd(pre, dy)
        sw, sb = ctx.slots
        dx, dw, db = linear_bwd(x, w, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                ctx.has_bias and ctx.needs_input_grad[2],
                                dw=None if sw is None else sw.detach(), db=None if sb is None else sb.detach())
        return dx, dw, db, None


class AffinePlan:
    """Static launch tables of the grouped `affine` products for one set of Linear layers and one batch size M."""

    def __init__(self, linears, M):
        dev = linears[0].weight.device
        K = linears[0].weight.shape[1]
        self.M, self.K, self.L = M, K, len(linears)
        self.N = [l.weight.shape[0] for l in linears]
        self.off = [0]
        for n in self.N:
            self.off.append(self.off[-1] + M * n)
        self.slab_floats = self.off[-1]
        self.part_floats = self.L * M * K
        rows = []
        self.has_slots = all(getattr(l.weight, "_grad_slot", None) is not None and
                             getattr(l.bias, "_grad_slot", None) is not None for l in linears)
        for l, n, o in zip(linears, self.N, self.off):
            dw = l.weight._grad_slot.data_ptr() if self.has_slots else 0
            db = l.bias._grad_slot.data_ptr() if self.has_slots else 0
            rows.append([l.weight.data_ptr(), l.bias.data_ptr(), dw, db, n, o])
        self.slots = [(l.weight._grad_slot, l.bias._grad_slot) for l in linears] if self.has_slots else None
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        mt = (M + 31) // 32
        fwd, dw_, dx = [], [], []
        for i, n in enumerate(self.N):
            fwd += [[i, bx, by, 0] for by in range(mt) for bx in range((n + 31) // 32)]
            dw_ += [[i, bx, by, 0] for by in range((n + 31) // 32) for bx in range((K + 31) // 32)]
            dx += [[i, bx, by, self.slab_floats + i * M * K] for by in range(mt) for bx in range((K + 31) // 32)]
        self.maps = [torch.tensor(m, dtype=torch.int32).to(dev) for m in (fwd, dw_, dx)]
        self.key = AffinePlan.make_key(linears, M)

    @staticmethod
    def make_key(linears, M):
        return (M,) + tuple((l.weight.data_ptr(), l.bias.data_ptr(),
                             getattr(l.weight, "_grad_slot", l.weight).data_ptr()) for l in linears)

    def launch(self, mode, emb, base):
        m = self.maps[mode]
        _lib.call("i2r_gemm_grouped_f32", _p(self.table), _p(m), m.shape[0], mode, _p(emb), _p(base), self.M, self.K,
                  _stream())


class AffineAllFn(torch.autograd.Function):
    """Every UNetBlock's ``affine(emb)`` (networks.py:170) at once: forward = one grouped launch into one slab whose
    per-layer pieces are returned; backward = one launch for all dW/db (into the gradient slots), one for the per-layer
    dx partials and one reduction.  The GroupNorm backward kernels write their d[scale|shift] straight into this
    function's dy slab (``_dss_slot`` on each returned piece), so nothing is copied or accumulated per layer."""

    @staticmethod
    def forward(ctx, emb, plan, *params):
        emb = _c(emb)
        slab = torch.empty((plan.slab_floats,), dtype=torch.float32, device=emb.device)
        plan.launch(0, emb, slab)
        outs = tuple(slab[plan.off[i]: plan.off[i + 1]].view(plan.M, n) for i, n in enumerate(plan.N))
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(emb)
            ctx.plan = plan
            # dy slab + room for the dx partials behind it; the GroupNorm backwards fill the pieces
            ctx.dys = torch.empty((plan.slab_floats + plan.part_floats,), dtype=torch.float32, device=emb.device)
        return outs

    @staticmethod
    def dss_slots(ctx_dys, plan):
        return [ctx_dys[plan.off[i]: plan.off[i + 1]].view(plan.M, n) for i, n in enumerate(plan.N)]

    @staticmethod
    def backward(ctx, *grads):
        (emb,) = ctx.saved_tensors
        plan, dys = ctx.plan, ctx.dys
        for i, (g, slot) in enumerate(zip(grads, AffineAllFn.dss_slots(dys, plan))):
            if g is None:
                slot.zero_()
            elif g.data_ptr() != slot.data_ptr():
                slot.copy_(g)
        if not plan.has_slots:
            raise _lib.I2RError("grouped affine backward needs gradient slots (FusedAdamW / GradBucketReducer)")
        plan.launch(1, emb, dys)
        dx = None
        if ctx.needs_input_grad[0]:
            plan.launch(2, emb, dys)
            part = dys[plan.slab_floats:].view(plan.L, plan.M * plan.K)
            dx = reduce_rows(part).view(plan.M, plan.K)
        out = [dx, None]
        for sw, sb in plan.slots:  # aliases of the slots: autograd adopts them as .grad without a copy
            out += [sw.detach(), sb.detach()]
        return tuple(out)


class EdmCombineFn(torch.autograd.Function):
    """D = c_skip x + c_out F (networks.py:667).  Only dF is propagated: the reference computes dL/dx too but
    discards it, so the input image is treated as a constant."""

    @staticmethod
    def forward(ctx, x, F, sigma, sigma_data):
        ctx.save_for_backward(sigma)
        ctx.sd = sigma_data
        return edm_combine(x, _c(F), sigma, sigma_data)

    @staticmethod
    def backward(ctx, dD):
        (sigma,) = ctx.saved_tensors
        return None, edm_combine_bwd(_c(dD), sigma, ctx.sd), None, None


class MaskedLossFn(torch.autograd.Function):
    """mean(weight * (D*mask - x*mask)^2) (our.py:64-66); forward also produces dL/dD."""

    @staticmethod
    def forward(ctx, D, x, mask, weight):
        loss, dD = masked_edm_loss(_c(D), x, mask, weight, want_grad=ctx.needs_input_grad[0])
        if dD is not None:
            ctx.save_for_backward(dD)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dD,) = ctx.saved_tensors
        return dD * g, None, None, None


class TsToImgFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ts, delay, embedding, cols, side, pad_value):
        ctx.cfg = (ts.shape[1], delay, embedding, cols)
        return ts_to_img(_c(ts), delay, embedding, cols, side, pad_value)

    @staticmethod
    def backward(ctx, dimg):
        L, delay, embedding, cols = ctx.cfg
        return ts_to_img_bwd(_c(dimg), L, delay, embedding, cols), None, None, None, None, None
