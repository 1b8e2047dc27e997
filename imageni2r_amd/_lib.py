"""ctypes binding of libi2r_hip.so (the C ABI declared in include/i2r_hip.h).

There is deliberately NO fallback: if the shared library is missing, or a tensor is not a contiguous HIP ("cuda")
tensor of the expected dtype, the call raises.  Build the library with ``python -c "import __graft_entry__ as g;
g.build()"`` (or ``make -C imageni2r_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_long, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("I2R_LIB_PATH") or os.path.join(_HERE, "csrc", "libi2r_hip.so")  # override: tuning builds

P, I, L, F, D, U = c_void_p, c_int, c_long, c_float, c_double, c_uint64

# name -> argument ctypes (all return int except i2r_last_error).  Must mirror include/i2r_hip.h.
SIGNATURES = {
    "i2r_version": [],
    "i2r_ts_to_img_f32": [P, P, P, I, I, I, I, I, I, I, F, P],
    "i2r_ts_to_img_bwd_f32": [P, P, I, I, I, I, I, I, I, P],
    "i2r_img_to_ts_f32": [P, P, I, I, I, I, I, I, I, P],
    "i2r_img_to_ts_f64": [P, P, I, I, I, I, I, I, I, P],
    "i2r_conv2d_packed_floats": [I, I, I, I],
    "i2r_conv2d_pack_weight_f32": [P, P, I, I, I, I, P],
    "i2r_conv2d_pack_block_floats": [],
    "i2r_conv2d_pack_multi_f32": [P, P, I, P],
    "i2r_conv2d_splitk_floats": [I, I, I, I, I, I],
    "i2r_conv2d_fwd_packed_f32": [P, P, P, P, P, P, L, I, I, I, I, I, I, P],
    "i2r_conv2d_fwd_f32": [P, P, P, P, P, P, L, I, I, I, I, I, I, P],
    "i2r_conv2d_dgrad_f32": [P, P, P, L, P, I, I, I, I, I, I, P],
    "i2r_conv2d_packed_h2_bytes": [I, I, I],
    "i2r_conv2d_pack_weight_h2": [P, P, I, I, I, P],
    "i2r_conv2d_splitk_h2_floats": [I, I, I, I, I, I],
    "i2r_conv2d_fwd_h2": [P, P, P, P, P, P, L, I, I, I, I, I, I, I, P],
    "i2r_conv2d_pack_weight_h2_dgrad": [P, P, I, I, I, P],
    "i2r_conv2d_pack_h2_block_pairs": [],
    "i2r_conv2d_pack_h2_multi": [P, P, I, P],
    "i2r_conv2d_dgrad_h2": [P, P, P, P, P, L, I, I, I, I, I, I, P],
    "i2r_absmax_pow2_scale_f32": [P, L, P, P, P],
    "i2r_conv2d_fwd_h2_noalloc_max_floats": [],
    "i2r_conv2d_fwd_h2_noalloc": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "i2r_conv1x1_cat_fwd_h2": [P, P, I, I, P, P, P, P, P, L, I, I, I, I, P],
    "i2r_conv2d_wgrad_f32": [P, P, P, P, I, I, I, I, I, I, I, P],
    "i2r_conv2d_wgrad_det_floats": [I, I, I, I, I, I],
    "i2r_conv2d_wgrad_det_f32": [P, P, P, P, P, L, I, I, I, I, I, I, I, P],
    "i2r_down2_f32": [P, P, L, I, I, F, P],
    "i2r_up2_f32": [P, P, L, I, I, F, P],
    "i2r_concat_channels_f32": [P, P, P, I, L, L, P],
    "i2r_split_channels_f32": [P, P, P, I, L, L, I, P],
    "i2r_groupnorm_fwd_f32": [P, P, P, P, I, P, P, P, I, I, I, I, F, I, F, U, U, P, I, P],
    "i2r_groupnorm_cat_fwd_f32": [P, P, I, I, P, P, P, I, P, I, I, I, F, I, I, P],
    "i2r_groupnorm_bwd_f32": [P, P, P, P, I, P, P, P, P, P, P, P, I, I, I, I, I, F, U, U, P, P],
    "i2r_groupnorm_bwd_acc_f32": [P, P, P, P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, F, U, U, P, P],
    "i2r_reduce_rows_f32": [P, P, I, I, P],
    "i2r_reduce_rows2_f32": [P, P, P, I, I, I, P],
    "i2r_attention_fwd_f32": [P, P, P, I, I, I, P],
    "i2r_attention_fwd_h2": [P, P, I, I, I, I, P],
    "i2r_attention_fwd_h2_lse": [P, P, P, I, I, I, P],
    "i2r_attention_bwd_f32": [P, P, P, P, P, P, I, I, I, P],
    "i2r_attention_bwd_h2": [P, P, P, P, P, P, P, I, I, I, P],
    "i2r_pos_embedding_f32": [P, P, I, I, I, P],
    "i2r_linear_fwd_f32": [P, P, P, P, P, I, I, I, I, P],
    "i2r_linear_bwd_f32": [P, P, P, P, P, P, I, I, I, P],
    "i2r_gemm_grouped_f32": [P, P, I, I, P, P, I, I, P],
    "i2r_silu_bwd_f32": [P, P, P, L, P],
    "i2r_add_f32": [P, P, P, L, P],
    "i2r_edm_scale_input": [P, I, P, I, P, P, I, L, F, P],
    "i2r_edm_combine": [P, I, P, P, I, P, I, L, F, P],
    "i2r_edm_combine_bwd": [P, P, I, P, I, L, F, P],
    "i2r_edm_noise": [P, P, P, P, P, P, P, I, L, F, F, F, U, U, P, P],
    "i2r_counter_add_u64": [P, U, P],
    "i2r_nan_mask_f32": [P, P, P, L, P],
    "i2r_masked_edm_loss": [P, P, P, P, P, P, P, I, I, L, P],
    "i2r_opt_chunk_elems": [],
    "i2r_fused_adamw_ema_step": [P, P, I, P, P, P, P, F, F, F, F, F, F, F, I, P],
    "i2r_heun_init_f64": [P, P, L, D, P],
    "i2r_heun_init_f64_from_f64": [P, P, L, D, P],
    "i2r_heun_euler_f64": [P, P, P, P, L, D, D, P],
    "i2r_heun_correct_f64": [P, P, P, P, P, L, D, D, P],
}

RETURNS_LONG = {"i2r_conv2d_packed_floats", "i2r_conv2d_splitk_floats", "i2r_conv2d_packed_h2_bytes",
                "i2r_conv2d_splitk_h2_floats", "i2r_conv2d_fwd_h2_noalloc_max_floats",
                "i2r_conv2d_wgrad_det_floats"}

_lib = None


class I2RError(RuntimeError):
    pass


def load():
    """Load (once) and return the CDLL.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise I2RError(
            f"imageni2r_amd: {LIB_PATH} not found -- the HIP extension is required (there is no CPU/ATen fallback). "
            "Build it with `make -C imageni2r_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library is stale
        fn.argtypes = argtypes
        fn.restype = c_long if name in RETURNS_LONG else c_int
    lib.i2r_last_error.argtypes = []
    lib.i2r_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


def query(name: str, *args) -> int:
    """Call a size-query entry point (returns a count, never an error code)."""
    return int(getattr(load(), name)(*args))


def call(name: str, *args):
    """Call an entry point; raise I2RError(i2r_last_error()) on a non-zero return."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise I2RError(f"{name} failed ({rc}): {lib.i2r_last_error().decode(errors='replace')}")
