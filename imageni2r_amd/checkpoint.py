"""Checkpoints in the reference's own ``.pth`` schema.

``save_checkpoint`` keeps the signature, directory layout, file name and dictionary keys of run_irregular.py:41-87, so a
file written here loads in the reference and the other way round (the state-dict key names of ``TS2img_Karras`` /
``LitEma`` / AdamW are the reference's).  ``restore_checkpoint`` is the loader utils/utils.py:26-30 calls but the
reference never defines.  Host-side code only: tensors move with ``state_dict`` / ``load_state_dict``.
"""
from __future__ import annotations

import glob
import os
import shutil

import torch

CHECKPOINT_KEYS = ("our_model_state_dict", "our_optimizer_state_dict", "ema_model", "tst_encoder", "tst_decoder",
                   "tst_optimizer", "disc_score", "pred_score", "fid_score", "correlation_score", "args")


def checkpoint_dir(args):
    """<model_save_path>/seq_len_<L>/<dataset>/missing_rate_<percent> (run_irregular.py:46-52)."""
    return os.path.join(args.model_save_path, f"seq_len_{args.seq_len}", args.dataset,
                        f"missing_rate_{int(args.missing_rate * 100)}")


def _state(obj):
    return obj.state_dict() if obj is not None else None


def _plain(score):
    """Scores arrive as numpy scalars (metrics/__init__.py:50 returns np.round(np.mean(..), 4)); a plain Python
    number pickles without numpy globals, so the file stays loadable with ``weights_only=True``.  str() of a
    numpy float64 and of the equal Python float are the same text, so the file name does not change."""
    if score is None or type(score) in (bool, int, float, str):  # (numpy.float64 IS a float subclass: exact types)
        return score
    try:
        return score.item() if hasattr(score, "item") else float(score)
    except Exception:  # noqa: BLE001 -- an unexpected type is stored as the caller gave it
        return score


def _numpy_scalar_globals():
    """The globals a pickled numpy scalar needs -- what reference-written checkpoints hold for the four scores.
    Allow-listing them lets ``weights_only=True`` read such files without executing anything from them."""
    import numpy as np
    allowed = [np.dtype, np.float64, np.float32, np.int64]
    try:
        allowed.append(np._core.multiarray.scalar)
    except AttributeError:  # numpy < 2
        allowed.append(np.core.multiarray.scalar)
    for name in ("Float64DType", "Float32DType", "Int64DType"):
        t = getattr(np.dtypes, name, None)
        if t is not None:
            allowed.append(t)
    return allowed


def save_checkpoint(args, our_model, our_optimizer, ema_model, encoder, decoder, tst_optimizer, disc_score,
                    pred_score=None, fid_score=None, correlation_score=None):
    """Write the single best checkpoint of a run, replacing whatever the directory held (:55-62).  Like the reference
    it reports a failure on stdout instead of raising (:86-87); the path written (or None) is returned on top."""
    try:
        full_path = checkpoint_dir(args)
        os.makedirs(full_path, exist_ok=True)
        for f in glob.glob(os.path.join(full_path, "*")):
            if os.path.isdir(f):
                shutil.rmtree(f)
            else:
                os.remove(f)
        filename = (f"disc_score_{disc_score}_pred_score_{pred_score}_fid_score_{fid_score}"
                    f"_correlation_score_{correlation_score}.pth")
        filepath = os.path.join(full_path, filename)
        torch.save({
            "our_model_state_dict": our_model.state_dict(),
            "our_optimizer_state_dict": our_optimizer.state_dict(),
            "ema_model": _state(ema_model),  # the reference passes None when args.ema is off (:283) and fails there
            "tst_encoder": _state(encoder),
            "tst_decoder": _state(decoder),
            "tst_optimizer": _state(tst_optimizer),
            "disc_score": _plain(disc_score),
            "pred_score": _plain(pred_score),
            "fid_score": _plain(fid_score),
            "correlation_score": _plain(correlation_score),
            "args": {k: v for k, v in vars(args).items()},
        }, filepath)
        print(f"Checkpoint saved at: {filepath}")
        return filepath
    except Exception as e:  # noqa: BLE001 -- the reference's contract: training goes on
        print(f"Failed to save checkpoint: {e}")
        return None


def find_checkpoint(path):
    """``path`` itself if it is a file, else the newest ``*.pth`` below it."""
    if os.path.isfile(path):
        return path
    found = sorted(glob.glob(os.path.join(path, "**", "*.pth"), recursive=True), key=os.path.getmtime)
    if not found:
        raise FileNotFoundError(f"no .pth checkpoint under {path}")
    return found[-1]


def restore_checkpoint(ckpt_dir, state, ema_model=None, map_location="cpu"):
    """Load a checkpoint into the live objects of ``state`` and return ``state``.

    ``state`` may hold ``model`` (TS2img_Karras), ``optimizer``, ``encoder``, ``decoder``, ``tst_optimizer``; each one
    present is loaded from its key of the file, and ``state['epoch']`` (read by utils/utils.py:29) is set from the
    file when it has one, else kept / 0.  The scores and the saved ``args`` are put under the same names as in the
    file.  ``ema_model``: a LitEma to load separately (the copy inside ``our_model_state_dict`` under ``model_ema.*``
    is loaded with the model anyway).  The file is read with ``weights_only=True``: tensors, plain Python values and
    numpy scalars (the reference stores its scores as numpy.float64) only; nothing in it is executed.
    """
    path = find_checkpoint(ckpt_dir)
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    for key, name in (("our_model_state_dict", "model"), ("our_optimizer_state_dict", "optimizer"),
                      ("tst_encoder", "encoder"), ("tst_decoder", "decoder"), ("tst_optimizer", "tst_optimizer")):
        target = state.get(name)
        if target is not None and ckpt.get(key) is not None:
            target.load_state_dict(ckpt[key])
    if ema_model is not None and ckpt.get("ema_model") is not None:
        ema_model.load_state_dict(ckpt["ema_model"])
    model = state.get("model")
    if model is not None:
        from . import ops
        ops.new_pack_epoch()  # load_state_dict copies in place: packed conv weights of the old values are stale
    for key in ("disc_score", "pred_score", "fid_score", "correlation_score", "args"):
        if key in ckpt:
            state[key] = ckpt[key]
    state["epoch"] = ckpt.get("epoch", state.get("epoch", 0))
    state["path"] = path
    return state
