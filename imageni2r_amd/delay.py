"""DelayEmbedder on the HIP gather kernels -- host-side mirror of models/img_transformations.py:39-170.

Same constructor, attributes (``img_shape``, ``mapping``, ``seq_len``) and methods as the reference.  The image is
produced on the device in one coalesced pass (closed form t = col*delay + row) instead of being assembled on the CPU
column by column; ``img_to_ts`` is one kernel instead of a Python loop over the sequence.
"""
from __future__ import annotations

import torch

from . import ops


def delay_geometry(seq_len: int, delay: int, embedding: int):
    """(cols, side) of the delay image: the loop bounds of img_transformations.py:87-99."""
    n_full = 0 if seq_len < embedding else (seq_len - embedding) // delay + 1
    cols = n_full + (1 if n_full * delay != seq_len else 0)
    return cols, max(embedding, cols)


class DelayEmbedder:
    def __init__(self, device, seq_len, delay, embedding, batch_size, num_features):
        self.device = device
        self.seq_len = seq_len
        self.delay = delay
        self.embedding = embedding
        self.batch_size = batch_size
        self.num_features = num_features
        self.img_shape = None
        self.mapping = None
        self.create_mapping()

    # -- geometry -----------------------------------------------------------------------------------------------
    def create_mapping(self):
        """ts index -> [(row, col), ...] in row-major scan order (:111-138), in closed form (no helper image).
        Also caches ``img_shape`` like the reference's helper pass does (:121 -> :102)."""
        cols, _ = delay_geometry(self.seq_len, self.delay, self.embedding)
        if cols > self.embedding:
            raise ValueError(f"delay embedding needs cols ({cols}) <= embedding ({self.embedding}): the reference "
                             "allocates an embedding x embedding image (img_transformations.py:86)")
        mapping = {}
        for row in range(self.embedding):
            for col in range(cols):
                t = col * self.delay + row
                if t < self.seq_len:
                    mapping.setdefault(t, []).append((row, col))
        self.mapping = mapping
        self.img_shape = (self.batch_size, self.num_features, self.embedding, cols)

    def pad_to_square(self, x, mask=0):
        """:59-70 (plumbing only: constant pad right/bottom to a square)."""
        _, _, cols, rows = x.shape
        side = max(cols, rows)
        return torch.nn.functional.pad(x, (0, side - rows, 0, side - cols), mode='constant', value=mask)

    def unpad(self, x, original_shape):
        _, _, original_cols, original_rows = original_shape
        return x[:, :, :original_cols, :original_rows]

    # -- transforms ---------------------------------------------------------------------------------------------
    def _prep(self, signal):
        if signal.dim() != 3:
            raise ValueError("signal must be (batch, length, features)")
        batch, length, features = signal.shape
        if self.seq_len != length:  # variable-length datasets (:83-84)
            self.seq_len = length
        cols, side = delay_geometry(length, self.delay, self.embedding)
        if cols > self.embedding:
            raise ValueError(f"cols ({cols}) > embedding ({self.embedding})")
        dev = torch.device(self.device)
        if dev.type != 'cuda':
            raise ops._lib.I2RError("DelayEmbedder: device must be a HIP ('cuda') device; there is no CPU fallback")
        sig = signal.to(device=dev, dtype=torch.float32)
        return sig, batch, features, cols, side

    def ts_to_img(self, signal, pad=True, mask=0):
        """(B, L, C) -> (B, C, S, S) fp32 on ``self.device`` (:79-108); NaNs are copied through."""
        sig, batch, features, cols, side = self._prep(signal)
        self.img_shape = (batch, features, self.embedding, cols)
        img = ops.TsToImgFn.apply(sig, self.delay, self.embedding, cols, side, float(mask))
        if not pad:
            img = img[:, :, :self.embedding, :cols]
        return img

    def ts_to_img_with_mask(self, signal, mask=0):
        """Fused variant: image and the observed mask ``1 - isnan(img)`` (run_irregular.py:216) in one pass."""
        sig, batch, features, cols, side = self._prep(signal)
        self.img_shape = (batch, features, self.embedding, cols)
        return ops.ts_to_img(sig.contiguous(), self.delay, self.embedding, cols, side, float(mask), want_mask=True)

    def img_to_ts(self, x_image_square):
        """(B, C, S, S) fp32|fp64 -> (B, L, C) fp32 (:140-170)."""
        if self.img_shape is None:
            raise RuntimeError("img_to_ts called before any ts_to_img (img_shape unknown)")
        _, _, rows, cols = self.img_shape
        x = x_image_square
        if x.dim() != 4 or x.shape[-1] != x.shape[-2]:
            raise ValueError("img_to_ts expects a square (B, C, S, S) image")
        if x.shape[-1] < rows or x.shape[-1] < cols:
            # the reference fails the same way when sampled at the nominal img_resolution
            raise IndexError(f"index {x.shape[-1]} is out of bounds: image side {x.shape[-1]} < delay image "
                             f"{rows}x{cols}")
        if not x.is_cuda:
            raise ops._lib.I2RError("img_to_ts: image is on the CPU; there is no CPU fallback")
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        return ops.img_to_ts(x.contiguous(), self.seq_len, self.delay, self.embedding, cols)
