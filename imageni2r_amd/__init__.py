"""imageni2r_amd -- MI355X (gfx950) native masked-diffusion hot path of ImagenI2R.

Drop-in classes (same names / signatures / state-dict keys as the reference's models/*.py):
    TS2img_Karras, EDMPrecond, DhariwalUNet, UNetBlock, Conv2d, GroupNorm, Linear, PositionalEmbedding,
    DelayEmbedder, DiffusionProcess, LitEma
MI355X-side additions (no counterpart in the reference): FusedAdamW (clip + AdamW + EMA in one multi-tensor kernel
step), GraphedTrainStep (the whole train step as one replayed HIP graph), set_deterministic (reproducible gradient
summation order), tst_boundary.TrainIteration (the reference's loop body on the TST / diffusion step boundary) and
metrics / context_fid (the evaluation stage's scores, kept on the device).
All arithmetic runs in libi2r_hip.so (imageni2r_amd/csrc, C ABI in include/i2r_hip.h); importing this package does not
need a GPU, calling an op does (there is no CPU fallback).
"""
from .networks import (Conv2d, DhariwalUNet, EDMPrecond, GroupNorm, Linear, PositionalEmbedding, UNetBlock)
from .delay import DelayEmbedder
from .ema import LitEma
from .karras import TS2img_Karras
from .sampler import DiffusionProcess
from .ops import conv_math, set_conv_math, set_deterministic
from .optim import FusedAdamW
from .train import GraphedTrainStep

__all__ = ["TS2img_Karras", "EDMPrecond", "DhariwalUNet", "UNetBlock", "Conv2d", "GroupNorm", "Linear",
           "PositionalEmbedding", "DelayEmbedder", "DiffusionProcess", "LitEma", "conv_math", "set_conv_math", "FusedAdamW",
           "GraphedTrainStep", "set_deterministic"]
