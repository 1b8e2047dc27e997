"""Host-side mirror of the reference's EDM network classes (models/networks.py), running on the HIP kernels.

Same class names, constructor signatures, attribute names, parameter shapes and state-dict keys as the reference
(Linear :30, Conv2d :49, GroupNorm :96, UNetBlock :134, PositionalEmbedding :193, DhariwalUNet :372, EDMPrecond :632),
and parameters are drawn from torch's CPU generator in the same order with the same expressions, so
``torch.manual_seed(s)`` gives bit-identical initial weights (checked against tests/golden/g2_init.json).
Only the classes the ImagenI2R path instantiates are provided (models/our.py:26 -> EDMPrecond/DhariwalUNet).

The forward methods only sequence kernels: every arithmetic step is an entry point of libi2r_hip.so (see ops.py).
"""
from __future__ import annotations

import itertools
import os

import numpy as np
import torch

from . import ops

_dropout_counter = itertools.count(1)
_layer_ids = itertools.count(1)


# init mode -> (draw, gain): `draw` consumes torch's global CPU generator exactly like the reference's initialisers
# (networks.py:19-24: uniform modes draw rand()*2-1, normal modes randn()), so a seeded construction yields the
# reference's bits; `gain` is the mode's scale from the fan sizes.
_INIT_MODES = {
    'xavier_uniform': (lambda shape: torch.rand(*shape) * 2 - 1, lambda fi, fo: np.sqrt(6 / (fi + fo))),
    'xavier_normal': (lambda shape: torch.randn(*shape), lambda fi, fo: np.sqrt(2 / (fi + fo))),
    'kaiming_uniform': (lambda shape: torch.rand(*shape) * 2 - 1, lambda fi, fo: np.sqrt(3 / fi)),
    'kaiming_normal': (lambda shape: torch.randn(*shape), lambda fi, fo: np.sqrt(1 / fi)),
}


def weight_init(shape, mode, fan_in, fan_out):
    if mode not in _INIT_MODES:
        raise ValueError(f'Invalid init mode "{mode}"')
    draw, gain = _INIT_MODES[mode]
    return gain(fan_in, fan_out) * draw(shape)


class Linear(torch.nn.Module):
    def __init__(self, in_features, out_features, bias=True, init_mode='kaiming_normal', init_weight=1, init_bias=0):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        kw = dict(mode=init_mode, fan_in=in_features, fan_out=out_features)
        self.weight = torch.nn.Parameter(weight_init([out_features, in_features], **kw) * init_weight)
        self.bias = torch.nn.Parameter(weight_init([out_features], **kw) * init_bias) if bias else None

    def forward(self, x, act_silu=False):
        """x @ W^T + b (:39-43); ``act_silu`` fuses the silu the UNet applies right after (:440, :447)."""
        return ops.LinearFn.apply(x, self.weight, self.bias, act_silu)


class Conv2d(torch.nn.Module):
    def __init__(self, in_channels, out_channels, kernel, bias=True, up=False, down=False, resample_filter=[1, 1],
                 fused_resample=False, init_mode='kaiming_normal', init_weight=1, init_bias=0):
        assert not (up and down)
        super().__init__()
        if fused_resample:
            raise NotImplementedError("fused_resample is never used on the ImagenI2R path (DhariwalUNet)")
        if list(resample_filter) != [1, 1]:
            raise NotImplementedError("only the [1,1] box resample filter of DhariwalUNet is implemented")
        if kernel not in (0, 1, 3):
            raise NotImplementedError(f"kernel size {kernel}: the path uses 3x3, 1x1 and weight-less resampling only")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.up = up
        self.down = down
        self.fused_resample = fused_resample
        kw = dict(mode=init_mode, fan_in=in_channels * kernel * kernel, fan_out=out_channels * kernel * kernel)
        self.weight = torch.nn.Parameter(
            weight_init([out_channels, in_channels, kernel, kernel], **kw) * init_weight) if kernel else None
        self.bias = torch.nn.Parameter(weight_init([out_channels], **kw) * init_bias) if kernel and bias else None
        f = torch.as_tensor(resample_filter, dtype=torch.float32)
        f = f.ger(f).unsqueeze(0).unsqueeze(1) / f.sum().square()
        self.register_buffer('resample_filter', f if up or down else None)  # kept for state-dict compatibility

    def packed_weight(self):
        """K-major packed copy of ``weight`` for the conv kernels, re-made whenever the parameter is written
        (``_version``), moved (``data_ptr``) or a new pack epoch starts (the sampler bumps it inside graph capture so
        that every replay re-packs from the live weights)."""
        w = self.weight
        key = (w.data_ptr(), w._version, ops.pack_epoch())
        if getattr(self, "_wp_key", None) != key:
            self._wp = ops.conv2d_pack(w.detach())
            self._wp_key = key
        return self._wp

    def packed_weight_h2(self):
        """Split-f16 packed copy of ``weight`` for the inference kernel (same invalidation rule as packed_weight)."""
        w = self.weight
        key = (w.data_ptr(), w._version, ops.pack_epoch())
        if getattr(self, "_wph_key", None) != key:
            self._wph = ops.conv2d_pack_h2(w.detach())
            self._wph_key = key
        return self._wph

    def packed_weight_h2_center(self):
        """Split-f16 packed copy of the CENTRE TAP of a 3x3 weight, as a 1x1 weight (same invalidation rule): on a 1x1
        image the other eight taps only ever meet the zero padding (:68-90 with padding 1)."""
        w = self.weight
        key = (w.data_ptr(), w._version, ops.pack_epoch())
        if getattr(self, "_wphc_key", None) != key:
            self._wphc = ops.conv2d_pack_h2(w.detach()[:, :, 1:2, 1:2].contiguous())
            self._wphc_key = key
        return self._wphc

    def packed_weight_h2_dgrad(self):
        """Split-f16 packed copy for the data gradient on the f16x2 kernel (same invalidation rule)."""
        w = self.weight
        key = (w.data_ptr(), w._version, ops.pack_epoch())
        if getattr(self, "_wpdh_key", None) != key:
            self._wpdh = ops.conv2d_pack_h2_dgrad(w.detach())
            self._wpdh_key = key
        return self._wpdh

    def packed_weight_dgrad(self):
        """Transposed + flipped packed copy for the data gradient, only when a current one exists (made by
        DhariwalUNet.pack_all_weights); otherwise None and backward packs on the fly."""
        w = self.weight
        if getattr(self, "_wpd_key", None) == (w.data_ptr(), w._version, ops.pack_epoch()):
            return self._wpd
        return None

    def forward(self, x, residual=None, x_split=False, out=None, qkv_out=False, fuse_up=False, up_res=False):
        """up/down resample, conv, + bias (:68-90); ``residual`` is added in the conv epilogue.  ``x_split``: x is the
        pre-split f16-pair output of GroupNorm.forward(split_out=True) (inference, f16x2 conv math only).
        ``out`` (inference f16x2 path only): destination tensor."""
        if x_split and ((self.up and not fuse_up) or self.down or torch.is_grad_enabled()
                        or not ops.h2_math()):
            raise ValueError("x_split is only valid for a plain inference conv in f16x2 mode")
        if fuse_up:  # inference f16x2, 3x3: the nearest x2 upsample (:79-81) happens in the conv's own addressing
            if not self.up or torch.is_grad_enabled() or not ops.h2_math():
                raise ValueError("fuse_up needs an up=True conv in f16x2 inference")
            return ops.conv2d_fwd_h2(ops._c(x), self.weight.shape, self.packed_weight_h2(), self.bias,
                                     None if residual is None else ops._c(residual), x_split, out, False, True, up_res)
        if isinstance(x, tuple):  # (x, skip): torch.cat([x, skip], 1) read in place -- inference 1x1 convs only
            return ops.conv1x1_cat_fwd_h2(ops._c(x[0]), ops._c(x[1]), self.weight.shape, self.packed_weight_h2(),
                                          self.bias, None if residual is None else ops._c(residual))
        if self.up:
            x = ops.Up2Fn.apply(x)
        if self.down:
            x = ops.Down2Fn.apply(x)
        if self.weight is not None:
            if not torch.is_grad_enabled() and ops.h2_math():
                if (self.weight.shape[2] == 3 and x.shape[2] == 1 and x.shape[3] == 1 and not up_res
                        and os.environ.get("I2R_CENTER_TAP") != "0"):
                    # a 3x3 conv on a 1x1 image is the 1x1 conv of its centre tap: a ninth of the multiply-adds
                    wshape = (self.weight.shape[0], self.weight.shape[1], 1, 1)
                    return ops.conv2d_fwd_h2(ops._c(x), wshape, self.packed_weight_h2_center(), self.bias,
                                             None if residual is None else ops._c(residual), x_split, out, qkv_out,
                                             False, False)
                return ops.conv2d_fwd_h2(ops._c(x), self.weight.shape, self.packed_weight_h2(), self.bias,
                                         None if residual is None else ops._c(residual), x_split, out, qkv_out,
                                         False, up_res)
            if torch.is_grad_enabled() and ops.train_fwd_h2(x.shape[0] * x.shape[2] * x.shape[3]):
                bwd_h2 = x.requires_grad and ops.train_bwd_h2(x.shape[0] * x.shape[2] * x.shape[3],
                                                              self.weight.numel())
                self._train_modes = (True, bool(bwd_h2))  # which fp32 packs this layer does NOT need (_pack_plan)
                return ops.Conv2dFn.apply(x, self.weight, self.bias, residual, None,
                                          None if bwd_h2 else self.packed_weight_dgrad(), self.packed_weight_h2(),
                                          self.packed_weight_h2_dgrad() if bwd_h2 else None)
            if torch.is_grad_enabled():
                self._train_modes = (False, False)
            return ops.Conv2dFn.apply(x, self.weight, self.bias, residual, self.packed_weight(),
                                      self.packed_weight_dgrad() if torch.is_grad_enabled() else None)
        if residual is not None:
            raise ValueError("residual needs a convolution to fuse into")
        return x


class GroupNorm(torch.nn.Module):
    def __init__(self, num_channels, num_groups=32, min_channels_per_group=4, eps=1e-5):
        super().__init__()
        self.num_groups = min(num_groups, num_channels // min_channels_per_group)
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(num_channels))
        self.bias = torch.nn.Parameter(torch.zeros(num_channels))
        self._rng_id = next(_layer_ids)  # fixed Philox stream id of this layer's dropout (device-step RNG mode)

    def forward(self, x, scale_shift=None, act_silu=False, drop_p=0.0, split_out=False, bypass=False):
        """group_norm (:104-106), optionally fused with (1+scale)*y+shift, SiLU and dropout of its consumer.
        ``split_out`` (inference only): emit f16 hi/lo pairs for a following f16x2 conv instead of fp32."""
        if isinstance(x, tuple):  # (x, skip): the decoder's channel concat, never materialised (inference only)
            return ops.groupnorm_cat_fwd(ops._c(x[0]), ops._c(x[1]), self.weight, self.bias, self.num_groups, self.eps,
                                         None if scale_shift is None else ops._c(scale_shift), act_silu, split_out)
        if split_out:
            y, _, _ = ops.groupnorm_fwd(ops._c(x), self.weight, self.bias, self.num_groups, self.eps,
                                        None if scale_shift is None else ops._c(scale_shift), act_silu, split_out=True)
            return y
        seed = offset = 0
        if drop_p > 0.0:
            seed = ops.rng_seed()
            # host counter per call, or (device step counter, fixed layer id) when a graph owns the RNG step
            offset = self._rng_id if ops.rng_step() is not None else next(_dropout_counter)
        return ops.GroupNormFn.apply(x, self.weight, self.bias, scale_shift, self.num_groups, self.eps, act_silu,
                                     drop_p, seed, offset, bypass)


class UNetBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, emb_channels, up=False, down=False, attention=False,
                 num_heads=None, channels_per_head=64, dropout=0, skip_scale=1, eps=1e-5, resample_filter=[1, 1],
                 resample_proj=False, adaptive_scale=True, init=dict(), init_zero=dict(init_weight=0),
                 init_attn=None):
        super().__init__()
        if not adaptive_scale or skip_scale != 1:
            raise NotImplementedError("only the DhariwalUNet configuration (adaptive_scale, skip_scale=1) is built")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.emb_channels = emb_channels
        self.num_heads = 0 if not attention else num_heads if num_heads is not None else out_channels // channels_per_head
        self.dropout = dropout
        self.skip_scale = skip_scale
        self.adaptive_scale = adaptive_scale
        if self.num_heads and out_channels != self.num_heads * 64:
            raise NotImplementedError("attention head dim is fixed at 64 (channels_per_head=64, networks.py:393)")

        # Sub-modules in the reference's registration order (it is also the order their parameters are drawn in):
        # norm0, conv0, affine, norm1, conv1, [skip], [norm2, qkv, proj] (networks.py:148-164).
        resample = dict(up=up, down=down, resample_filter=resample_filter)
        cin, cout = in_channels, out_channels
        needs_skip = cout != cin or up or down
        skip_kernel = 1 if (resample_proj or cout != cin) else 0  # 0: weight-less resampling only
        layers = [
            ("norm0", lambda: GroupNorm(num_channels=cin, eps=eps)),
            ("conv0", lambda: Conv2d(in_channels=cin, out_channels=cout, kernel=3, **resample, **init)),
            ("affine", lambda: Linear(in_features=emb_channels, out_features=cout * 2, **init)),
            ("norm1", lambda: GroupNorm(num_channels=cout, eps=eps)),
            ("conv1", lambda: Conv2d(in_channels=cout, out_channels=cout, kernel=3, **init_zero)),
            ("skip", (lambda: Conv2d(in_channels=cin, out_channels=cout, kernel=skip_kernel, **resample, **init))
             if needs_skip else None),
        ]
        if self.num_heads:
            layers += [
                ("norm2", lambda: GroupNorm(num_channels=cout, eps=eps)),
                ("qkv", lambda: Conv2d(in_channels=cout, out_channels=cout * 3, kernel=1,
                                       **(init_attn if init_attn is not None else init))),
                ("proj", lambda: Conv2d(in_channels=cout, out_channels=cout, kernel=1, **init_zero)),
            ]
        for name, make in layers:
            setattr(self, name, make() if make is not None else None)

    def forward(self, x, emb, params=None):
        """networks.py:166-187 as 6 (+4 with attention) kernels:
        GN+SiLU -> conv0 -> affine -> GN+mod+SiLU(+dropout) -> [skip conv] -> conv1(+residual)
        [-> GN -> qkv conv -> attention -> proj conv(+residual)].
        ``params``: this block's ``affine(emb)`` when the network computed all of them in one grouped launch."""
        orig = x
        # inference in f16x2 conv math: the norms hand the convs pre-split f16 pairs (no conversion in the conv)
        sp = (not torch.is_grad_enabled()) and ops.h2_math()
        # up blocks in that mode: the x2 upsample of conv0's input and of the weight-less skip never hit memory
        fu = (sp and self.conv0.up and self.skip is not None and self.skip.weight is None and not isinstance(x, tuple)
              and x.shape[3] * 2 <= 32 and (x.shape[3] & (x.shape[3] - 1)) == 0
              and not ops.knob("no_fuse_up"))
        sp0 = sp and (fu or not (self.conv0.up or self.conv0.down))
        # training: the residual / skip branch leaves through norm0's bypass output, so x has one consumer and its two
        # gradients are summed inside the GroupNorm backward kernel instead of by an autograd accumulation kernel
        byp = torch.is_grad_enabled() and not isinstance(x, tuple) and not ops.knob("no_gn_bypass")
        if byp:
            h0, orig = self.norm0(x, act_silu=True, bypass=True)
        else:
            h0 = self.norm0(x, act_silu=True, split_out=sp0)
        x = self.conv0(h0, x_split=sp0, fuse_up=fu)
        if params is None:
            params = self.affine(emb)  # (B or 1, 2*Cout) = [scale | shift] (:170-172)
        x = self.norm1(x, scale_shift=params, act_silu=True, drop_p=self.dropout if self.training else 0.0,
                       split_out=sp)
        skip = orig if (fu or self.skip is None) else self.skip(orig)
        x = self.conv1(x, residual=skip, x_split=sp, up_res=fu)
        if self.num_heads:
            chunk = self._attn_chunk(x) if sp else 0
            if chunk:
                # inference at large batch: the attention tail runs over sample chunks whose qkv / attention tensors
                # (3C and C channels) stay in the 256 MB infinity cache between the four kernels
                out = torch.empty_like(x)
                for i in range(0, x.shape[0], chunk):
                    xc = x[i:i + chunk]
                    self.proj(self._attend(xc, True), residual=xc, out=out[i:i + chunk])
                return out
            if byp:
                xn, xb = self.norm2(x, bypass=True)
                x = self.proj(ops.AttentionFn.apply(self.qkv(xn), self.num_heads), residual=xb)
            else:
                x = self.proj(self._attend(x, sp), residual=x)
        return x

    def _attend(self, x, sp):
        """norm2 -> qkv conv -> attention.  In f16x2 inference the qkv conv writes split pairs (q pre-scaled) and the
        attention kernel only permutes them into fragments."""
        if sp:
            pre = ops.attention_h2_ok(x.shape[2] * x.shape[3]) and not ops.knob("no_qkv_presplit")
            qkv = self.qkv(self.norm2(x, split_out=True), x_split=True, qkv_out=pre)
            return ops.attention_fwd(qkv, self.num_heads, presplit=pre)[0]
        return ops.AttentionFn.apply(self.qkv(self.norm2(x)), self.num_heads)

    def _attn_chunk(self, x):
        """Samples per chunk of the chunked attention tail (knob "attn_chunk_bytes" of qkv per chunk; 0 disables)."""
        per_sample = 3 * x.shape[1] * x.shape[2] * x.shape[3] * 4
        budget = ops.knob("attn_chunk_bytes")
        if not budget:
            return 0
        chunk = 1 << max(0, int(budget // per_sample).bit_length() - 1)
        return chunk if x.shape[0] >= 2 * chunk else 0


class PositionalEmbedding(torch.nn.Module):
    def __init__(self, num_channels, max_positions=10000, endpoint=False):
        super().__init__()
        if endpoint:
            raise NotImplementedError("endpoint=True is not used by DhariwalUNet")
        self.num_channels = num_channels
        self.max_positions = max_positions
        self.endpoint = endpoint

    def forward(self, x):
        return ops.pos_embedding(x.to(torch.float32).reshape(-1).contiguous(), self.num_channels, self.max_positions)


def adm_unet_plan(img_resolution, in_channels, model_channels, channel_mult, num_blocks, attn_resolutions):
    """Topology of the ADM UNet the reference builds (networks.py:402-433) as two ordered lists of
    ``(name, in_channels, out_channels, kind, attention)`` with kind in {"conv", "block", "down", "up"}, plus the channel
    count that reaches the output norm.  Names carry the NOMINAL resolution ``img_resolution >> level`` (the state-dict
    contract).  Encoder: a 3x3 stem, then per level an optional down block and ``num_blocks`` blocks;
    decoder, deepest level first: two bottleneck blocks (the first with attention) or an up block, then
    ``num_blocks + 1`` blocks, each consuming one encoder output as a skip connection (its channels are added to the
    block's input)."""
    levels = [(img_resolution >> lv, model_channels * mult) for lv, mult in enumerate(channel_mult)]
    enc, ch = [], in_channels
    for lv, (res, width) in enumerate(levels):
        tag = f"{res}x{res}"
        if lv == 0:
            enc.append((f"{tag}_conv", ch, width, "conv", False))
            ch = width
        else:
            enc.append((f"{tag}_down", ch, ch, "down", False))
        for i in range(num_blocks):
            enc.append((f"{tag}_block{i}", ch, width, "block", res in attn_resolutions))
            ch = width
    skip_widths = [e[2] for e in enc]
    dec = []
    for lv, (res, width) in reversed(list(enumerate(levels))):
        tag = f"{res}x{res}"
        if lv == len(levels) - 1:
            dec.append((f"{tag}_in0", ch, ch, "block", True))
            dec.append((f"{tag}_in1", ch, ch, "block", False))
        else:
            dec.append((f"{tag}_up", ch, ch, "up", False))
        for i in range(num_blocks + 1):
            dec.append((f"{tag}_block{i}", ch + skip_widths.pop(), width, "block", res in attn_resolutions))
            ch = width
    return enc, dec, ch


class DhariwalUNet(torch.nn.Module):
    def __init__(self, img_resolution, in_channels, out_channels, label_dim=0, augment_dim=0, model_channels=192,
                 channel_mult=[1, 2, 3, 4], channel_mult_emb=4, num_blocks=3, attn_resolutions=[32, 16, 8],
                 dropout=0.10, label_dropout=0):
        super().__init__()
        if label_dim or augment_dim:
            raise NotImplementedError("class / augment conditioning is not used by ImagenI2R (our.py:26-27)")
        self.label_dropout = label_dropout
        emb_channels = model_channels * channel_mult_emb
        init = dict(init_mode='kaiming_uniform', init_weight=np.sqrt(1 / 3), init_bias=np.sqrt(1 / 3))
        init_zero = dict(init_mode='kaiming_uniform', init_weight=0, init_bias=0)
        block_kwargs = dict(emb_channels=emb_channels, channels_per_head=64, dropout=dropout, init=init,
                            init_zero=init_zero)

        self.map_noise = PositionalEmbedding(num_channels=model_channels)
        self.map_augment = None
        self.map_layer0 = Linear(in_features=model_channels, out_features=emb_channels, **init)
        self.map_layer1 = Linear(in_features=emb_channels, out_features=emb_channels, **init)
        self.map_label = None

        # The topology comes from a plan (adm_unet_plan): entries are instantiated in plan order, which is the
        # reference's registration order and therefore its parameter-draw order and its state-dict key order.
        enc_plan, dec_plan, final_channels = adm_unet_plan(img_resolution, in_channels, model_channels, channel_mult,
                                                           num_blocks, attn_resolutions)
        self.enc = torch.nn.ModuleDict()
        self.dec = torch.nn.ModuleDict()
        for holder, plan in ((self.enc, enc_plan), (self.dec, dec_plan)):
            for name, cin, cout, kind, attention in plan:
                if kind == "conv":
                    holder[name] = Conv2d(in_channels=cin, out_channels=cout, kernel=3, **init)
                else:
                    holder[name] = UNetBlock(in_channels=cin, out_channels=cout, up=(kind == "up"),
                                             down=(kind == "down"), attention=attention, **block_kwargs)
        self.out_norm = GroupNorm(num_channels=final_channels)
        self.out_conv = Conv2d(in_channels=final_channels, out_channels=out_channels, kernel=3, **init_zero)
        # dropout stream ids depend only on the position inside this network (same model => same masks per step)
        for i, mod in enumerate(m for m in self.modules() if isinstance(m, GroupNorm)):
            mod._rng_id = i + 1

    def _pack_plan(self, dgrad=True):
        """Host side of pack_all_weights: the job list, destination views, pointer table and block map (cached while
        the weights stay where they are).  None when there is nothing to pack.  Launches nothing."""
        convs = [m for m in self.modules() if isinstance(m, Conv2d) and m.weight is not None]
        fwd_too = not ops.train_fwd_h2()  # every forward on the f16x2 kernel: only the data-gradient layout is fp32
        if not (fwd_too or dgrad):
            return None
        # A layer whose last training forward ran on the f16x2 kernel needs no fp32 forward pack, and none of the
        # data-gradient layout either if its data gradient runs there too (Conv2d.forward records both).  Unknown
        # (first step) = pack both; a layer that changes its mind packs on the fly (packed_weight's key check).
        if ops.train_fwd_h2_enabled() and not os.environ.get("I2R_PACK_ALL"):
            modes = tuple(getattr(c, "_train_modes", (False, False)) for c in convs)
        else:
            modes = tuple((False, False) for _ in convs)
        key = (tuple(c.weight.data_ptr() for c in convs), bool(dgrad), fwd_too, modes)
        # every plan made stays alive: captured graphs hold raw pointers into its pool (a short last batch has other
        # modes than the full ones, and its capture must not free the pool the full-batch graph replays from)
        sets = self.__dict__.setdefault("_packsets", {})
        ps = sets.get(key)
        if ps is None:
            dev = convs[0].weight.device
            blk = ops._lib.query("i2r_conv2d_pack_block_floats")
            jobs = []
            for c, (f_h2, b_h2) in zip(convs, modes):
                Cout, Cin, k, _ = c.weight.shape
                for dg in ([0] if fwd_too and not f_h2 else []) + ([1] if dgrad and not b_h2 else []):
                    jobs.append((c, dg, ops._lib.query("i2r_conv2d_packed_floats", Cout, Cin, k, dg)))
            ps = [key, jobs, [], None, None, 0, None, None]
            if jobs:
                pool = torch.empty(sum((n + 3) // 4 * 4 for _, _, n in jobs), dtype=torch.float32, device=dev)
                rows, bmap, views, off = [], [], [], 0
                for t, (c, dg, n) in enumerate(jobs):
                    v = pool[off: off + n]
                    off += (n + 3) // 4 * 4
                    Cout, Cin, k, _ = c.weight.shape
                    rows.append([c.weight.data_ptr(), v.data_ptr(), Cout, Cin, k, dg])
                    bmap.extend([t, b] for b in range((n + blk - 1) // blk))
                    views.append(v)
                ps[2:7] = [views, torch.tensor(rows, dtype=torch.int64).to(dev),
                           torch.tensor(bmap, dtype=torch.int32).to(dev), len(bmap), pool]
            # the split-f16 layouts of the layers that run on the f16x2 kernels, likewise in one launch
            hjobs = []
            for c, (f_h2, b_h2) in zip(convs, modes):
                Cout, Cin, k, _ = c.weight.shape
                if f_h2:
                    hjobs.append((c, 0, ops._lib.query("i2r_conv2d_packed_h2_bytes", Cout, Cin, k)))
                if b_h2 and dgrad:
                    hjobs.append((c, 1, ops._lib.query("i2r_conv2d_packed_h2_bytes", Cin, Cout, k)))
            if hjobs and not os.environ.get("I2R_PACK_H2_PER_LAYER"):
                hblk = ops._lib.query("i2r_conv2d_pack_h2_block_pairs")
                hpool = torch.empty(sum((n + 255) // 256 * 256 for _, _, n in hjobs), dtype=torch.uint8, device=dev)
                rows, bmap, hviews, off = [], [], [], 0
                for t, (c, dg, n) in enumerate(hjobs):
                    v = hpool[off: off + n]
                    off += (n + 255) // 256 * 256
                    Cout, Cin, k, _ = c.weight.shape
                    rows.append([c.weight.data_ptr(), v.data_ptr(), Cout, Cin, k, dg])
                    bmap.extend([t, b] for b in range((n // 4 + hblk - 1) // hblk))
                    hviews.append(v)
                ps[7] = (hjobs, hviews, torch.tensor(rows, dtype=torch.int64).to(dev),
                         torch.tensor(bmap, dtype=torch.int32).to(dev), len(bmap), hpool)
            ps = tuple(ps)
            sets[key] = ps
        return ps

    def pack_all_weights(self, dgrad=True):
        """Pack every conv weight some kernel of the coming step reads -- fp32 forward / data-gradient layouts in one
        launch, split-f16 layouts in another -- into persistent buffers and mark them current, so the following
        forward/backward launches no per-layer pack kernels.  Call once per optimiser step (GraphedTrainStep does,
        inside its graph)."""
        ps = self._pack_plan(dgrad)
        if ps is None:
            return
        _, jobs, views, table, bmap_t, nblocks, _, h2 = ps
        if jobs:
            ops.pack_multi(table, bmap_t, nblocks)
            for (c, dg, _), v in zip(jobs, views):
                k3 = (c.weight.data_ptr(), c.weight._version, ops.pack_epoch())
                if dg:
                    c._wpd, c._wpd_key = v, k3
                else:
                    c._wp, c._wp_key = v, k3
        if h2 is not None:
            hjobs, hviews, htable, hbmap, hblocks, _ = h2
            ops.pack_h2_multi(htable, hbmap, hblocks)
            for (c, dg, _), v in zip(hjobs, hviews):
                k3 = (c.weight.data_ptr(), c.weight._version, ops.pack_epoch())
                if dg:
                    c._wpdh, c._wpdh_key = v, k3
                else:
                    c._wph, c._wph_key = v, k3

    @staticmethod
    def _cat_in_place(x, skip, block):
        """Inference in f16x2 mode, geometry the dual-source kernels cover (H*W a power of two >= 4, C1 % 16 == 0,
        a 1x1 skip conv): the decoder concat (:458) is not materialised."""
        hw = x.shape[2] * x.shape[3]
        return ((not torch.is_grad_enabled()) and ops.h2_math() and hw >= 4 and (hw & (hw - 1)) == 0
                and x.shape[1] % 16 == 0 and block.skip is not None and block.skip.weight is not None
                and block.skip.weight.shape[-1] == 1 and not (block.conv0.up or block.conv0.down)
                and (x.shape[1] + skip.shape[1]) // block.norm0.num_groups * hw <= 16384)

    def embed(self, noise_labels):
        """Noise-level embedding (:437-447): posemb -> Linear+SiLU -> Linear+SiLU."""
        emb = self.map_noise(noise_labels)
        emb = self.map_layer0(emb, act_silu=True)
        return self.map_layer1(emb, act_silu=True)

    def affine_all(self, emb):
        """``block.affine(emb)`` of every UNetBlock in one grouped launch (ops.AffineAllFn) -> {block: [scale|shift]}.
        Used in inference, and in training when the affine parameters have gradient slots (FusedAdamW /
        GradBucketReducer); otherwise None and each block runs its own Linear."""
        blocks = [b for b in list(self.enc.values()) + list(self.dec.values()) if isinstance(b, UNetBlock)]
        lins = [b.affine for b in blocks]
        need_grad = torch.is_grad_enabled() and (emb.requires_grad or lins[0].weight.requires_grad)
        if need_grad and any(ops.param_slot(l.weight) is None or ops.param_slot(l.bias) is None for l in lins):
            return None  # no slots, or gradients are being accumulated over several backwards
        key = ops.AffinePlan.make_key(lins, emb.shape[0])
        plans = self.__dict__.setdefault("_affine_plans", {})  # per (batch rows, parameter / slot addresses)
        plan = plans.get(key)
        if plan is None:
            # plans are a few KB of launch tables; captured graphs hold raw pointers into them, so none is ever
            # dropped while the network lives (keys: batch rows x parameter/slot addresses -- a handful)
            # (a long-lived serving process that has met 256 distinct batch sizes: later ones take the per-block Linear
            # path -- slower, never fatal; no cached plan may be dropped, graphs replay against them)
            if len(plans) >= 256:
                return None
            plan = plans[key] = ops.AffinePlan(lins, emb.shape[0])
        flat = [t for l in lins for t in (l.weight, l.bias)]
        outs = ops.AffineAllFn.apply(emb, plan, *flat)
        if need_grad and outs[0].grad_fn is not None:
            for o, slot in zip(outs, ops.AffineAllFn.dss_slots(outs[0].grad_fn.dys, plan)):
                o._dss_slot = slot  # the consuming GroupNorm's backward writes d[scale|shift] here
        return dict(zip(blocks, outs))

    def forward(self, x, noise_labels, class_labels=None, augment_labels=None):
        emb = self.embed(noise_labels)
        ss = self.affine_all(emb)
        skips = []
        for block in self.enc.values():
            x = block(x, emb, None if ss is None else ss[block]) if isinstance(block, UNetBlock) else block(x)
            skips.append(x)
        for block in self.dec.values():
            if x.shape[1] != block.in_channels:
                skip = skips.pop()
                if self._cat_in_place(x, skip, block):
                    x = (x, skip)  # norm0 and the 1x1 skip conv read the two tensors in place
                else:
                    x = ops.ConcatFn.apply(x, skip)
            x = block(x, emb, None if ss is None else ss[block])
        sp = (not torch.is_grad_enabled()) and ops.h2_math()
        return self.out_conv(self.out_norm(x, act_silu=True, split_out=sp), x_split=sp)


class EDMPrecond(torch.nn.Module):
    def __init__(self, img_resolution, img_channels, label_dim=0, use_fp16=False, sigma_min=0,
                 sigma_max=float('inf'), sigma_data=0.5, model_type='DhariwalUNet', **model_kwargs):
        super().__init__()
        if model_type != 'DhariwalUNet':
            raise NotImplementedError("ImagenI2R only instantiates DhariwalUNet (our.py:26-27)")
        if use_fp16:
            raise NotImplementedError("use_fp16 is never set by ImagenI2R; the path is fp32")
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.label_dim = label_dim
        self.use_fp16 = use_fp16
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.sigma_data = sigma_data
        self.model = DhariwalUNet(img_resolution=img_resolution, in_channels=img_channels, out_channels=img_channels,
                                  label_dim=label_dim, **model_kwargs)

    def forward(self, x, sigma, class_labels=None, force_fp32=False, **model_kwargs):
        """D = c_skip x + c_out F(c_in x, ln(sigma)/4) (:654-668).  x may be fp32 or fp64 (the sampler's state)."""
        if not x.is_cuda:
            raise ops._lib.I2RError("EDMPrecond.forward: input is on the CPU; the HIP path has no CPU fallback")
        sigma = torch.as_tensor(sigma)
        sig = ops._sigma_vec(sigma, x.device)
        if sig.numel() not in (1, x.shape[0]):
            raise ValueError("sigma must be a scalar or have one entry per sample")
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        x = x if x.is_contiguous() else x.contiguous()
        xin, c_noise = ops.edm_scale_input(x, sig, self.sigma_data)
        F_x = self.model(xin, c_noise, class_labels=None, **model_kwargs)
        return ops.EdmCombineFn.apply(x, F_x, sig, self.sigma_data)

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)
