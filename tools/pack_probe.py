#!/usr/bin/env python3
"""Stand-alone check of DhariwalUNet.pack_all_weights (the one-launch multi-tensor weight pack) in both job layouts:
forward + data-gradient rows (default) and data-gradient rows only (I2R_TRAIN_FWD_H2=1, the training forward then packs
its own f16x2 weights per layer).  Before the launch the pointer table is validated on the host (every destination range
inside the pool, sizes equal to the C-ABI's own size query, block map consistent); after it every packed view must equal
the single-tensor pack of the same weight and the guard words after each view (the pool rounds sizes up) must be intact.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageni2r_amd import TS2img_Karras, ops  # noqa: E402
from imageni2r_amd.networks import Conv2d  # noqa: E402
from tests.helpers import tiny_args  # noqa: E402

dev = "cuda"
for h2fwd in (False, True):
    ops.set_train_fwd_h2(h2fwd)
    for name, args in (("tiny", tiny_args(device=dev)),
                       ("A", tiny_args(device=dev, ch_mult=[1, 2, 2, 2], unet_channels=128, attn_resolution=[8, 4, 2],
                                       input_channels=5))):
        torch.manual_seed(0)
        m = TS2img_Karras(args, dev).to(dev)
        net = m.net.model
        with torch.no_grad():
            for p in net.parameters():
                if float(p.abs().max()) == 0:
                    p.normal_(0, 0.05)
        convs = [c for c in net.modules() if isinstance(c, Conv2d) and c.weight is not None]
        # build the plan without launching: pack_all_weights builds it lazily, so replicate its host checks first
        blk = ops._lib.query("i2r_conv2d_pack_block_floats")
        ops.new_pack_epoch()
        key, jobs, views, table, bmap, nblocks, pool, _h2 = net._pack_plan(dgrad=True)  # host only: nothing launched yet
        rows = table.cpu().tolist()
        lo, hi = pool.data_ptr(), pool.data_ptr() + pool.numel() * 4
        assert len(rows) == len(jobs) == len(views)
        want_dg = [1] if h2fwd else [0, 1]
        assert [r[5] for r in rows] == want_dg * len(convs), "job layout"
        for (c, dg, n), v, r in zip(jobs, views, rows):
            Cout, Cin, k, _ = c.weight.shape
            assert r[0] == c.weight.data_ptr() and r[1] == v.data_ptr() and r[2:5] == [Cout, Cin, k]
            assert n == ops._lib.query("i2r_conv2d_packed_floats", Cout, Cin, k, dg) == v.numel()
            assert lo <= r[1] and r[1] + 4 * n <= hi, "destination outside the pool"
        bm = bmap.cpu().tolist()
        assert len(bm) == nblocks
        for t, (c, dg, n) in enumerate(jobs):
            assert [b for tt, b in bm if tt == t] == list(range((n + blk - 1) // blk)), "block map"
        pool.fill_(float("nan"))
        net.pack_all_weights(dgrad=True)  # the table has passed the host checks above: launch
        torch.cuda.synchronize()
        worst = 0
        for (c, dg, n), v in zip(jobs, views):
            ref = ops.conv2d_pack(c.weight.detach(), dgrad=bool(dg))
            assert torch.equal(v, ref), (name, tuple(c.weight.shape), dg)
            worst = max(worst, n)
        # cached views are marked current exactly for the layouts that were packed
        for c in convs:
            assert c.packed_weight_dgrad() is not None
            assert (getattr(c, "_wp_key", None) == c._wpd_key) == (not h2fwd)
        print(f"h2fwd={h2fwd} {name}: {len(jobs)} jobs, {nblocks} blocks, largest {worst} floats -- table and packs ok",
              flush=True)
        # one training forward/backward through the chosen path against the exact path
        x = torch.randn(2, args.input_channels, 8, 8, device=dev)
        sig = torch.tensor([0.7, 2.0], device=dev)
        m.train(False)
        outs = []
        for on in (h2fwd, False):
            ops.set_train_fwd_h2(on)
            ops.new_pack_epoch()
            m.zero_grad(set_to_none=True)
            D = m.net(x, sig)
            D.square().mean().backward()
            outs.append((D.detach().clone(), [p.grad.detach().clone() for p in m.net.parameters()]))
        torch.cuda.synchronize()
        ops.set_train_fwd_h2(h2fwd)
        eD = float((outs[0][0] - outs[1][0]).abs().max() / outs[1][0].abs().max())
        eg = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(outs[0][1], outs[1][1]))
        print(f"   forward rel diff {eD:.2e}, worst gradient rel diff {eg:.2e}", flush=True)
        assert eD < 1e-5 and eg < 1e-4
ops.set_train_fwd_h2(False)
print("probe ok")
