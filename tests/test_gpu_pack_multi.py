"""The one-launch multi-tensor weight pack (i2r_conv2d_pack_multi_f32: 64 x 64 transpose tiles through LDS) against the
single-tensor pack of the same weights, bit for bit: forward and data-gradient layouts, 1x1 and 3x3, channel counts that
are not multiples of 4 / 16 / 64, tensors smaller than one block and larger than many, and the words between the views
untouched."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # Cout, Cin, k
    (64, 64, 3), (128, 12, 3), (12, 128, 3), (130, 100, 3), (512, 512, 3), (256, 768, 3), (3, 5, 3), (33, 65, 3),
    (384, 128, 1), (128, 384, 1), (100, 130, 1), (1536, 512, 1), (5, 3, 1), (65, 33, 1), (256, 256, 1),
]


def test_multi_pack_equals_single_tensor_pack():
    from imageni2r_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(3)
    blk = _lib.query("i2r_conv2d_pack_block_floats")
    ws = [torch.randn(co, ci, k, k, device="cuda", generator=g) for co, ci, k in SHAPES]
    jobs = [(w, dg, _lib.query("i2r_conv2d_packed_floats", w.shape[0], w.shape[1], w.shape[2], dg))
            for w in ws for dg in (0, 1)]
    GUARD = 8
    pool = torch.full((sum(n + GUARD for _, _, n in jobs),), float("nan"), device="cuda")
    rows, bmap, views, off = [], [], [], 0
    for t, (w, dg, n) in enumerate(jobs):
        v = pool[off: off + n]
        off += n + GUARD
        rows.append([w.data_ptr(), v.data_ptr(), w.shape[0], w.shape[1], w.shape[2], dg])
        bmap.extend([t, b] for b in range((n + blk - 1) // blk))
        views.append(v)
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    bmap_t = torch.tensor(bmap, dtype=torch.int32).cuda()
    ops.pack_multi(table, bmap_t, len(bmap))
    torch.cuda.synchronize()
    off = 0
    for (w, dg, n), v in zip(jobs, views):
        ref = ops.conv2d_pack(w, dgrad=bool(dg))
        assert ref.numel() == n
        assert torch.equal(v, ref.reshape(-1)), (tuple(w.shape), dg)
        assert torch.isnan(pool[off + n: off + n + GUARD]).all(), ("guard words written", tuple(w.shape), dg)
        off += n + GUARD


def test_multi_split_f16_pack_equals_the_per_layer_packs():
    """i2r_conv2d_pack_h2_multi against i2r_conv2d_pack_weight_h2 / _h2_dgrad, byte for byte."""
    from imageni2r_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(4)
    blk = _lib.query("i2r_conv2d_pack_h2_block_pairs")
    ws = [torch.randn(co, ci, k, k, device="cuda", generator=g) * 3 for co, ci, k in SHAPES]
    jobs = []
    for w in ws:
        co, ci, k, _ = w.shape
        jobs.append((w, 0, _lib.query("i2r_conv2d_packed_h2_bytes", co, ci, k)))
        jobs.append((w, 1, _lib.query("i2r_conv2d_packed_h2_bytes", ci, co, k)))
    GUARD = 64
    pool = torch.full((sum(n + GUARD for _, _, n in jobs),), 0xA5, dtype=torch.uint8, device="cuda")
    rows, bmap, views, off = [], [], [], 0
    for t, (w, dg, n) in enumerate(jobs):
        assert n % 4 == 0
        v = pool[off: off + n]
        off += n + GUARD
        rows.append([w.data_ptr(), v.data_ptr(), w.shape[0], w.shape[1], w.shape[2], dg])
        bmap.extend([t, b] for b in range((n // 4 + blk - 1) // blk))
        views.append(v)
    ops.pack_h2_multi(torch.tensor(rows, dtype=torch.int64).cuda(), torch.tensor(bmap, dtype=torch.int32).cuda(),
                      len(bmap))
    torch.cuda.synchronize()
    off = 0
    for (w, dg, n), v in zip(jobs, views):
        ref = ops.conv2d_pack_h2_dgrad(w) if dg else ops.conv2d_pack_h2(w)
        assert ref.numel() == n and torch.equal(v, ref), (tuple(w.shape), dg)
        assert bool((pool[off + n: off + n + GUARD] == 0xA5).all()), ("guard bytes written", tuple(w.shape), dg)
        off += n + GUARD
