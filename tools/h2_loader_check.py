"""Host-side enumeration of the weight-stage loader of conv_fwd_h2_kernel (csrc/conv_fwd_h2.hip): for every
instantiation (KS, NW, TERMS) list each thread's global piece offset and LDS destination exactly as the kernel computes
them, and check that (a) every fetched piece lies inside its packed row, (b) every LDS write (16 B) lies inside the
dynamic LDS the host allocates, (c) used pieces cover the hi (and, TERMS = 3, lo) half of every row exactly once,
(d) spare slots only touch row padding.  Run before any GPU launch of a changed loader: python tools/h2_loader_check.py
"""
import itertools

BM = 128


def check(KS, NW, TERMS):
    NT = 64 * NW
    KC = 48 if KS == 3 else 64
    ROWL = KC * 2 + 16
    RB = 2 * KC * 2
    PPR = RB // 16
    PPRL = PPR if TERMS == 3 else PPR // 2
    NA = (BM * PPRL + NT - 1) // NT
    ABUF = 2 * BM * ROWL
    lds_alloc = 4 * 128 * (48 * 2 + 16) if KS == 3 else 4 * 128 * (64 * 2 + 16)
    assert 2 * ABUF == lds_alloc, (KS, ABUF, lds_alloc)
    seen = {}
    spare = 0
    for tid, j in itertools.product(range(NT), range(NA)):
        idx = tid + NT * j
        used = idx < BM * PPRL
        row, piece = idx // PPRL, idx % PPRL
        lo = piece >= PPR // 2
        if used:
            assert row < BM
            goff = row * RB + piece * 16
            assert goff + 16 <= (row + 1) * RB  # inside this row of the packed chunk
            lds = (BM * ROWL if lo else 0) + row * ROWL + (piece - lo * (PPR // 2)) * 16
            assert (lds % ROWL if not lo else (lds - BM * ROWL) % ROWL) + 16 <= KC * 2  # data part of the row
            assert (row, piece) not in seen
            seen[(row, piece)] = lds
        else:
            spare += 1
            lds = (idx % BM) * ROWL + KC * 2
            assert lds % ROWL == KC * 2  # the 16-byte padding of a hi row
        for buf in (0, 1):
            assert 0 <= buf * ABUF + lds and buf * ABUF + lds + 16 <= lds_alloc
    assert len(seen) == BM * PPRL and len(set(seen.values())) == len(seen)
    if TERMS == 1:
        assert all(p < PPR // 2 for _, p in seen)
    # what the MFMA loop reads: hi rows always, lo rows only for TERMS = 3 -- all written above
    for row, ks in itertools.product(range(BM), range(KC // 16)):
        for hh in (0, 1):
            rd = row * ROWL + ks * 32 + 16 * hh
            assert rd in seen.values()
            if TERMS == 3:
                assert BM * ROWL + rd in seen.values()
    return NA, spare


if __name__ == "__main__":
    for KS, NW, TERMS in [(3, 4, 3), (3, 8, 3), (1, 4, 3), (1, 8, 3), (3, 4, 1), (3, 8, 1)]:
        na, spare = check(KS, NW, TERMS)
        print(f"KS={KS} NW={NW} TERMS={TERMS}: {na} pieces/thread, {spare} spare slots -- ok")
