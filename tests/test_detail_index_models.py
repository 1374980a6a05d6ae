"""CPU models of two index transformations of art_amd/csrc/detail.hip (round 5): no GPU, no kernel code -- the arithmetic the kernels rely on,
restated in numpy and checked exhaustively.

1. detail_blocks_kernel deals its blocks to the eight XCDs by block column (workgroup b -> XCD b % 8, block idx b / 8 of that XCD's eighth): every
   block of the frame has to be computed exactly once, padding workgroups have to fall outside the frame.
2. detail_gather_kernel sums the up-to-nine blocks that contain a pixel as straight-line code: candidate s of an axis is block y / 25 - 1 + s at
   row y % 25 + 50 - 25 s, present iff s > 0 or (y >= 25 and y % 25 < 14).  That has to be the set, the rows and the ORDER of the nested loops
   it replaced (`for vblk in [max(0, y / 25 - 1), min(n - 1, y / 25 + 1)]: i = y - (vblk - 1) * 25; if 0 <= i < 64`)."""
import numpy as np
import pytest

TS, OFF, BLKRAD, NXCD = 64, 25, 1, 8


@pytest.mark.parametrize("w,h", [(8184, 5456), (11634, 8722), (333, 251), (64, 64), (25, 1000), (199, 26)])
def test_xcd_block_order_covers_every_block_once(w, h):
    nw, nh = int(np.ceil(np.float32(w) / 25)) + 2, int(np.ceil(np.float32(h) / 25)) + 2
    wk = (nw + NXCD - 1) // NXCD
    grid = NXCD * wk * nh
    b = np.arange(grid)
    xcd, idx = b % NXCD, b // NXCD
    vblk, hblk = idx // wk, xcd * wk + idx % wk
    keep = hblk < nw
    assert vblk.max() == nh - 1
    blk = vblk[keep] * nw + hblk[keep]
    assert len(blk) == nw * nh and len(np.unique(blk)) == nw * nh
    # an XCD's blocks are one contiguous stripe of block columns
    for k in range(NXCD):
        cols = np.unique(hblk[keep & (xcd == k)])
        if len(cols):
            assert cols[0] == k * wk and cols[-1] - cols[0] + 1 == len(cols)


@pytest.mark.parametrize("n", [1, 24, 25, 26, 38, 39, 63, 64, 100, 251, 5456, 8184])
def test_gather_candidates_are_the_nested_loops_terms(n):
    nblk = int(np.ceil(np.float32(n) / 25)) + 2
    for y in range(n):
        # the loops of rounds 1 - 4
        vb0, vb1 = max(0, y // OFF + BLKRAD - 2), min(nblk - 1, y // OFF + BLKRAD)
        loop = [(v, y - (v - BLKRAD) * OFF) for v in range(vb0, vb1 + 1) if 0 <= y - (v - BLKRAD) * OFF < TS]
        # the straight-line form
        yc, ry = y // OFF, y % OFF
        new = []
        for s in range(3):
            ok = s > 0 or (yc > 0 and ry < TS - 2 * OFF)
            if ok:
                new.append((yc - 1 + s, ry + 2 * OFF - OFF * s))
        assert new == loop, (n, y, new, loop)
        # what a missing candidate loads instead: the s = 2 term, always present and in range
        assert 0 <= yc + 1 < nblk and 0 <= ry < TS
