"""CPU models of loop / index arithmetic that round 5 changed in denoise.hip without changing what is computed: which coefficients
mad_sample_kernel's batched loop visits (every 32nd chunk of 256, now eight chunks per iteration with their loads first), which elements
mad_window_kernel's clamped, unconditional loads count, and which map pixels a workgroup of chroma_map_lds_kernel takes (two rows of four
pixels per thread and batch, persistent workgroups).  Each model walks the kernel's loops with the kernel's expressions."""
import numpy as np
import pytest


def sample_indices_batched(n, grid, U=8):
    """mad_sample_kernel: chunk0 = blockIdx * 32, += gridDim * 32 * U; chunk k of a batch = chunk0 + k * gridDim * 32; 256 threads per chunk"""
    seen = []
    cstride = grid * 32
    for b in range(grid):
        chunk0 = b * 32
        while chunk0 * 256 < n:
            for k in range(U):
                base = (chunk0 + k * cstride) * 256
                seen.extend(i for i in range(base, base + 256) if i < n)
            chunk0 += cstride * U
    return seen


def sample_indices_plain(n, grid):
    """the loop before: one chunk per iteration"""
    seen = []
    for b in range(grid):
        chunk = b * 32
        while chunk * 256 < n:
            seen.extend(i for i in range(chunk * 256, min(chunk * 256 + 256, n)))
            chunk += grid * 32
    return seen


@pytest.mark.parametrize("n", [1, 255, 256, 8191, 8192 * 3 + 17, 64 * 32 * 256 * 8 + 5, 700001])
@pytest.mark.parametrize("grid", [1, 3, 64])
def test_mad_sample_batches_visit_the_same_coefficients_once(n, grid):
    a, b = sample_indices_batched(n, grid), sample_indices_plain(n, grid)
    assert len(a) == len(set(a))
    assert sorted(a) == sorted(b)
    # every 32nd chunk of 256
    assert all((i // 256) % 32 == 0 for i in a)


@pytest.mark.parametrize("n,grid,U", [(1, 1, 8), (1000, 2, 8), (256 * 192 * 8 + 77, 192, 8), (300000, 192, 8), (65537, 7, 8)])
def test_mad_window_counts_every_element_once(n, grid, U):
    """i0 = blockIdx * 256 + thread, += gridDim * 256 * U; element k of a batch = i0 + k * gridDim * 256, loaded from min(i, n - 1), counted iff i < n"""
    stride = grid * 256
    count = np.zeros(n, np.int32)
    for b in range(grid):
        for t in range(256):
            i0 = b * 256 + t
            while i0 < n:
                for k in range(U):
                    i = i0 + k * stride
                    src = i if i < n else n - 1          # (the clamped load: always inside the band)
                    assert 0 <= src < n
                    if i < n:
                        count[i] += 1
                i0 += stride * U
    assert (count == 1).all()


@pytest.mark.parametrize("hei,wid,grid", [(1, 1, 1), (5, 4097, 5), (300, 9000, 256), (257, 1024, 256), (1024, 5000, 256), (3, 100, 3)])
def test_chroma_map_lds_workgroups_cover_the_map_once(hei, wid, grid):
    """yb = blockIdx, += 2 gridDim; row of batch element k = yb + (k / 4) gridDim, column = x0 + (k % 4) * 1024 + thread, x0 += 4096;
    loads come from clamped (row, column), results are stored iff both are inside"""
    NR, NPX = 2, 4
    assert grid <= hei                                    # (launch: min(CUs, hei) workgroups)
    count = np.zeros((hei, wid), np.int32)
    threads = np.arange(1024)
    for b in range(grid):
        yb = b
        while yb < hei:
            for x0 in range(0, wid, NPX * 1024):
                for k in range(NR * NPX):
                    ii = yb + (k // NPX) * grid
                    jj = x0 + (k % NPX) * 1024 + threads
                    lr = min(ii, hei - 1)
                    lc = np.minimum(jj, wid - 1)
                    assert 0 <= lr < hei and lc.min() >= 0 and lc.max() < wid
                    if ii < hei:
                        ok = jj < wid
                        count[ii, jj[ok]] += 1
            yb += NR * grid
    assert (count == 1).all()
