"""CPU model of the fused ShrinkAll pass (art_amd/csrc/shrinkblur.hip): no GPU, no kernel code -- the DECOMPOSITION and the
SCHEDULE the kernel relies on, restated in numpy and checked.

1. The decomposition.  boxblur(T*, A*, A*, radx, rady, W, H) (rtengine/boxblur.h:558-742) is a running sum along every row into a temporary
   and a running sum down every column; both are fp32 accumulations whose value depends on their order.  The kernel walks a band in strips of
   64 rows and blocks of 64 columns with the lags of section 14.2 (block j produces columns [64 j - rad, 64 j + 64 - rad), a strip that holds
   rows [R0, R0 + 64) produces rows [R0 - rad, R0 + 64 - rad), the strip above hands down its column sums and its last 2 rad + 1 row-blurred rows).
   `strips_blur` below does exactly that, strip by strip and block by block, carrying ONLY what the kernel carries; it has to give the bits of
   the plain whole-plane loops for every size, radius and edge position.
2. The schedule.  Per step the roles touch a circular window of 256 factor columns and three row-blurred buffers without any synchronisation
   but one barrier per step; `check_schedule` replays the steps of a strip with a tag per LDS slot (which column / which block it holds) and
   fails on any read of a slot that does not hold what the reader expects, and on any write that lands on a slot another role reads in the
   same step (except the one documented case: a wave's own update-before-factors order)."""
import numpy as np
import pytest

F = np.float32
R, C = 64, 64                 # FS_R, FS_C


# ---------------------------------------------------------------- 1. the arithmetic
def hblur_rows(s, rad):
    """boxblur.h:565-600 for all rows at once: running sum along the row (columns are the sequential axis)"""
    h, w = s.shape
    out = np.empty_like(s)
    ln = rad + 1
    t = s[:, 0].copy()
    for q in range(1, rad + 1):
        t = t + s[:, q]
    t = t / F(ln)
    out[:, 0] = t
    reclen = None
    for col in range(1, w):
        if col <= rad:
            t = (t * F(ln) + s[:, col + rad]) / F(ln + 1)
            ln += 1
            if col == rad:
                reclen = F(1) / F(ln)
        elif col < w - rad:
            t = t + (s[:, col + rad] - s[:, col - rad - 1]) * reclen
        else:
            t = (t * F(ln) - s[:, col - rad - 1]) / F(ln - 1)
            ln -= 1
        out[:, col] = t
    return out


def vblur_cols(t_, rad, vec):
    """boxblur.h:602-742 for all columns at once; `vec`: per column, the 4-lane form (true) or the scalar tail's (false)"""
    h, w = t_.shape
    out = np.empty_like(t_)
    lenf, leni = F(rad + 1), rad + 1
    tv = np.where(vec, 0, 0).astype(F)
    a = t_[0].copy()
    for i in range(1, rad + 1):
        a = a + t_[i]
    a = a / lenf
    b = t_[0] / F(leni)
    for i in range(1, rad + 1):
        b = b + t_[i] / F(leni)
    tv = np.where(vec, a, b).astype(F)
    out[0] = tv
    for row in range(1, h):
        if row <= rad:
            a = (tv * lenf + t_[row + rad]) / (lenf + F(1))
            b = (tv * F(leni) + t_[row + rad]) / F(leni + 1)
            lenf, leni = lenf + F(1), leni + 1
            tv = np.where(vec, a, b).astype(F)
        elif row < h - rad:
            d = t_[row + rad] - t_[row - rad - 1]
            tv = np.where(vec, tv + d * (F(1) / F(2 * rad + 1)), tv + d / F(2 * rad + 1)).astype(F)
            lenf, leni = F(2 * rad + 1), 2 * rad + 1
        else:
            a = (tv * lenf - t_[row - rad - 1]) / (lenf - F(1))
            b = (tv * F(leni) - t_[row - rad - 1]) / F(leni - 1)
            lenf, leni = lenf - F(1), leni - 1
            tv = np.where(vec, a, b).astype(F)
        out[row] = tv
    return out


def strips_blur(s, rad):
    """The kernel's walk: strips of R rows, blocks of C columns, the lags, the hand-over -- nothing else is carried."""
    h, w = s.shape
    nov = 2 * rad + 1
    nb = (w + rad + C - 1) // C
    vec = np.arange(w) < (w // 4) * 4
    out = np.full_like(s, np.nan)
    nstrips = (h + R - 1) // R
    hand = None                                   # what the strip above left: per block (rows [nov, <=64 cols]), column sums
    for strip in range(nstrips):
        R0, Rb = strip * R, min(strip * R + R, h)
        first, last = strip == 0, Rb == h
        ro0, ro1 = max(0, R0 - rad), (h if last else Rb - rad)
        # row-sum state of the strip's rows (one lane each)
        tval = np.zeros(Rb - R0, F)
        hlen, reclen = rad + 1, None
        left_for_next = []
        for J in range(nb):
            X0 = J * C
            cols = [c for c in range(X0 - rad, X0 + C - rad) if 0 <= c < w]
            hb = {}                               # row-blurred values of this block: column -> values of the strip's rows
            for col in cols:
                if col == 0:
                    tval = s[R0:Rb, 0].copy()
                    for q in range(1, rad + 1):
                        tval = tval + s[R0:Rb, q]
                    tval = tval / F(hlen)
                elif col <= rad:
                    tval = (tval * F(hlen) + s[R0:Rb, col + rad]) / F(hlen + 1)
                    hlen += 1
                    if col == rad:
                        reclen = F(1) / F(hlen)
                elif col < w - rad:
                    tval = tval + (s[R0:Rb, col + rad] - s[R0:Rb, col - rad - 1]) * reclen
                else:
                    tval = (tval * F(hlen) - s[R0:Rb, col - rad - 1]) / F(hlen - 1)
                    hlen -= 1
                hb[col] = tval.copy()
            if not cols:
                left_for_next.append(None)
                continue
            own = np.stack([hb[c] for c in cols], axis=1)                       # (rows of the strip, columns of the block)
            if first:
                rows0, blk = R0, own
            else:
                above, tv_in = hand[J]
                rows0, blk = R0 - nov, np.concatenate([above, own], axis=0)      # image row of blk[0]
            cv = vec[cols]
            tv = np.zeros(len(cols), F) if first else tv_in.copy()
            lenf, leni = (F(rad + 1), rad + 1) if first else (F(nov), nov)
            rlen = F(1) / F(nov)
            hbrow = lambda r: blk[r - rows0]
            for r in range(ro0, ro1):
                if r == 0:
                    a = hbrow(0).copy()
                    for i in range(1, rad + 1):
                        a = a + hbrow(i)
                    a = a / lenf
                    b = hbrow(0) / F(leni)
                    for i in range(1, rad + 1):
                        b = b + hbrow(i) / F(leni)
                    tv = np.where(cv, a, b).astype(F)
                elif r <= rad:
                    a = (tv * lenf + hbrow(r + rad)) / (lenf + F(1))
                    b = (tv * F(leni) + hbrow(r + rad)) / F(leni + 1)
                    lenf, leni = lenf + F(1), leni + 1
                    tv = np.where(cv, a, b).astype(F)
                elif r < h - rad:
                    d = hbrow(r + rad) - hbrow(r - rad - 1)
                    tv = np.where(cv, tv + d * rlen, tv + d / F(leni)).astype(F)
                else:
                    a = (tv * lenf - hbrow(r - rad - 1)) / (lenf - F(1))
                    b = (tv * F(leni) - hbrow(r - rad - 1)) / F(leni - 1)
                    lenf, leni = lenf - F(1), leni - 1
                    tv = np.where(cv, a, b).astype(F)
                out[r, cols] = tv
            left_for_next.append(None if last else (own[R - nov:R].copy(), tv.copy()))
        hand = left_for_next
    return out


@pytest.mark.parametrize("w,h,rad", [(64, 64, 1), (65, 65, 2), (130, 129, 3), (200, 150, 6), (131, 257, 7), (300, 70, 15), (97, 200, 11), (256, 128, 4)])
def test_strip_walk_gives_the_bits_of_the_whole_plane_loops(w, h, rad):
    rng = np.random.default_rng(w * 1000 + h + rad)
    s = rng.uniform(0, 1, (h, w)).astype(F) ** 3
    with np.errstate(all="ignore"):
        vec = np.arange(w) < (w // 4) * 4
        ref = vblur_cols(hblur_rows(s, rad), rad, vec)
        got = strips_blur(s, rad)
    assert not np.isnan(got).any()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


# ---------------------------------------------------------------- 2. the schedule of a strip's workgroup
NE = 13                       # elementwise waves
SWIN = 256


def check_schedule(w, rad, nrows_strip=R, first=False, last=False):
    nov = 2 * rad + 1
    nb = (w + rad + C - 1) // C
    srows = nrows_strip + rad
    S = np.full((srows, SWIN), -10 ** 9)                 # tag: image column the slot holds
    HBtag = [None, None, None]                           # per buffer: (block whose row sums it holds, block whose column sums it holds)
    wave_of_row = lambda wr: wr % NE
    for T in range(nb + 4):
        writes, reads = {}, {}                           # slot -> wave, for the cross-role check of this step

        # elementwise: update of block T - 3 (reads), then the factors of block T (writes), per wave in that order
        upd_reads = {}
        if T >= 3 and T - 3 < nb:
            J = T - 3
            for lane in range(C):
                col = J * C - rad + lane
                if 0 <= col < w:
                    for wr in range(0, nrows_strip if not last else srows):
                        assert S[wr, col & (SWIN - 1)] == col, f"update of block {J} finds column {S[wr, col & 255]} instead of {col}"
                        upd_reads[(wr, col & (SWIN - 1))] = wave_of_row(wr)
            assert HBtag[J % 3] == (J, J), f"update of block {J}: buffer holds {HBtag[J % 3]}"
        if T < nb:
            for lane in range(C):
                col = T * C + lane
                if col < w:
                    for wr in range(srows):
                        slot = (wr, col & (SWIN - 1))
                        if slot in upd_reads:            # the documented case: same rows -> same wave, update first
                            assert upd_reads[slot] == wave_of_row(wr)
                        writes[slot] = wave_of_row(wr)
                        S[wr, col & (SWIN - 1)] = col
        # row sums of block T - 1: read columns [X0 - 2 rad - 1, X0 + 64) of the strip's rows
        if 1 <= T <= nb:
            J = T - 1
            for jj in range(C):
                col = J * C - rad + jj
                if not (0 <= col < w):
                    continue
                need = []
                if col == 0:
                    need = list(range(0, rad + 1))
                elif col <= rad:
                    need = [col + rad]
                elif col < w - rad:
                    need = [col + rad, col - rad - 1]
                else:
                    need = [col - rad - 1]
                for c2 in need:
                    for wr in range(rad, rad + nrows_strip):
                        slot = (wr, c2 & (SWIN - 1))
                        # (written in an earlier step: this step's factors must not have touched the slot)
                        assert slot not in writes, f"row sums of block {J} read column {c2} while this step's factors overwrite its slot"
                        assert S[wr, c2 & (SWIN - 1)] == c2
            HBtag[J % 3] = (J, None)
        # column sums of block T - 2 in place
        if 2 <= T <= nb + 1:
            J = T - 2
            assert HBtag[J % 3] == (J, None), f"column sums of block {J}: buffer holds {HBtag[J % 3]}"
            HBtag[J % 3] = (J, J)
        # the three buffers of a step are distinct
        assert len({(T - 1) % 3, (T - 2) % 3, (T - 3) % 3}) == 3
    return True


@pytest.mark.parametrize("rad", [1, 2, 3, 6, 7, 9, 15])
@pytest.mark.parametrize("w", [64, 100, 257, 640])
def test_roles_never_meet_on_a_slot(rad, w):
    assert check_schedule(w, rad)
    assert check_schedule(w, rad, last=True)


def test_hand_over_lag():
    """strip s publishes block J at its step J + 4 (the stores of step J + 2 / J + 3 have left), strip s + 1 sends for block T at its step T:
    it runs at least four steps behind, so a band keeps at most NB / 4 strips busy -- the reason for one launch over all 45 bands"""
    nb = (4096 + 6 + 63) // 64
    publish_step = lambda J: J + 4
    need_at = lambda T: T                                 # prefetch_hand(T) at step T needs progress >= T + 1, i.e. block T published
    lag = max(publish_step(J) - need_at(J) for J in range(nb))
    assert lag == 4
    assert 15 * (nb // lag) < 256 <= 45 * (nb // lag)     # one channel cannot fill 256 CUs, three can


# ---------------------------------------------------------------- 3. the hand-over ring (round 5)
def simulate_hand_over_ring(nstrips, nb, ring, rng, greedy_below=False, wait=True):
    """The strips of ONE band, each a sequence of steps T = 0 .. nb + 3 whose hand-over wave does, in program order (shrinkblur.hip, the
    `else` role of the step loop): publish progress = T - 3 (T >= 4); wait for progress[s - 1] >= T + 1 and READ block T of slot s % ring
    (T < nb); STORE block T - 2's rows and block T - 3's column sums into slot (s + 1) % ring.  Strips advance in an arbitrary interleaving
    (any strip that is not blocked may take its next step: workgroups run at unrelated speeds); stores land the moment they are issued -- the
    worst case for a slot that is rewritten behind its reader.  Every read has to find what strip s - 1 wrote."""
    step = [0] * nstrips                      # next step of strip s
    prog = [0] * nstrips
    rows_tag = [[None] * nb for _ in range(ring)]      # which strip wrote block j's rows / column sums of ring slot k (per band)
    tv_tag = [[None] * nb for _ in range(ring)]
    started = 1                               # strips come into being in ticket order
    reads = 0
    while any(step[s] < nb + 4 for s in range(nstrips)):
        runnable = []
        for s in range(started):
            T = step[s]
            if T >= nb + 4:
                continue
            if wait and s > 0 and T < nb and prog[s - 1] < T + 1:
                continue                      # (blocked in prefetch_hand: the strip's other roles wait at the barrier)
            runnable.append(s)
        if started < nstrips and (not runnable or rng.random() < 0.3):
            started += 1
            continue
        assert runnable, "deadlock"
        s = max(runnable) if greedy_below else runnable[int(rng.integers(len(runnable)))]
        T = step[s]
        last = s == nstrips - 1
        if not last and T >= 4:
            prog[s] = T - 3
        if s > 0 and T < nb:
            k = s % ring
            assert rows_tag[k][T] == s - 1 and tv_tag[k][T] == s - 1, (s, T, rows_tag[k][T], tv_tag[k][T])
            reads += 1
        if not last:
            k = (s + 1) % ring
            if 0 <= T - 2 < nb:
                rows_tag[k][T - 2] = s
            if 0 <= T - 3 < nb:
                tv_tag[k][T - 3] = s
        step[s] = T + 1
    return reads


@pytest.mark.parametrize("greedy", [False, True])
def test_two_hand_over_slots_per_band_are_enough(greedy):
    """FS_RING = 2: strip s + 1 rewrites the slot strip s reads only behind its own wait for strip s's counter, and strip s counts a block
    three steps after it consumed it.  Random interleavings, and the adversarial one (always the lowest strip that can move = the writer below
    races ahead of its reader as far as the protocol lets it)."""
    rng = np.random.default_rng(5)
    for trial in range(40):
        nstrips = int(rng.integers(2, 12)); nb = int(rng.integers(1, 20))
        assert simulate_hand_over_ring(nstrips, nb, 2, rng, greedy) == (nstrips - 1) * nb


def test_the_model_sees_a_strip_that_does_not_wait():
    """the model does see what it is there to exclude: without the wait for the strip above a strip reads slots nobody has filled yet (or,
    with few slots, ones the strip below has already rewritten).  (With the wait even ONE slot per band passes this model -- a strip stores
    block j two steps after it read block j -- two keep a wave from reading and rewriting the same addresses.)"""
    rng = np.random.default_rng(6)
    with pytest.raises(AssertionError):
        for trial in range(20):
            simulate_hand_over_ring(6, 12, 2, rng, wait=False)
    for trial in range(20):
        simulate_hand_over_ring(6, 12, 1, rng)
