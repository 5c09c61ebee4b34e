"""Oracle interpolation vs vectors dumped from the reference (sample_* strategies through
get_extended_block, and the four-block fractional-ME functions).  No upstream unit test exists."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    ns = nf = 0
    for name, arrs in H.read_golden("ipol", depth):
        if name == "sample":
            (PW, PH, x0, y0, w, h, fx, fy, chroma, _), plane, px, hi = arrs
            plane = plane.reshape(PH, PW)
            assert np.array_equal(orc.ipol_sample(depth, plane, PW, PH, x0, y0, w, h, fx, fy, chroma, False), px)
            assert np.array_equal(orc.ipol_sample(depth, plane, PW, PH, x0, y0, w, h, fx, fy, chroma, True), hi)
            ns += 1
        elif name == "fme":
            (PW, PH, bx, by, w, h), plane, cur, cands, costs = arrs
            got = orc.frac_satd(depth, cur.reshape(64, 64), 0, 0, plane.reshape(PH, PW), PW, PH, bx, by, w, h, cands)
            assert np.array_equal(got, costs)
            nf += 1
    assert ns >= 12 and nf >= 6


def test_integer_phase_is_identity(orc):
    rng = np.random.default_rng(0)
    p = rng.integers(0, 256, (40, 48)).astype(np.uint8)
    got = orc.ipol_sample(8, p, 48, 40, 5, 7, 16, 8, 0, 0)
    assert np.array_equal(got.reshape(8, 16), p[7:15, 5:21])


@pytest.mark.parametrize("depth", [8, 10])
def test_extended_block_properties(orc, depth):
    """get_extended_block / _wraparound restatement (ipol-generic.c:761-883): inside -> offsets into the frame and an
    untouched buffer; outside -> rows clamped, columns clamped or taken modulo the width, SIMD rows zero."""
    rng = np.random.default_rng(depth)
    PH, PW, S = 40, 56, 60
    plane = rng.integers(0, 1 << depth, (PH, S)).astype(H.px_dtype(depth))
    for wrap in (0, 1):
        for _ in range(200):
            bw, bh = int(rng.choice([4, 8, 16])), int(rng.choice([4, 8, 16]))
            pl, pr, pt, pb, pbs = (int(v) for v in rng.integers(0, 5, 5))
            bx, by = int(rng.integers(-30, PW + 20)), int(rng.integers(-30, PH + 20))
            if wrap and (bx - pl < -PW or bx + bw + pr > 2 * PW):
                continue
            inside, off, es, buf = orc.get_extended_block(depth, wrap, plane, PW, PH, bx, by, bw, bh, pl, pr, pt, pb, pbs)
            x0, x1 = bx - pl, bx + bw + pr
            y0, y1 = by - pt, by + bh + pb + pbs
            want_inside = x0 >= 0 and (x1 < PW if wrap else x1 <= PW) and y0 >= 0 and y1 <= PH
            assert bool(inside) == want_inside
            if inside:
                assert off == y0 * S + x0 and es == S and np.all(buf == 0x5a)
                continue
            ys = np.clip(np.arange(y0, y1 - pbs), 0, PH - 1)
            xs = np.arange(x0, x1)
            xs = np.where(xs < 0, xs + PW, np.where(xs >= PW, xs - PW, xs)) if wrap else np.clip(xs, 0, PW - 1)
            want = np.concatenate([plane[np.ix_(ys, xs)], np.zeros((pbs, x1 - x0), plane.dtype)])
            assert es == x1 - x0 and np.array_equal(buf.reshape(-1, es), want)
