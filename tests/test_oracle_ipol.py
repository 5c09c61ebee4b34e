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
