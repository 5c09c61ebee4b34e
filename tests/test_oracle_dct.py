"""Oracle transforms: reference goldens + structural properties (no upstream KAT exists for
dct/mts: tests/dct_tests.c and tests/mts_tests.c only compare implementations with generic)."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    n_sq = n_mts = 0
    for name, arrs in H.read_golden("dct", depth):
        if name == "square":
            (n, inv, bd), inp, want = arrs
            assert np.array_equal(orc.dct_nxn(depth, bd, n, inp, bool(inv)), want)
            n_sq += 1
        elif name == "mts":
            meta, inp, want = arrs
            w, h, inv, bd, color, intra, inter, lf, crlf, tr_idx, mts_type = [int(v) for v in meta[:11]]
            got = orc.mts_dct(depth, bd, color, intra, inter, 0, lf, crlf, tr_idx, w, h, inp, mts_type, inv)
            assert np.array_equal(got, want), (w, h, inv, tr_idx, lf)
            n_mts += 1
    assert n_sq >= 24 and n_mts >= 100


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_roundtrip_is_near_identity(orc, n):
    rng = np.random.default_rng(n)
    x = rng.integers(-255, 256, n * n).astype(np.int16)
    y = orc.dct_nxn(8, 8, n, orc.dct_nxn(8, 8, n, x), inverse=True)
    assert np.abs(y.astype(np.int32) - x).max() <= (1 if n <= 8 else 5)   # integer kernels are only near-orthogonal


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_dc_only(orc, n):
    """A constant block has only a DC coefficient: c * 64 * 64 * n * n >> (2 log2 n + 5) = c * n * 128 / ... exact."""
    x = np.full(n * n, 100, np.int16)
    y = orc.dct_nxn(8, 8, n, x)
    assert np.count_nonzero(y[1:]) == 0
    lg = int(np.log2(n))
    s1, s2 = lg - 1, lg + 6
    t = (64 * 100 * n + (1 << (s1 - 1))) >> s1
    assert y[0] == (64 * t * n + (1 << (s2 - 1))) >> s2


def test_square_equals_mts_path_without_skips(orc):
    rng = np.random.default_rng(3)
    for n in (4, 8, 16, 32):
        x = rng.integers(-1023, 1024, n * n).astype(np.int16)
        assert np.array_equal(orc.dct_nxn(10, 10, n, x), orc.tr(10, 10, False, 0, 0, n, n, 0, 0, x))
        assert np.array_equal(orc.dct_nxn(10, 10, n, x, True), orc.tr(10, 10, True, 0, 0, n, n, 0, 0, x))


def thin_goldens(depth):
    out = []
    for name, arrs in H.read_golden("rdoq", depth):
        if name == "thin":
            meta, src, want = arrs
            w, h, inverse, isp, mts_type, bd = (int(v) for v in meta)
            out.append((w, h, inverse, isp, mts_type, src, want))
    return out


@pytest.mark.parametrize("depth", [8, 10])
def test_thin_blocks_vs_reference(orc, depth):
    """1xN / Nx1 / 2xN / Nx2 blocks (ISP, 2xN chroma): mts_dct_generic's 2-point DCT-2 and single-pass cases."""
    g = thin_goldens(depth)
    assert len(g) == 80
    shapes = set()
    for w, h, inverse, isp, mts_type, src, want in g:
        got = orc.mts_dct(depth, depth, 0, 1, 0, isp, 0, 0, 0, w, h, src, mts_type, inverse)
        assert np.array_equal(got, want), (w, h, inverse, isp, mts_type)
        shapes.add((w, h))
    assert len(shapes) >= 12
