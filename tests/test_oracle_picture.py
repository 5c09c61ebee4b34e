"""Oracle (oracle/orc_picture.c) pinned against the reference's own known-answer
tests and against vectors dumped from the reference's generic strategy."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("test", [0, 1, 2])
@pytest.mark.parametrize("log_w", [2, 3, 4, 5, 6])
def test_satd_kat(orc, test, log_w):
    b1, b2 = H.satd_kat_buffers(test, log_w)
    n = 1 << log_w
    r1, r2 = orc.satd_nxn(8, b1, b2, n), orc.satd_nxn(8, b2, b1, n)
    assert r1 == r2 == H.SATD_KAT[test][log_w - 2]


@pytest.mark.parametrize("case", H.SAD_BORDER_KAT)
def test_sad_border_kat(orc, case):
    (rx, ry), expect = case
    pic, ref = np.ascontiguousarray(H.SAD_PIC_8x8), np.ascontiguousarray(H.SAD_REF_8x8)
    assert orc.image_calc_sad(8, pic, ref, 8, 8, 0, 0, rx, ry, 8, 8) == expect


@pytest.mark.parametrize("dim", H.SAD_DIMS)
def test_reg_sad_dims(orc, dim):
    w, h = dim
    a, b = H.sad_big_planes()
    naive = int(np.abs(a[:h, :w].astype(np.int64) - b[:h, :w]).sum())
    assert orc.reg_sad(8, a, b, w, h, 64, 64) == naive
    z, m = np.zeros((64, 64), np.uint8), np.full((64, 64), 255, np.uint8)
    assert orc.reg_sad(8, z, m, w, h, 64, 64) == 255 * w * h      # overflow case, sad_tests.c:318-343


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_intra_sad_kat(orc, n):
    # tests/intra_sad_tests.c: black/white = 255*N*N; symmetric
    z, m = np.zeros(n * n, np.uint8), np.full(n * n, 255, np.uint8)
    assert orc.sad_nxn(8, z, m, n) == orc.sad_nxn(8, m, z, n) == 255 * n * n


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    dt = H.px_dtype(depth)
    seen = set()
    for name, arrs in H.read_golden("picture", depth):
        seen.add(name)
        if name == "strided":
            (w, h, S, qoff), a, b, qbase, outs, res = arrs
            assert a.dtype == dt
            assert orc.reg_sad(depth, a, b, w, h, S, S) == outs[0]
            assert orc.satd_any_size(depth, w, h, a, S, b, S) == outs[1]
            assert orc.pixels_calc_ssd(depth, a, b, S, S, w, h) == outs[2]
            q = orc.satd_any_size_quad(depth, w, h, qbase, [0, qoff, 2 * qoff, 3 * qoff], 64, a, S)
            assert list(q) == list(outs[3:7])
            assert np.array_equal(orc.generate_residual(depth, a, b, w, h, S, S), res)
        elif name == "image_calc_sad":
            (W, Hh, bw, bh), pa, pb, cases, outs = arrs
            pa, pb = pa.reshape(Hh, W), pb.reshape(Hh, W)
            for (px, py, rx, ry), o in zip(cases.reshape(-1, 4), outs):
                assert orc.image_calc_sad(depth, pa, pb, W, Hh, px, py, rx, ry, bw, bh) == o
        elif name == "nxn":
            (n,), pr64, p0, p1, orig, outs = arrs
            assert orc.sad_nxn(depth, pr64, pr64[64 * 64:], n) == outs[0]
            assert orc.satd_nxn(depth, pr64, pr64[64 * 64:], n) == outs[1]
            preds = np.concatenate([p0, p1])
            assert list(orc.sad_nxn_dual(depth, preds, orig, n)) == list(outs[2:4])
            assert list(orc.satd_nxn_dual(depth, preds, orig, n)) == list(outs[4:6])
    assert seen == {"strided", "image_calc_sad", "nxn"}
