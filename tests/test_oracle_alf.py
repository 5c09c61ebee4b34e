"""Oracle ALF vs vectors dumped from the reference strategies (classification, 7x7 / 5x5 filters, statistics)."""
import numpy as np
import pytest

import helpers as H


def alf_goldens(depth):
    out = {"luma": [], "chroma": [], "stats": [], "ccstats": []}
    for name, arrs in H.read_golden("alf", depth):
        out[name].append(arrs)
    return out


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    g = alf_goldens(depth)
    assert len(g["luma"]) >= 2 and len(g["chroma"]) >= 2 and len(g["stats"]) >= 1
    for (W, Hh, shift, cs), plane, cls, coef, clip, want in g["luma"]:
        plane = plane.reshape(Hh, W)
        got_cls = orc.alf_classify_frame(depth, plane, W, Hh, shift)
        assert np.array_equal(got_cls, cls.reshape(cs, cs)[: Hh // 4, : W // 4])
        dst = np.zeros_like(plane)
        orc.alf_filter_rect(depth, plane, dst, W, Hh, 0, 0, W, Hh, False, coef, clip, got_cls)
        assert np.array_equal(dst.ravel(), want)
    for (CW, CH), plane, coef, clip, want in g["chroma"]:
        plane = plane.reshape(CH, CW)
        dst = np.zeros_like(plane)
        orc.alf_filter_rect(depth, plane, dst, CW, CH, 0, 0, CW, CH, True, coef, clip, None)
        assert np.array_equal(dst.ravel(), want)
    (W, Hh, shift, cs), plane, cls, _, _, _ = g["luma"][0]
    (rx, ry, rw, rh), org, ee, yv, pa = g["stats"][0]
    plane = plane.reshape(Hh, W)
    gcls = np.ascontiguousarray(cls.reshape(cs, cs)[: Hh // 4, : W // 4])
    e, y, p = orc.alf_stats_rect(depth, org.reshape(Hh, W), plane, W, Hh, rx, ry, rw, rh, False, gcls)
    assert np.array_equal(e.ravel(), ee) and np.array_equal(y.ravel(), yv) and np.array_equal(p, pa)


def cc_ctus(W, Hh):
    """The 3 x 3 CTUs of the CC-ALF statistics records (chroma rectangles), in the records' order."""
    return [((k % 3) * 32, (k // 3) * 32, min(32, W // 2 - (k % 3) * 32), min(32, Hh // 2 - (k // 3) * 32)) for k in range(9)]


@pytest.mark.parametrize("depth", [8, 10])
def test_cc_alf_statistics(orc, depth):
    """get_blk_stats_cc_alf (alf.c:2613, a static function: tools/refcheck/rc_alfstatic.c compiles the reference's alf.c into the dev tool to
    reach it) for every CTU of a 136 x 136 picture, Cb and Cr: ordinary content and black / white extremes."""
    import ctypes
    recs = alf_goldens(depth)["ccstats"]
    assert len(recs) == 2
    for (W, Hh, it), luma, rec_u, rec_v, org_u, org_v, ee, yv, pix in recs:
        CW, CH = W // 2, Hh // 2
        luma = np.ascontiguousarray(luma.reshape(Hh, W))
        fn = orc.fn(depth, "cc_alf_stats_rect", None)
        for k, (x, y, w, h) in enumerate(cc_ctus(W, Hh)):
            for c, (org, rec) in enumerate(((org_u, rec_u), (org_v, rec_v))):
                e, yy, p = np.zeros(49, np.int64), np.zeros(7, np.int32), np.zeros(1, np.int64)
                fn(H.ptr(np.ascontiguousarray(org)), CW, H.ptr(np.ascontiguousarray(rec)), CW, H.ptr(luma), W, W, Hh, x, y, w, h, H.ptr(e), H.ptr(yy), H.ptr(p))
                i = k * 2 + c
                assert np.array_equal(e, ee[i * 49:(i + 1) * 49]) and np.array_equal(yy, yv[i * 7:(i + 1) * 7]) and p[0] == pix[i], (it, k, c)
