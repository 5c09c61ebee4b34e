"""Oracle ALF vs vectors dumped from the reference strategies (classification, 7x7 / 5x5 filters, statistics)."""
import numpy as np
import pytest

import helpers as H


def alf_goldens(depth):
    out = {"luma": [], "chroma": [], "stats": []}
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
