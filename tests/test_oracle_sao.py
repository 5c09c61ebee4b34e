"""Oracle SAO vs vectors dumped from the reference (uvg_sao_reconstruct on a real frame state,
calc_sao_edge_dir on packed CTU copies).  No upstream unit test exists for SAO."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    nr = ns = 0
    for name, arrs in H.read_golden("sao", depth):
        if name == "recon":
            (pw, ph, ps, fx, fy, w, h, typ, eo, is_v), plane, bp, offs, want = arrs
            plane = plane.reshape(ph + 1, ps)
            got = np.full((ph, ps), 0x55, plane.dtype)
            buf = np.ascontiguousarray(plane)             # row 0 is the spare row above the picture
            rec = buf[1:]
            orc.sao_reconstruct_rect(depth, rec, got, pw, ph, fx, fy, w, h, typ, eo, bp, offs, is_v)
            assert np.array_equal(got.ravel(), want)
            nr += 1
        elif name == "stats":
            (PW, PH), po, pr, rects, edge, band = arrs
            e, b = orc.sao_stats_rects(depth, po.reshape(PH, PW), pr.reshape(PH, PW), rects)
            assert np.array_equal(e.ravel(), edge) and np.array_equal(b.ravel(), band)
            ns += 1
    assert nr >= 8 and ns == 1


def test_edge_offsets_hand_cases(orc):
    """Offset rounding (C division truncating toward zero), clipping, sign constraints, strict-< class choice."""
    edge = np.zeros((3, 4, 2, 5), np.int32)
    # rect 0, class 1: cat1 mean +2.5 -> (25+5)/10 = 3; cat2 negative mean -> 0; cat3 mean -2.6 -> (-26+5)/10 = -2; cat4 huge -> -7
    edge[0, 1, 0] = [99, 25, -30, -26, -900]
    edge[0, 1, 1] = [50, 10, 10, 10, 10]
    # rect 1: all counts zero -> every class has ddistortion 0, first class wins with zero offsets
    # rect 2: classes 2 and 3 identical and best -> class 2 (strict <)
    edge[2, 2, 0] = edge[2, 3, 0] = [0, 40, 0, 0, 0]
    edge[2, 2, 1] = edge[2, 3, 1] = [0, 10, 0, 0, 0]
    p, d = np.zeros((3, 8), np.int32), np.zeros(3, np.int32)
    orc.lib.orc_sao_edge_offsets(H.ptr(edge), None, 3, H.ptr(p), H.ptr(d))
    assert p[0].tolist() == [2, 1, 0, 0, 3, 0, -2, -7]
    assert d[0] == (10 * 9 - 2 * 3 * 25) + (10 * 4 - 2 * -2 * -26) + (10 * 49 - 2 * -7 * -900)
    assert p[1].tolist() == [2, 0, 0, 0, 0, 0, 0, 0] and d[1] == 0
    assert p[2].tolist() == [2, 2, 0, 0, 4, 0, 0, 0] and d[2] == 10 * 16 - 2 * 4 * 40
