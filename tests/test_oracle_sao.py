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
