"""The CTU search kernel's LOGIC without a GPU: uvg266_amd/csrc/ctu_core.h built for the host with one emulated lane (tests/emul/)
against the reference-run goldens.  (Barriers, lane mapping and the cross-workgroup wavefront are what the -m gpu tests add.)"""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32"])
def test_emulated_kernel_equals_the_reference_run(name):
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    r = H.emul_search_picture(depth, H.search_params(W, Hh, qp), y, u, v)
    assert np.array_equal(r["models"], g["models"])
    h4, w4 = Hh // 4, W // 4
    assert np.array_equal(r["cu"][:h4, :w4], g["cu"][:h4, :w4]) and np.array_equal(r["trees"][:h4, :w4], g["trees"][:h4, :w4])
    for p in ("rec_y", "rec_u", "rec_v"):
        assert np.array_equal(r[p], g[p]), p
    assert np.array_equal(H.ctu_crcs(r, W, Hh)[:, 2], H.ctu_crcs(dict(r, coeff=g["coeff"]), W, Hh)[:, 2])


def test_emulated_kernel_1080p_crcs():
    g = H.ctu_golden("ref_ctucrc_1920x1080_10_qp27")
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    r = H.emul_search_picture(depth, H.search_params(W, Hh, qp), y, u, v)
    assert np.array_equal(H.ctu_crcs(r, W, Hh), g["crc"])


def test_outcome_does_not_depend_on_when_a_cu_cost_arrives():
    """The depth pipeline evaluates a CU unsplit on another wave while its children are tried; the reference knows the CU's cost
    first and uses it to cut the children short.  Whether that cost arrives at once or only after the last child must not change
    anything (every cut decides "not split", and the final comparison decides the same)."""
    g = H.ctu_golden("ref_ctu_416x240_10_qp37")
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    r = H.emul_search_picture(depth, H.search_params(W, Hh, qp), y, u, v, lazy=True)
    assert np.array_equal(r["models"], g["models"]) and np.array_equal(r["rec_y"], g["rec_y"])
    assert np.array_equal(r["cu"][:Hh // 4, :W // 4], g["cu"][:Hh // 4, :W // 4])
    g = H.ctu_golden("ref_ctu_832x480_8_qp22")
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    r = H.emul_search_picture(depth, H.search_params(W, Hh, qp), y, u, v, lazy=True)
    assert np.array_equal(r["models"], g["models"]) and np.array_equal(r["rec_y"], g["rec_y"])
