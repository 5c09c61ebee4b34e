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


@pytest.mark.parametrize("dmin,dmax,combine", [(1, 3, 1), (1, 2, 1), (1, 1, 1), (2, 4, 0), (2, 2, 0), (3, 4, 0)])
def test_other_pu_depth_ranges_against_the_oracle(dmin, dmax, combine):
    """--pu-depth-intra other than 1-4: leaves above the 4x4 depth (the walk borrows the depth waves' scratch for them), and with
    depth_max < 3 the 64x64 candidate (combine_intra_cus) is only built after the walk instead of beside it (ctu_core.h post64).
    Content incl. flat areas where the 64x64 CU wins.  Ranges that start below depth 1 only without combine_intra_cus: the reference
    then also combines at depth 1 (search.c:2082-2143, every depth without a search), the kernel does not, and uvghip_ctu_plan_create
    refuses the combination (tests/test_gpu_ctu_search.py)."""
    orc = H.load_oracle()
    wins64 = 0
    cases = []
    for name in ("ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12"):          # (their sources: smooth enough for 64x64 CUs)
        W, Hh, depth, qp, y, u, v = H.golden_source(H.ctu_golden(name))
        cases.append((W, Hh, depth, qp, (y, u, v)))
    cases.append((200, 136, 10, 27, H.varied_picture(200, 136, 7, 10)))
    cases.append((128, 128, 8, 45, H.varied_picture(128, 128, 1011, 8)))
    for W, Hh, depth, qp, pic in cases:
        prm = H.search_params(W, Hh, qp)
        prm.depth_min, prm.depth_max, prm.combine_intra_cus = dmin, dmax, combine
        r = H.emul_search_picture(depth, prm, *pic)
        o = H.oracle_search_picture(orc, depth, prm, *pic)
        assert np.array_equal(H.ctu_crcs(r, W, Hh), H.ctu_crcs(o, W, Hh)), (W, Hh, depth, qp)
        assert np.array_equal(r["models"], o["models"]), (W, Hh, depth, qp)
        wins64 += int((o["cu"][:Hh // 4:16, :W // 4:16, 1] == 6).sum())
    assert wins64 > 0 or not combine
