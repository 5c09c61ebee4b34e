"""The in-loop filter side of the closed loop in the oracle -- per-CTU deblocking in the encoder's order (orcN_deblock_lcu), the SAO
decision of every CTU on the block it sees at that moment with the bit estimates of the coder's two SAO models
(orcN_sao_search_picture), SAO of the deblocked picture -- against records of the real encoder (tests/golden/ref_ctu*.npz:
sao, sao_models, snap_*, final_* / filter_crc; tools/refcheck/ctu_dump.c): the decisions, the models after every CTU's SAO
syntax, the block each decision saw, and the picture the encoder returned."""
import zlib

import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32"])
def test_filters_and_sao_decisions_equal_the_encoder_run(orc, name):
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    r = H.oracle_sao_picture(orc, depth, W, Hh, qp, float(g["lam"][0]), (y, u, v), (g["rec_y"], g["rec_u"], g["rec_v"]), H.scu_from_cu(g["cu"], qp))
    for k in ("snap_y", "snap_u", "snap_v"):
        assert np.array_equal(r[k], g[k]), k + ": the block the SAO decision reads (deblocked by the CTU's own edges only)"
    assert np.array_equal(H.sao_info_comparable(r["sao"]), H.sao_info_comparable(g["sao"]))
    assert np.array_equal(r["sao_models"], g["sao_models"])
    for k in ("final_y", "final_u", "final_v"):
        assert np.array_equal(r[k], g[k]), k + ": the picture after deblocking and SAO"
    assert (g["sao"][:, :, 0] == 1).any() or depth == 8          # the 10-bit run has band decisions


@pytest.mark.parametrize("name", ["ref_ctucrc_1920x1080_8_qp22", "ref_ctucrc_1920x1080_10_qp27"])
def test_1080p_search_then_filters_equal_the_encoder_run(orc, name):
    """End to end in the oracle: the search's own reconstruction and side information feed the filters."""
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    s = H.oracle_search_picture(orc, depth, H.search_params(W, Hh, qp), y, u, v)
    r = H.oracle_sao_picture(orc, depth, W, Hh, qp, float(g["lam"][0]), (y, u, v), (s["rec_y"], s["rec_u"], s["rec_v"]), H.scu_from_cu(s["cu"], qp))
    assert np.array_equal(H.sao_info_comparable(r["sao"]), H.sao_info_comparable(g["sao"]))
    assert np.array_equal(r["sao_models"], g["sao_models"])
    assert np.array_equal(H.filter_crcs(r, W, Hh), g["filter_crc"])
