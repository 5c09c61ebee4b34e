"""Merge candidates of inter CUs (uvg_inter_get_merge_cand, src/inter.c:1989-2192: spatial A0 / A1 / B0 / B1 / B2 with their
coding-order and duplicate tests, the temporal candidate with POC scaling and vector compression, the history table, the pairwise
average, zero vectors) -- the oracle's restatement (oracle/orc_inter_cand.c) against 1530 calls the real encoder made during two
low-delay encodes, each recorded with everything the function reads (tests/golden/ref_merge_*.npz; tools/refcheck/ctu_dump.c).
Fields of a list a candidate does not use are not compared: the reference leaves whatever its caller's array held."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("name", ["ref_merge_192x128_8_qp17_6frames", "ref_merge_136x72_10_qp27_8frames"])
def test_merge_candidates_equal_the_encoders(orc, name):
    g = {k: v for k, v in H.ctu_golden(name).items()}          # (decompress once)
    fn = orc.fn(8, "merge_candidates")
    kinds = dict(temporal=0, history=0, bi=0, sizes=set())
    for k in range(len(g["ctx"])):
        ctx = np.ascontiguousarray(g["ctx"][k])
        lcu = np.ascontiguousarray(g["lcu"][k]).copy()
        out = np.zeros((6, 7), np.int32)
        n = fn(H.ptr(ctx), H.ptr(lcu), H.ptr(np.ascontiguousarray(g["col"][k])), H.ptr(np.ascontiguousarray(g["hmvp"][k])), H.ptr(out))
        want = g["out"][k]
        assert n == int(ctx[48]), (k, n, int(ctx[48]))
        for i in range(n):
            assert out[i, 0] == want[i, 0], (k, i, "direction", out[:n].tolist(), want[:n].tolist())
            for l in (0, 1):
                if want[i, 0] & (1 << l):
                    assert out[i, 1 + l] == want[i, 1 + l] and (out[i, 3 + 2 * l:5 + 2 * l] == want[i, 3 + 2 * l:5 + 2 * l]).all(), \
                        (k, i, l, ctx[:13].tolist(), out[:n].tolist(), want[:n].tolist())
        kinds["bi"] += int((want[:n, 0] == 3).sum())
        kinds["sizes"].add(int(ctx[3]))
        kinds["history"] += int(g["hmvp"][k][0] > 0)
    assert kinds["sizes"] >= {8, 16, 32, 64} and kinds["bi"] > 100 and kinds["history"] > 100, kinds


@pytest.mark.parametrize("name", ["ref_amvp_192x128_8_qp17_6frames", "ref_amvp_136x72_10_qp27_8frames"])
def test_amvp_predictors_equal_the_encoders(orc, name):
    """uvg_inter_get_mv_cand (src/inter.c:1606-1737): the two motion vector predictors of a reference list -- left, above, temporal,
    history, zero; rounded to quarter samples."""
    g = {k: v for k, v in H.ctu_golden(name).items()}
    fn = orc.fn(8, "amvp_candidates", None)
    lists, nonzero = set(), 0
    for k in range(len(g["ctx"])):
        ctx = np.ascontiguousarray(g["ctx"][k])
        lcu = np.ascontiguousarray(g["lcu"][k]).copy()
        out = np.zeros(4, np.int32)
        fn(H.ptr(ctx), H.ptr(lcu), H.ptr(np.ascontiguousarray(g["col"][k])), H.ptr(np.ascontiguousarray(g["hmvp"][k])), H.ptr(out))
        assert (out == g["out"][k]).all(), (k, ctx[:13].tolist(), ctx[50:53].tolist(), out.tolist(), g["out"][k].tolist())
        lists.add(int(ctx[50]))
        nonzero += bool(g["out"][k].any())
    assert lists == {0, 1} and nonzero > 200
