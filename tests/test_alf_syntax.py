"""The ALF syntax of an `--alf full` picture against the real encoder's .266 (tests/golden/ref_ctu_*_alf.npz: a whole all-intra picture with
its CTU records, the ALF decisions and the APSs written in front of it; one worker thread, see tools/refcheck/make_ctu_goldens.py::full):
the CTU-level syntax (uvg_encode_alf_bits, alf.c:1365) in the oracle's coder -> the slice data, byte for byte."""
import ctypes

import numpy as np
import pytest

import helpers as H

ALF_GOLDENS = ["ref_ctu_320x192_10_qp27_alf", "ref_ctu_192x128_8_qp22_alf"]


def oracle_rows_alf(orc, g):
    W, Hh, depth, qp = (int(a) for a in g["meta"][:4])
    prm = H.search_params(W, Hh, qp)
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    cu = np.zeros((hc * 16, wc * 16, 20), np.uint8)
    cu[:, :, :11] = g["cu"]
    cu[:, :, 12:] = np.ascontiguousarray(g["trees"].astype(np.uint32)).view(np.uint8).reshape(hc * 16, wc * 16, 8)
    co = np.ascontiguousarray(g["coeff"], np.int16)
    sa = np.ascontiguousarray(g["sao"], np.int32)
    cap = 4096 + wc * hc * 20000
    out, off = np.zeros(cap, np.uint8), np.zeros(hc + 1, np.int64)
    fn = orc.fn(depth, "encode_picture_rows_alf", ctypes.c_long)
    n_alts = int(g["alf_chroma_aps"][112])
    n = fn(ctypes.byref(prm), H.ptr(cu), H.ptr(co), H.ptr(sa), H.ptr(np.ascontiguousarray(g["alf_meta"], np.int32)), H.ptr(np.ascontiguousarray(g["alf_flags"], np.uint8)),
           H.ptr(np.ascontiguousarray(g["alf_set_idx"], np.int16)), n_alts, H.ptr(out), ctypes.c_long(cap), H.ptr(off))
    assert n >= 0
    return out[:n].copy(), off


@pytest.mark.parametrize("name", ALF_GOLDENS)
def test_slice_data_with_the_ctu_level_alf_syntax(orc, name):
    g = H.ctu_golden(name)
    assert int(g["alf_meta"][4]) == 1                       # ALF is on in these pictures
    data, off = oracle_rows_alf(orc, g)
    assert np.array_equal(off, g["row_off"])
    assert np.array_equal(data, g["row_bytes"])
    stream = g["bitstream"].tobytes()
    at = stream.find(data.tobytes())
    assert at > 0 and len(stream) - at - len(data) < 64


def test_without_the_alf_syntax_the_rows_differ(orc):
    """(the same hand-over through the plain coder is NOT the encoder's slice data: the ALF bins are really there)"""
    g = H.ctu_golden(ALF_GOLDENS[0])
    W, Hh, depth, qp = (int(a) for a in g["meta"][:4])
    data, off, _ = H.oracle_encode_rows(orc, depth, H.search_params(W, Hh, qp), dict(cu=g["cu"], trees=g["trees"], coeff=g["coeff"]), g["sao"])
    assert not np.array_equal(off, g["row_off"])
