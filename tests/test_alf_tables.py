"""Host side of the ALF reconstruction (csrc/alf_picture.hip: uvghip_alf_expand_tables; no device involved): the fixed filter sets and the
APSs' coded filters as the per-class tables the block filter takes, against a restatement here of src/alf.c:2925-2986, 5244-5279 on the
APSs of real encoder runs (tests/golden/ref_alf_*.npz)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from test_oracle_alf_picture import GOLDENS


def test_the_generated_table_header_is_current():
    path = os.path.join(H.ROOT, "uvg266_amd", "csrc", "vvc_alf_tables.h")
    before = open(path).read()
    subprocess.check_call([sys.executable, os.path.join(H.ROOT, "tools", "gen_alf_tables.py")], stdout=subprocess.DEVNULL)
    assert open(path).read() == before


@pytest.mark.parametrize("name", GOLDENS)
def test_expanded_tables(name):
    from uvg266_amd import lib
    L = lib.load_library()
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    depth = int(g["dims"][2])
    fixed = np.load(os.path.join(H.GOLDEN, "ref_alf_fixed.npy")).astype(np.int64)
    coef64, cmap = fixed[:64 * 13].reshape(64, 13), fixed[64 * 13:].reshape(16, 25)
    cv = [1 << depth] + [1 << (7 - 2 * i + depth - 8) for i in (1, 2, 3)]
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for f in range(int(g["dims"][4])):
        n_aps = int(g["meta"][f][7])
        luma = np.ascontiguousarray(g["luma_aps"][f][:max(n_aps, 1)], np.int16)
        chroma = np.ascontiguousarray(g["chroma_aps"][f], np.int16)
        lc, lk = np.full((24, 25, 13), 77, np.int16), np.full((24, 25, 13), 77, np.int16)
        cc, ck = np.full((8, 7), 77, np.int16), np.full((8, 7), 77, np.int16)
        assert L.uvghip_alf_expand_tables(depth, n_aps, p(luma), p(chroma), p(lc), p(lk), p(cc), p(ck)) == 0
        for s in range(16):
            for cl in range(25):
                assert lc[s, cl, :12].tolist() == coef64[cmap[s, cl], :12].tolist() and lc[s, cl, 12] == 1 << (depth - 1)
        assert (lk[:16] == cv[0]).all()
        for a in range(n_aps):
            aps = luma[a].astype(np.int64)
            co, ki, mp, nl = aps[:325].reshape(25, 13), aps[325:650].reshape(25, 13), aps[650:675], int(aps[676])
            for cl in range(25):
                assert lc[16 + a, cl, :12].tolist() == co[mp[cl], :12].tolist() and lc[16 + a, cl, 12] == 1 << (depth - 1)
                assert lk[16 + a, cl, :12].tolist() == [cv[int(k)] if nl else cv[0] for k in ki[mp[cl], :12]] and lk[16 + a, cl, 12] == cv[0]
        assert not lc[16 + n_aps:].any() and not lk[16 + n_aps:].any()
        nl = int(chroma[113])
        for t in range(8):
            assert cc[t, :6].tolist() == chroma[t * 7:t * 7 + 6].tolist() and cc[t, 6] == 1 << (depth - 1)
            assert ck[t, :6].tolist() == [cv[int(k)] if nl else cv[0] for k in chroma[56 + t * 7:56 + t * 7 + 6]] and ck[t, 6] == cv[0]


def test_bad_indices_are_refused():
    from uvg266_amd import lib
    L = lib.load_library()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    luma = np.zeros((1, 677), np.int16)
    luma[0, 650] = 26                     # a class mapped to a filter that cannot exist
    out = [np.zeros((24, 25, 13), np.int16) for _ in range(2)]
    assert L.uvghip_alf_expand_tables(8, 1, p(luma), None, p(out[0]), p(out[1]), None, None) != 0
    luma[0, 650] = 0; luma[0, 676] = 1; luma[0, 325] = 4          # a clip index outside 0..3
    assert L.uvghip_alf_expand_tables(8, 1, p(luma), None, p(out[0]), p(out[1]), None, None) != 0
    assert L.uvghip_alf_expand_tables(9, 0, None, None, p(out[0]), p(out[1]), None, None) != 0
