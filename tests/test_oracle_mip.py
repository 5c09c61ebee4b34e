"""Oracle MIP vs vectors dumped from the reference's generic mip_predict (no upstream unit test).  The weight tables are
a fixture too (ref_mipmat.bin = the reference's responses to unit perturbations of the reduced boundary) from which
the headers are generated; checked here to be in sync."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    n = 0
    for name, (hdr, top, left, want) in H.read_golden("mip", depth):
        w, h, mode, transp = (int(v) for v in hdr)
        out = np.zeros(w * h, want.dtype)
        orc.fn(depth, "mip_predict", None)(H.ptr(top), H.ptr(left), w, h, mode, transp, H.ptr(out))
        assert np.array_equal(out, want), (w, h, mode, transp)
        n += 1
    assert n >= 100


def test_tables_in_sync():
    before = {p: open(os.path.join(ROOT, p)).read() for p in ("oracle/orc_mip_tables.h", "uvg266_amd/csrc/vvc_mip_tables.h")}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/gen_mip_tables.py")], stdout=subprocess.DEVNULL)
    for p, txt in before.items():
        assert open(os.path.join(ROOT, p)).read() == txt, f"{p} is stale: run tools/gen_mip_tables.py"


@pytest.mark.parametrize("depth", [8, 10])
def test_flat_boundary_predicts_flat(orc, depth):
    """A constant boundary makes every input difference zero, so the prediction is the boundary value itself -- for
    the large blocks (size id 2) at any level, for the small ones (whose first input is 2^(depth-1) - boundary) at
    mid-grey."""
    dt = np.uint8 if depth == 8 else np.uint16
    for w, h in ((4, 4), (8, 8), (4, 16), (16, 16), (32, 8), (32, 32)):
        sid = 0 if (w, h) == (4, 4) else (1 if (w == 4 or h == 4 or (w, h) == (8, 8)) else 2)
        for mode in range((16, 8, 6)[sid]):
            for val in ((0, 77 << (depth - 8), (1 << depth) - 1, 1 << (depth - 1)) if sid == 2 else (1 << (depth - 1),)):
                top = np.full(70, val, dt); left = np.full(70, val, dt)
                out = np.zeros(w * h, dt)
                orc.fn(depth, "mip_predict", None)(H.ptr(top), H.ptr(left), w, h, mode, 0, H.ptr(out))
                assert int(out.min()) == int(out.max()) == val, (w, h, mode, val, out[:4])
