"""Oracle LFNST vs vectors dumped from the reference's uvg_fwd_lfnst / uvg_inv_lfnst (plain C, src/transform.c);
no upstream unit test exists for LFNST.  The kernel tables themselves are a fixture too (ref_lfnstmat.bin =
the reference's responses to 128 * e_i) from which the headers are generated; this file also checks that the
generated headers are in sync with the fixture and that the kernels are what the standard says they are:
near-orthonormal 16-row bases."""
import os
import subprocess
import sys

import numpy as np

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ref_goldens(orc):
    n = 0
    for name, (hdr, src, want) in H.read_golden("lfnst", 8):
        w, h, mode, idx, inverse = (int(v) for v in hdr)
        c = src.copy()
        lw, lh = w.bit_length() - 1, h.bit_length() - 1
        (orc.lib.orc_lfnst_inv if inverse else orc.lib.orc_lfnst_fwd)(H.ptr(c), w, h, mode, lw, lh, idx)
        assert np.array_equal(c, want), (w, h, mode, idx, inverse)
        n += 1
    assert n >= 300


def test_tables_in_sync_and_orthonormal():
    b = open(os.path.join(ROOT, "tests/golden/ref_lfnstmat.bin"), "rb").read()
    m8 = np.frombuffer(b, np.int16, 4 * 2 * 16 * 48, 4).reshape(4, 2, 16, 48).astype(np.int64)
    m4 = np.frombuffer(b, np.int16, 4 * 2 * 16 * 16, 4 + 4 * 2 * 16 * 48 * 2).reshape(4, 2, 16, 16).astype(np.int64)
    for m in (m8, m4):
        for s in range(4):
            for k in range(2):
                g = m[s, k] @ m[s, k].T                      # rows: unit norm 128 (+- rounding), mutually orthogonal
                assert np.all(np.abs(np.diag(g) - 128 * 128) < 400) and np.abs(g - np.diag(np.diag(g))).max() < 400
    before = {p: open(os.path.join(ROOT, p)).read() for p in ("oracle/orc_lfnst_tables.h", "uvg266_amd/csrc/vvc_lfnst_tables.h")}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/gen_lfnst_tables.py")], stdout=subprocess.DEVNULL)
    for p, txt in before.items():
        assert open(os.path.join(ROOT, p)).read() == txt, f"{p} is stale: run tools/gen_lfnst_tables.py"


def test_roundtrip_and_untouched_region(orc):
    """16 kept coefficients: inv(fwd(x)) reproduces the projection of x on the kernel's 16 rows (to rounding);
    nothing outside the top-left 8x8 (4x4) region is touched; lfnst_idx 0 leaves the TU alone."""
    rng = np.random.default_rng(3)
    for w, h in ((16, 16), (4, 16), (32, 8), (8, 8), (4, 4)):
        for mode in (0, 1, 2, 18, 34, 35, 50, 66):
            x = rng.integers(-300, 301, (h, w)).astype(np.int16)
            lw, lh = w.bit_length() - 1, h.bit_length() - 1
            y = x.copy(); orc.lib.orc_lfnst_fwd(H.ptr(y), w, h, mode, lw, lh, 0)
            assert np.array_equal(y, x)
            y = x.copy(); orc.lib.orc_lfnst_fwd(H.ptr(y), w, h, mode, lw, lh, 1)
            sb = 8 if (w >= 8 and h >= 8) else 4
            mask = np.zeros((h, w), bool); mask[:sb, :sb] = True
            if sb == 8: mask[4:8, 4:8] = False
            assert np.array_equal(y[~mask], x[~mask])
            z = y.copy(); orc.lib.orc_lfnst_inv(H.ptr(z), w, h, mode, lw, lh, 1)
            z2 = z.copy(); orc.lib.orc_lfnst_fwd(H.ptr(z2), w, h, mode, lw, lh, 1)
            # fwd(inv(fwd(x))) == fwd(x) up to rounding of the two 7-bit normalisations
            assert np.abs(z2[mask].astype(np.int32) - y[mask].astype(np.int32)).max() <= 16     # rows are orthonormal only to the precision of 8-bit entries
