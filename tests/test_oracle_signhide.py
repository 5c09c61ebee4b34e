"""Sign-data hiding (cfg.signhide_enable, presets slow / slower): the oracle's uvg_rdoq + uvg_rdoq_sign_hiding and uvg_quant
with its sign-bit hiding against the reference-run records."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("depth", [8, 10])
def test_signhide_goldens(orc, depth):
    g = H.signhide_goldens(depth)
    nr = nq = differs = 0
    for c in g:
        if c["kind"] == 0:
            q, _ = orc.rdoq_sh(depth, c["coef"], c["w"], c["h"], c["color"], c["cu_type"], c["cbf_u"], c["lfnst"], c["mts"], c["qps"], c["lam"], c["ctx"])
            plain, _ = orc.rdoq(depth, c["coef"], c["w"], c["h"], c["color"], c["cu_type"], c["cbf_u"], c["lfnst"], c["mts"], c["qps"], c["lam"], c["ctx"])
            differs += not np.array_equal(q, plain)
            nr += 1
        else:
            q = orc.quant_sh(depth, c["coef"], c["w"], c["h"], depth, c["qps"], c["ts"], c["intra"], c["lfnst"])
            nq += 1
        assert np.array_equal(q, c["q"]), (c["kind"], c["w"], c["h"], c["color"], c["lfnst"])
    assert nr == 200 and nq == 200 and differs > 100


def test_hidden_sign_parity_property(orc):
    """What the decoder relies on: in every coefficient group whose first and last non-zero levels are >= 4 scan positions apart,
    the parity of the sum of the levels equals the sign bit of the first non-zero one."""
    scan = np.zeros(64, np.uint32); scg = np.zeros(4, np.uint32)
    orc.lib.orc8_rdoq_scans(8, 8, H.ptr(scan), H.ptr(scg))
    rng = np.random.default_rng(1)
    checked = 0
    for _ in range(200):
        coef = (rng.normal(0, 300, 64)).astype(np.int16)
        q = orc.quant_sh(8, coef, 8, 8, 8, 27, 0, 1, 0)
        for cg in range(4):
            lv = q[scan[cg * 16:(cg + 1) * 16]].astype(np.int64)
            nz = np.nonzero(lv)[0]
            if len(nz) and nz[-1] - nz[0] >= 4:
                assert (int(lv[nz[0]:nz[-1] + 1].sum()) & 1) == int(lv[nz[0]] < 0)
                checked += 1
    assert checked > 100
