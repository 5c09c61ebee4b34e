"""uvg_quant_cbcr_residual (joint Cb-Cr residual coding, quant-generic.c:241-442) composed from the oracle's pieces against the
reference-run records -- the composition the staged HIP path (uvghip_quant_cbcr_residual_batch) follows."""
import numpy as np
import pytest

import helpers as H


def cdiv(a, b):
    """C integer division (truncates toward zero) on int arrays."""
    return (np.abs(a) // b) * np.sign(a)


def oracle_quant_cbcr(orc, d, c):
    """-> (ret, levels, u_rec block, v_rec block)."""
    w, h = c["w"], c["h"]
    cb = (c["uref"][:h, :w].astype(np.int32) - c["upred"][:h, :w]).astype(np.int16).astype(np.int32)
    cr = (c["vref"][:h, :w].astype(np.int32) - c["vpred"][:h, :w]).astype(np.int16).astype(np.int32)
    mask = c["joint"] * (-1 if c["sign"] else 1)
    comb = {2: cdiv(4 * cb + 2 * cr, 5), -2: cdiv(4 * cb - 2 * cr, 5), 3: cdiv(cb + cr, 2), -3: cdiv(cb - cr, 2),
            1: cdiv(4 * cr + 2 * cb, 5), -1: cdiv(4 * cr - 2 * cb, 5)}[mask].astype(np.int16)
    color = 2 if c["joint"] == 1 else 1
    coef = orc.tr(d, d, False, 0, 0, w, h, 0, 0, comb.ravel())
    if c["rdoq"] and (w > 4 or not c["rdoq_skip"]):
        q, _ = orc.rdoq(d, coef, w, h, color, c["cu_type"], c["cbf_u"], 0, 0, c["qps"], c["lam"], c["ctx"])
    else:
        q = orc.quant(d, coef, w, h, d, c["qps"], 0, c["intra"])
    has = bool(q.any())
    pu, pv = c["upred"][:h, :w].astype(np.int32), c["vpred"][:h, :w].astype(np.int32)
    if not has or c["early_skip"]:
        return (c["joint"] if has else 0), q, pu, pv
    r = orc.tr(d, d, True, 0, 0, w, h, 0, 0, orc.dequant(d, q, w, h, d, c["qps"], 0)).reshape(h, w).astype(np.int32)
    ur, vr = {2: (r, r >> 1), -2: (r, (-r) >> 1), 3: (r, r), -3: (r, -r), 1: (r >> 1, r), -1: ((-r) >> 1, r)}[mask]
    mx = (1 << d) - 1
    wrap = lambda a: a.astype(np.int16).astype(np.int32)
    return c["joint"], q, np.clip(wrap(wrap(ur) + pu), 0, mx), np.clip(wrap(wrap(vr) + pv), 0, mx)


@pytest.mark.parametrize("depth", [8, 10])
def test_jccr_goldens(orc, depth):
    g = H.jccr_goldens(depth)
    assert len(g) == 120
    seen = set()
    for c in g:
        ret, q, ur, vr = oracle_quant_cbcr(orc, depth, c)
        tag = (c["w"], c["h"], c["joint"], c["sign"], c["rdoq"], c["early_skip"])
        assert ret == c["ret"] and np.array_equal(q, c["q"]), tag
        w, h, so = c["w"], c["h"], c["so"]
        assert np.array_equal(c["urec"][: so * h].reshape(h, so)[:, :w], ur), tag
        assert np.array_equal(c["vrec"][: so * h].reshape(h, so)[:, :w], vr), tag
        seen.add((c["joint"], c["sign"], bool(c["ret"])))
    assert len(seen) == 12
