"""Oracle RDOQ (oracle/orc_rdoq.c) vs the vectors the reference's uvg_rdoq produced (tests/golden/ref_rdoq_*.bin,
dumped by tools/refcheck/rc_rdoq.inc: random transformed blocks, standard and arbitrary CABAC context states)."""
import numpy as np
import pytest

import helpers as H


def rdoq_goldens(depth):
    """-> list of dicts(w, h, color, block_type, cbf_u, lfnst, mts, qps, nz, lam, ctx, coef, want)"""
    out = []
    for name, arrs in H.read_golden("rdoq", depth):
        assert name == "rdoq"
        meta, lam, ctx, coef, want = arrs
        w, h, color, bt, cbf_u, lfnst, mts, qps, bd, nz = (int(v) for v in meta)
        assert bd == depth
        out.append(dict(w=w, h=h, color=color, block_type=bt, cbf_u=cbf_u, lfnst=lfnst, mts=mts, qps=qps, nz=nz, lam=float(lam[0]),
                        ctx=ctx, coef=coef, want=want))
    return out


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    g = rdoq_goldens(depth)
    assert len(g) >= 150
    shapes, feats, nonzero = set(), set(), 0
    for c in g:
        got, abs_sum = orc.rdoq(depth, c["coef"], c["w"], c["h"], c["color"], c["block_type"], c["cbf_u"], c["lfnst"], c["mts"], c["qps"],
                                c["lam"], c["ctx"])
        assert np.array_equal(got, c["want"]), c
        assert abs_sum == int(np.abs(c["want"].astype(np.int64)).sum())
        shapes.add((c["w"], c["h"])); nonzero += c["nz"] > 0
        feats.add((c["color"] > 0, c["block_type"], c["lfnst"] > 0, c["mts"] > 0))
    assert len(shapes) == 16 and nonzero >= 80 and len(feats) >= 6


def test_ctx_struct_size():
    """uvghip_rdoq_ctx_t (include/uvg266_hip.h) and the oracle's struct are 244 bytes: 4 + 24 + 3*42 + 2*40 + 10."""
    assert 4 + 24 + 3 * 42 + 2 * 40 + 10 == H.Oracle.RDOQ_CTX_BYTES


@pytest.mark.parametrize("depth", [8, 10])
def test_properties(orc, depth):
    """RDOQ never raises |level| above plain rounding's, keeps signs, and an all-zero block stays zero."""
    rng = np.random.default_rng(depth)
    ctx = rng.integers(0, 256, 244).astype(np.uint8)
    for w, h in ((4, 4), (8, 16), (32, 32), (16, 4)):
        coef = (rng.normal(0, 300, w * h) / (1 + 0.3 * (np.add.outer(np.arange(h), np.arange(w))).ravel())).astype(np.int16)
        qps = 27 + 6 * (depth - 8)
        lv, _ = orc.rdoq(depth, coef, w, h, 0, 1, 0, 0, 0, qps, 20.0, ctx)
        plain = np.zeros(w * h, np.int16)
        orc.fn(depth, "quant", None)(H.ptr(coef), H.ptr(plain), w, h, depth, qps, 0, 1)
        assert np.all(np.abs(lv) <= np.abs(plain) + 1)
        assert np.all((lv == 0) | (np.sign(lv) == np.sign(coef)))
        z, s = orc.rdoq(depth, np.zeros(w * h, np.int16), w, h, 0, 1, 0, 0, 0, qps, 20.0, ctx)
        assert not z.any() and s == 0
