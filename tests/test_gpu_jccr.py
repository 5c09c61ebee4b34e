"""uvghip_quant_cbcr_residual_batch (joint Cb-Cr residual coding, staged launches) vs the reference-run records of
uvg_quant_cbcr_residual: levels, return value and the whole of both output buffers."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_goldens(hip, depth):
    import torch
    from uvg266_amd import api
    px = H.px_dtype(depth)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    seen = set()
    for c in H.jccr_goldens(depth):
        w, h, S, so = c["w"], c["h"], c["S"], c["so"]
        fill = 7 if depth == 8 else 0x0707
        urec, vrec = (torch.full((S, so), fill, dtype=dev(c["uref"]).dtype, device="cuda") for _ in range(2))
        tus = api.make_tus([[0, 0]])
        coeff, ret = api.quant_cbcr_residual_batch(dev(c["uref"]), dev(c["vref"]), dev(c["upred"]), dev(c["vpred"]), urec, vrec, tus, w, h, depth,
                                                   c["joint"], c["sign"], qp_scaled=c["qps"], slice_is_intra=bool(c["intra"]), cu_type=c["cu_type"],
                                                   rdoq=bool(c["rdoq"]), rdoq_skip=bool(c["rdoq_skip"]), cbf_u=c["cbf_u"], lam=c["lam"], ctx=c["ctx"],
                                                   early_skip=bool(c["early_skip"]))
        tag = (w, h, c["joint"], c["sign"], c["rdoq"], c["early_skip"])
        assert int(ret[0]) == c["ret"], tag
        assert np.array_equal(coeff[0].cpu().numpy().ravel(), c["q"]), tag
        for got, want in ((urec, c["urec"]), (vrec, c["vrec"])):
            assert np.array_equal(got.cpu().numpy().astype(px).ravel()[: so * S], want[: so * S]), tag
        seen.add((c["joint"], c["sign"], bool(c["ret"])))
    assert len(seen) == 12


def test_batch_of_many_tus_vs_oracle(hip, orc):
    """A 416x240 chroma plane pair cut into 8x8 TUs, one launch per (joint_cb_cr, sign): every TU vs the oracle composition."""
    import torch
    from uvg266_amd import api, layout, pipeline
    from test_oracle_jccr import oracle_quant_cbcr
    rng = np.random.default_rng(4)
    W, Hh, n = 208, 120, 8
    u = rng.integers(40, 216, (Hh, W)).astype(np.uint8)
    v = rng.integers(40, 216, (Hh, W)).astype(np.uint8)
    du = rng.integers(-40, 41, (Hh, W))
    up = np.clip(u.astype(np.int32) - du, 0, 255).astype(np.uint8)
    noise = rng.integers(-2, 3, (Hh, W))
    xy = layout.block_grid(W, Hh, n)
    tus = api.make_tus(xy)
    dev = lambda a: torch.from_numpy(a).cuda()
    ctx = pipeline.synthetic_rdoq_ctx()
    for joint, sign in ((1, 0), (2, 1), (3, 0), (3, 1)):
        # Cr residual = +-Cb residual / 2: the correlation the (joint, sign) mode exists for
        vp = np.clip(v.astype(np.int32) - (-1 if sign else 1) * du // 2 + noise, 0, 255).astype(np.uint8)
        urec, vrec = torch.zeros_like(dev(u)), torch.zeros_like(dev(v))
        coeff, ret = api.quant_cbcr_residual_batch(dev(u), dev(v), dev(up), dev(vp), urec, vrec, tus, n, n, 8, joint, sign, qp_scaled=20, rdoq=True,
                                                   lam=1.5, ctx=ctx)
        coeff, ret, ur, vr = coeff.cpu().numpy(), ret.cpu().numpy(), urec.cpu().numpy(), vrec.cpu().numpy()
        for i, (x, y) in enumerate(xy[::3]):
            i *= 3
            c = dict(w=n, h=n, joint=joint, sign=sign, qps=20, intra=1, cu_type=1, rdoq=1, rdoq_skip=0, cbf_u=0, early_skip=0, lam=1.5, ctx=ctx,
                     uref=u[y:y + n, x:x + n], vref=v[y:y + n, x:x + n], upred=up[y:y + n, x:x + n], vpred=vp[y:y + n, x:x + n])
            wret, wq, wu, wv = oracle_quant_cbcr(orc, 8, c)
            assert ret[i] == wret and np.array_equal(coeff[i].ravel(), wq), (joint, sign, x, y)
            assert np.array_equal(ur[y:y + n, x:x + n], wu) and np.array_equal(vr[y:y + n, x:x + n], wv), (joint, sign, x, y)
        assert ret.any()
