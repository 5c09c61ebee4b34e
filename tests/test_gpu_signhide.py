"""Sign-data hiding on the device (cfg.signhide_enable: presets slow / slower): uvghip_rdoq_signhide_batch and
uvghip_quant_signhide_batch vs the reference-run records and the oracle."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_goldens(hip, depth):
    import torch
    from uvg266_amd import api
    n_rdoq = n_quant = 0
    for c in H.signhide_goldens(depth):
        blk = torch.from_numpy(np.stack([c["coef"].reshape(c["h"], c["w"])] * 3)).cuda()
        if c["kind"] == 0:
            lev, _, _ = api.rdoq_batch(blk, depth, c["color"], c["cu_type"], c["cbf_u"], c["lfnst"], c["mts"], c["qps"], c["lam"], c["ctx"], signhide=True)
            n_rdoq += 1
        else:
            lev = api.quant_signhide_batch(blk, depth, c["qps"], bool(c["ts"]), bool(c["intra"]), c["lfnst"])
            n_quant += 1
        for k in (0, 2):
            assert np.array_equal(lev[k].cpu().numpy().ravel(), c["q"]), (c["kind"], c["w"], c["h"], c["color"], c["lfnst"], c.get("mts"))
    assert n_rdoq == 200 and n_quant == 200


@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (32, 32), (16, 4), (8, 32)])
def test_random_batches_vs_oracle(hip, orc, w, h):
    import torch
    from uvg266_amd import api, pipeline
    rng = np.random.default_rng(w + 7 * h)
    ctx = pipeline.synthetic_rdoq_ctx()
    n = 150 if w * h <= 256 else 40
    fall = 1.0 / (1.0 + 0.3 * (np.arange(w)[None, :] + np.arange(h)[:, None]))
    coef = (rng.normal(0, 1, (n, h, w)) * fall * rng.choice([60, 300, 1500], (n, 1, 1))).astype(np.int16)
    dcoef = torch.from_numpy(coef).cuda()
    for color, qps, lam in ((0, 27, 14.0), (1, 33, 50.0)):
        lev, _, _ = api.rdoq_batch(dcoef, 8, color, 1, 0, 0, 0, qps, lam, ctx, signhide=True)
        plain, _, _ = api.rdoq_batch(dcoef, 8, color, 1, 0, 0, 0, qps, lam, ctx)
        lev, plain = lev.cpu().numpy(), plain.cpu().numpy()
        for i in range(n):
            want, _ = orc.rdoq_sh(8, coef[i].ravel(), w, h, color, 1, 0, 0, 0, qps, lam, ctx)
            assert np.array_equal(lev[i].ravel(), want), (color, i)
        if w * h > 16:
            assert (lev != plain).any()
        q = api.quant_signhide_batch(dcoef, 8, qps, False, True, 0).cpu().numpy()
        for i in range(0, n, 3):
            assert np.array_equal(q[i].ravel(), orc.quant_sh(8, coef[i].ravel(), w, h, 8, qps, 0, 1, 0)), (color, i)
