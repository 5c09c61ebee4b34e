"""HIP RDOQ vs the reference-run goldens and vs the oracle on large random batches (bit-exact levels)."""
import numpy as np
import pytest

import helpers as H
from test_oracle_rdoq import rdoq_goldens

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_vs_reference_goldens(hip, depth):
    import torch
    from uvg266_amd import api
    k = 0
    for c in rdoq_goldens(depth):
        w, h = c["w"], c["h"]
        coef = torch.from_numpy(c["coef"].reshape(1, h, w)).cuda()
        lv, s, has = api.rdoq_batch(coef, depth, c["color"], c["block_type"], c["cbf_u"], c["lfnst"], c["mts"], c["qps"], c["lam"], c["ctx"])
        assert np.array_equal(lv.cpu().numpy().ravel(), c["want"]), {k_: v for k_, v in c.items() if k_ not in ("coef", "want", "ctx")}
        assert int(s[0]) == int(np.abs(c["want"].astype(np.int64)).sum()) and int(has[0]) == int(c["want"].any())
        k += 1
    assert k >= 150


def _blocks(rng, n, w, h, depth, qps, style):
    """Transformed-block statistics: energy compaction towards the top-left, scaled so that levels of 0..~20 come out."""
    step = 2.0 ** ((qps - 4) / 6.0) * (1 << max(0, 15 - depth - ((int(np.log2(w)) + int(np.log2(h))) >> 1)))
    yy, xx = np.mgrid[0:h, 0:w]
    amp = step * [6.0, 1.5, 0.7, 3.0, 0.55][style % 5] / (1.0 + 0.35 * (xx + yy))
    v = rng.uniform(-1, 1, (n, h, w)) * amp
    v[rng.random(n) < 0.05] = 0
    return np.clip(v, -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("depth", [8, 10])
def test_random_batches_vs_oracle(hip, orc, depth):
    """Hundreds of blocks per launch (lanes of a wave take different paths through the coefficient-group decisions),
    every shape, luma / chroma, intra / inter, LFNST and MTS zero-out variants, arbitrary context states."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(100 + depth)
    cases = [(w, h, 0, 1, 0, 0, 0) for w in (4, 8, 16, 32) for h in (4, 8, 16, 32)]
    cases += [(8, 8, 1, 1, 0, 0, 0), (16, 16, 2, 1, 1, 0, 0), (4, 4, 2, 2, 0, 0, 0), (32, 32, 0, 2, 0, 0, 0), (16, 8, 0, 1, 0, 1, 0),
              (8, 8, 0, 1, 0, 2, 0), (4, 4, 0, 1, 0, 1, 0), (32, 32, 0, 1, 0, 0, 2), (32, 16, 0, 1, 0, 0, 1), (16, 16, 0, 2, 0, 0, 3)]
    checked = 0
    for i, (w, h, color, bt, cbf_u, lfnst, mts) in enumerate(cases):
        n = 200 if w * h <= 256 else 70
        qp = int(rng.integers(17, 45))
        qps = qp + 6 * (depth - 8)
        lam = 0.57 * 2.0 ** ((qp - 12) / 3.0) * float(rng.uniform(0.5, 1.5))
        ctx = rng.integers(0, 256, 244).astype(np.uint8)
        coef = _blocks(rng, n, w, h, depth, qps, i)
        if lfnst:
            coef[:, 4:, :] = 0; coef[:, :, 4:] = 0
        if mts and (w == 32 or h == 32):
            coef[:, 16:, :] = 0; coef[:, :, 16:] = 0
        lv, s, has = api.rdoq_batch(torch.from_numpy(coef).cuda(), depth, color, bt, cbf_u, lfnst, mts, qps, lam, ctx)
        lv, s, has = lv.cpu().numpy(), s.cpu().numpy(), has.cpu().numpy()
        nz = 0
        for b in range(n):
            want, ws = orc.rdoq(depth, coef[b], w, h, color, bt, cbf_u, lfnst, mts, qps, lam, ctx)
            assert np.array_equal(lv[b].ravel(), want), (w, h, color, bt, lfnst, mts, b)
            assert s[b] == ws and has[b] == int(want.any())
            nz += int(want.any())
        assert nz > n // 10
        checked += n
    assert checked > 3000


def test_workspace_is_required(hip):
    import ctypes
    import torch
    from uvg266_amd import lib
    coef = torch.zeros((4, 8, 8), dtype=torch.int16, device="cuda")
    out = torch.zeros_like(coef)
    ctx = (ctypes.c_uint8 * 244)()
    rc = hip.uvghip_rdoq_batch(8, coef.data_ptr(), out.data_ptr(), 8, 8, 4, 0, 1, 0, 0, 0, 22, 10.0, ctypes.cast(ctx, ctypes.c_void_p),
                               None, 0, None, None, None)
    assert rc != 0 and b"workspace" in hip.uvghip_last_error()
    assert hip.uvghip_rdoq_workspace_bytes(8, 8, 4) == (3 * 64 + 64) * 4 * 8
