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
    checked = total_nz = 0
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
        if i % 5 in (0, 1, 3):
            assert nz > n // 10
        checked += n; total_nz += nz
    assert checked > 3000 and total_nz > 1500


@pytest.mark.parametrize("depth", [8, 10])
def test_bin_budget_edge_vs_oracle(hip, orc, depth):
    """Blocks whose regular-bin budget (28 bins per 16 coefficients, rdo.c:1470) runs out somewhere inside a coefficient
    group, right at its end, or just not: flat spectra over a fine sweep of amplitudes, so that the kernel's choice between
    the anti-diagonal phases and the position-by-position walk of a group lands on both sides of every threshold."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(7 + depth)
    checked = 0
    for w, h, n in ((4, 4, 1600), (8, 8, 480), (16, 16, 96), (8, 4, 300)):
        qp = 27
        qps = qp + 6 * (depth - 8)
        lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
        ctx = rng.integers(40, 216, 244).astype(np.uint8)
        step = 2.0 ** ((qps - 4) / 6.0) * (1 << max(0, 15 - depth - ((int(np.log2(w)) + int(np.log2(h))) >> 1)))
        amp = np.linspace(0.4, 6.0, n)[:, None, None] * step
        coef = np.clip(rng.uniform(-1, 1, (n, h, w)) * amp, -32768, 32767).astype(np.int16)
        lv, s, has = api.rdoq_batch(torch.from_numpy(coef).cuda(), depth, 0, 1, 0, 0, 0, qps, lam, ctx)
        lv = lv.cpu().numpy()
        for b in range(n):
            want, ws = orc.rdoq(depth, coef[b], w, h, 0, 1, 0, 0, 0, qps, lam, ctx)
            assert np.array_equal(lv[b].ravel(), want), (w, h, b)
        checked += n
    assert checked > 2000


@pytest.mark.parametrize("depth", [8, 10])
def test_large_levels_vs_oracle(hip, orc, depth):
    """Levels far above 254 at a low QP, dense enough to exhaust the regular-bin budget: the 32x32 kernel keeps clamped levels
    (254 + parity) in LDS and must fetch the exact neighbours for the template sums of the bypass-coded part (their int16
    accumulator wraps, rdo.c:846-871); other shapes keep exact levels throughout."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(41 + depth)
    for w, h, n in ((32, 32, 10), (16, 16, 12), (32, 16, 6)):
        for qp, amp in ((2, 32767), (7, 9000), (12, 2500)):
            qps = qp + 6 * (depth - 8)
            lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
            ctx = rng.integers(30, 226, 244).astype(np.uint8)
            coef = rng.integers(-amp, amp + 1, (n, h, w)).astype(np.int16)
            coef[n // 2:, h // 2:, :] //= 64                            # some blocks with a quiet half
            lv, s, has = api.rdoq_batch(torch.from_numpy(coef).cuda(), depth, 0, 1, 0, 0, 0, qps, lam, ctx)
            lv, s = lv.cpu().numpy(), s.cpu().numpy()
            big = 0
            for b in range(n):
                want, ws = orc.rdoq(depth, coef[b], w, h, 0, 1, 0, 0, 0, qps, lam, ctx)
                assert np.array_equal(lv[b].ravel(), want), (w, h, qp, b)
                assert s[b] == ws
                big += int((np.abs(want.astype(np.int32)) >= 254).sum())
            if amp >= 9000:
                assert big > 100, (w, h, qp, big)


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
    assert hip.uvghip_rdoq_workspace_bytes(8, 8, 4) >= 8


def _qr_hip(api, depth, c, ref, pred, tus):
    """One golden / synthetic case through uvghip_quantize_residual_batch; planes and TU positions given."""
    import torch
    w, h, color = c["w"], c["h"], c["color"]
    idx = c["lfnst"]
    lf = idx if (c["cu_type"] == 1 and color == 0) else 0
    hor, ver, sw, sh = api.mts_select(w, h, color, c["cu_type"], 0, idx, 0, 0, 0)
    rec = torch.full_like(pred, 7 if depth == 8 else 0x0707)
    n = tus.shape[0]
    lt = api.make_lfnst_tus([[c["imode"], lf, w.bit_length() - 1, h.bit_length() - 1]] * n) if lf else None
    coeff, has = api.quantize_residual_batch(ref, pred, rec, tus, w, h, depth, color, hor, ver, sw, sh, c["qps"], bool(c["intra"]), c["cu_type"],
                                             bool(c["trskip"]), bool(c["rdoq"]), bool(c["rdoq_skip"]), 0, 0, idx, c["lam"], c["ctx"], lt)
    return coeff, has, rec


@pytest.mark.parametrize("depth", [8, 10])
def test_quantize_residual_branches_vs_reference(hip, depth):
    """RDOQ / LFNST / transform-skip branches of uvg_quantize_residual through the staged HIP path vs the vectors the
    reference produced (levels, has_coeffs and the reconstruction)."""
    from uvg266_amd import api
    from test_gpu_picture import dev
    from test_oracle_rdoq import qr_goldens
    seen = set()
    for c in qr_goldens(depth):
        coeff, has, rec = _qr_hip(api, depth, c, dev(c["ref"]), dev(c["pred"]), api.make_tus([[0, 0]]))
        tag = {k: v for k, v in c.items() if k in ("w", "h", "color", "qps", "intra", "trskip", "rdoq", "rdoq_skip", "lfnst", "imode")}
        assert int(has[0]) == c["has"] and np.array_equal(coeff.cpu().numpy().ravel(), c["q"]), tag
        assert np.array_equal(rec.cpu().numpy(), c["rec"]), tag
        seen.add((c["rdoq"], c["lfnst"] > 0, c["trskip"], c["has"]))
    assert len(seen) >= 7


@pytest.mark.parametrize("depth", [8, 10])
def test_quantize_residual_batches_vs_oracle(hip, orc, depth):
    """Many TUs per launch at arbitrary positions of a picture, every branch, vs the oracle composition."""
    import torch
    from uvg266_amd import api
    from test_gpu_picture import dev, rand_plane
    from test_oracle_rdoq import oracle_quantize_residual
    rng = np.random.default_rng(depth + 50)
    PH, PW = 160, 224
    sc = 1 << (depth - 8)
    ref = rand_plane(rng, PH, PW, depth)
    pred = np.clip(ref.astype(np.int32) + rng.integers(-20, 21, ref.shape) * sc, 0, (1 << depth) - 1).astype(ref.dtype)
    dref, dpred = dev(ref), dev(pred)
    cases = [dict(w=8, h=8, color=0, cu_type=1, intra=1, trskip=0, rdoq=1, rdoq_skip=0, lfnst=0, imode=0),
             dict(w=4, h=4, color=0, cu_type=1, intra=1, trskip=0, rdoq=1, rdoq_skip=1, lfnst=0, imode=0),
             dict(w=16, h=16, color=0, cu_type=1, intra=1, trskip=0, rdoq=1, rdoq_skip=0, lfnst=2, imode=34),
             dict(w=32, h=8, color=0, cu_type=1, intra=1, trskip=0, rdoq=0, rdoq_skip=0, lfnst=1, imode=50),
             dict(w=4, h=16, color=0, cu_type=2, intra=0, trskip=1, rdoq=0, rdoq_skip=0, lfnst=0, imode=0),
             dict(w=32, h=32, color=0, cu_type=2, intra=0, trskip=0, rdoq=1, rdoq_skip=0, lfnst=0, imode=0),
             dict(w=16, h=8, color=1, cu_type=1, intra=1, trskip=0, rdoq=1, rdoq_skip=0, lfnst=0, imode=0)]
    total_nz = 0
    for c in cases:
        w, h = c["w"], c["h"]
        qp = int(rng.integers(17, 30))
        c.update(qps=qp + 6 * (depth - 8), lam=0.57 * 2.0 ** ((qp - 12) / 3.0), ctx=rng.integers(0, 256, 244).astype(np.uint8))
        xy = np.array([[x, y] for y in range(0, PH - h + 1, h) for x in range(0, PW - w + 1, w)], np.int32)
        xy = xy[rng.permutation(len(xy))[:40]]
        coeff, has, rec = _qr_hip(api, depth, c, dref, dpred, api.make_tus(xy))
        coeff, has, rec = coeff.cpu().numpy(), has.cpu().numpy(), rec.cpu().numpy()
        nz = 0
        for i, (x, y) in enumerate(xy):
            cc = dict(c, ref=np.ascontiguousarray(ref[y:y + h, x:x + w]), pred=np.ascontiguousarray(pred[y:y + h, x:x + w]))
            whas, wq, wrec = oracle_quantize_residual(orc, depth, cc)
            assert has[i] == whas and np.array_equal(coeff[i].ravel(), wq), (c["w"], c["h"], c["rdoq"], c["lfnst"], c["trskip"], i)
            assert np.array_equal(rec[y:y + h, x:x + w], wrec[:h, :w])
            nz += whas
        total_nz += nz
    assert total_nz > 40
