"""HIP quant group + fused TU round trip vs oracle and reference goldens (bit-exact)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from test_gpu_picture import Registry, dev, rand_plane
from test_gpu_dct import coef_like

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_quant_dequant_vs_reference_goldens(hip, depth):
    from uvg266_amd import api
    for name, arrs in H.read_golden("quant", depth):
        if name != "quant":
            continue
        (w, h, bd, qps, ts, intra, color, qp), coef, q, dq = arrs
        w, h = int(w), int(h)
        got_q = api.quant_batch(dev(coef.reshape(1, h, w)), int(bd), int(qps), bool(ts), bool(intra)).cpu().numpy().ravel()
        assert np.array_equal(got_q, q)
        got_d = api.dequant_batch(dev(q.reshape(1, h, w)), int(bd), int(qps), bool(ts)).cpu().numpy().ravel()
        assert np.array_equal(got_d, dq)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (4, 16), (32, 8), (16, 32)])
def test_quant_batch_vs_oracle(hip, orc, depth, shape):
    from uvg266_amd import api
    w, h = shape
    rng = np.random.default_rng(w * 7 + h + depth)
    n = 41
    for qps in (12, 22 + 6 * (depth - 8), 37, 51):
        for ts in (0, 1):
            for intra in (0, 1):
                x = coef_like(rng, (n, h, w), 2 if qps == 12 else 0, depth)
                x = (x.astype(np.int32) * (1 if qps < 30 else 6)).clip(-32768, 32767).astype(np.int16)
                got = api.quant_batch(dev(x), depth, qps, ts, intra).cpu().numpy()
                gd = api.dequant_batch(dev(got), depth, qps, ts).cpu().numpy()
                for b in (0, n // 2, n - 1):
                    want = orc.quant(depth, np.ascontiguousarray(x[b]).ravel(), w, h, depth, qps, ts, intra)
                    assert np.array_equal(got[b].ravel(), want)
                    assert np.array_equal(gd[b].ravel(), orc.dequant(depth, want, w, h, depth, qps, ts))
    # sums
    x = coef_like(rng, (n, h, w), 1, depth)
    s = api.coeff_abs_sum_batch(dev(x)).cpu().numpy()
    assert np.array_equal(s, np.abs(x.astype(np.int64)).reshape(n, -1).sum(1))
    wts = 0x0123_0456_0789_0abc
    f = api.fast_coeff_cost_batch(dev(x), wts).cpu().numpy()
    assert list(f) == [orc.fast_coeff_cost(depth, np.ascontiguousarray(x[b]).ravel(), w, h, wts) for b in range(n)]


@pytest.mark.parametrize("depth", [8, 10])
def test_tu_roundtrip_vs_reference_goldens(hip, depth):
    from uvg266_amd import api
    n = 0
    for name, arrs in H.read_golden("quant", depth):
        if name != "tu":
            continue
        (w, h, bd, qps, intra, S, has, color), ref, pred, q, rec = arrs
        w, h, S = int(w), int(h), int(S)
        dref, dpred = dev(ref.reshape(S, S)), dev(pred.reshape(S, S))
        drec = dpred.clone()
        coeff, got_has = api.tu_roundtrip_batch(dref, dpred, drec, api.make_tus([[0, 0]]), w, h, int(qps), bool(intra))
        assert int(got_has[0]) == has
        assert np.array_equal(coeff.cpu().numpy().ravel(), q)
        assert np.array_equal(drec.cpu().numpy()[:h, :w], rec.reshape(S, S)[:h, :w])
        n += 1
    assert n >= 20


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (16, 16), (32, 32), (16, 8), (4, 32), (32, 16)])
def test_tu_roundtrip_batch_vs_oracle(hip, orc, depth, shape):
    from uvg266_amd import api
    w, h = shape
    rng = np.random.default_rng(w + 3 * h + depth)
    Hh, W = 160, 224
    orig = rand_plane(rng, Hh, W, depth)
    noise = rng.integers(-20, 21, (Hh, W))
    pred = np.clip(orig.astype(np.int32) + noise * rng.integers(0, 2, (Hh, W)), 0, (1 << depth) - 1).astype(orig.dtype)
    xs, ys = np.meshgrid(np.arange(0, W - w + 1, w), np.arange(0, Hh - h + 1, h))
    xy = np.stack([xs.ravel(), ys.ravel()], 1)
    xy = xy[rng.permutation(len(xy))[: 53]]
    for th, tv, sw, sh in [(0, 0, 0, 0), (2, 2, 16 if w == 32 else 0, 16 if h == 32 else 0), (1, 2, 16 if w == 32 else 0, 16 if h == 32 else 0)]:
        for qps, intra in ((22 + 6 * (depth - 8), 1), (34, 0)):
            drec = dev(pred).clone()
            coeff, has = api.tu_roundtrip_batch(dev(orig), dev(pred), drec, api.make_tus(xy), w, h, qps, intra, th, tv, sw, sh)
            rec = drec.cpu().numpy(); coeff = coeff.cpu().numpy(); has = has.cpu().numpy()
            want_rec = pred.copy()
            for i, (x0, y0) in enumerate(xy):
                whas, wq, wrec = orc.tu_roundtrip(depth, depth, th, tv, sw, sh, w, h, qps, intra, orig, pred, W, int(x0), int(y0))
                assert whas == has[i] and np.array_equal(coeff[i].ravel(), wq), (i, th, tv)
                want_rec[y0:y0 + h, x0:x0 + w] = wrec[y0:y0 + h, x0:x0 + w]
            assert np.array_equal(rec, want_rec)


def test_state_free_strategy_pointers(hip, orc):
    reg = Registry(hip)
    assert hip.uvg_strategy_register_quant_hip(None, 8) == 1
    assert set(reg.table) == {"coeff_abs_sum", "fast_coeff_cost"}
    c = (np.arange(64 * 64) - 2048).astype(np.int16)              # tests/coeff_sum_tests.c:43-60
    f = ctypes.CFUNCTYPE(ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t)(reg.table["coeff_abs_sum"])
    assert f(H.ptr(c), c.size) == int(np.abs(c.astype(np.int64)).sum())
    g = ctypes.CFUNCTYPE(ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64)(reg.table["fast_coeff_cost"])
    c2 = np.random.default_rng(1).integers(-6, 7, 256).astype(np.int16)
    assert g(H.ptr(c2), 16, 16, 0x00400030_00200010) == orc.fast_coeff_cost(8, c2, 16, 16, 0x00400030_00200010)


def test_full_size_tu_properties(hip):
    """1080p luma as 8x8 TUs: recon error bounded by the quantiser step; zero residual -> no coeffs, rec == pred."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(11)
    Hh, W = 1080, 1920
    orig = rand_plane(rng, Hh, W, 8)
    pred = np.clip(orig.astype(np.int32) + rng.integers(-12, 13, (Hh, W)), 0, 255).astype(np.uint8)
    xs, ys = np.meshgrid(np.arange(0, W, 8), np.arange(0, Hh, 8))
    tus = api.make_tus(np.stack([xs.ravel(), ys.ravel()], 1))
    dorig, dpred = dev(orig), dev(pred)
    drec = dpred.clone()
    coeff, has = api.tu_roundtrip_batch(dorig, dpred, drec, tus, 8, 8, 22)
    err = (drec.int() - dorig.int()).abs()
    assert int(err.max()) <= 12 and float(err.float().mean()) < 2.5     # QP22: step ~ 8
    drec2 = dorig.clone()
    coeff2, has2 = api.tu_roundtrip_batch(dorig, dorig, drec2, tus, 8, 8, 22)
    assert int(has2.sum()) == 0 and int(coeff2.abs().sum()) == 0 and torch.equal(drec2, dorig)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (16, 16), (32, 32), (16, 8), (8, 32)])
@pytest.mark.parametrize("misalign", [0, 1])
def test_tu_halves_vs_oracle(hip, orc, depth, shape, misalign):
    """uvghip_tu_forward_batch -> quant -> dequant -> uvghip_tu_inverse_batch == the oracle's fused round trip: square blocks
    on the register / wave kernels (coefficient buffer 16-byte aligned), rectangles and a buffer off by one coefficient on
    the generic kernel; DCT-2 and the MTS kernels; the stored coefficients are the forward transform's int16 output."""
    import torch
    from uvg266_amd import api, lib
    L = lib.init(0)
    w, h = shape
    rng = np.random.default_rng(5 * w + 3 * h + depth + misalign)
    Hh, W = 160, 224
    orig = rand_plane(rng, Hh, W, depth)
    noise = rng.integers(-25, 26, (Hh, W))
    pred = np.clip(orig.astype(np.int32) + noise * rng.integers(0, 2, (Hh, W)), 0, (1 << depth) - 1).astype(orig.dtype)
    xs, ys = np.meshgrid(np.arange(0, W - w + 1, w), np.arange(0, Hh - h + 1, h))
    xy = np.stack([xs.ravel(), ys.ravel()], 1)
    xy = xy[rng.permutation(len(xy))[: 37]]
    n = len(xy)
    tus = api.make_tus(xy)
    dorig, dpred = dev(orig), dev(pred)
    st = orig.strides[0] // orig.itemsize
    P = lambda t: t.data_ptr()
    sk = lambda d: 16 if d == 32 else 0
    for th, tv in ((0, 0), (1, 2), (2, 1)):
        sw, sh = (sk(w), sk(h)) if (th or tv) else (0, 0)
        qps, intra = 22 + 6 * (depth - 8), 1
        store = torch.zeros(n * w * h + 8, dtype=torch.int16, device="cuda")
        coef = store[misalign: misalign + n * w * h].view(n, h, w)
        assert L.uvghip_tu_forward_batch(depth, th, tv, sw, sh, w, h, 0, P(dorig), st, P(dpred), st, P(tus), n, P(coef), None) == 0
        lev = api.quant_batch(coef.contiguous(), depth, qps, False, bool(intra))
        store2 = torch.zeros(n * w * h + 8, dtype=torch.int16, device="cuda")
        deq = store2[misalign: misalign + n * w * h].view(n, h, w)
        deq.copy_(api.dequant_batch(lev, depth, qps, False))
        drec = dpred.clone()
        assert L.uvghip_tu_inverse_batch(depth, th, tv, sw, sh, w, h, 0, P(deq), P(dpred), st, P(drec), st, P(tus), n, None) == 0
        if w == h and not misalign:      # the fused dequantise-on-load inverse gives the same picture
            drec2 = dpred.clone()
            assert L.uvghip_tu_dequant_inverse_batch(depth, th, tv, w, h, qps, P(lev), P(dpred), st, P(drec2), st, P(tus), n, None) == 0
            assert torch.equal(drec2, drec), (th, tv)
        elif w != h:
            assert L.uvghip_tu_dequant_inverse_batch(depth, th, tv, w, h, qps, P(lev), P(dpred), st, P(dpred.clone()), st, P(tus), n, None) != 0
        rec, lv, cf = drec.cpu().numpy(), lev.cpu().numpy(), coef.cpu().numpy()
        want_rec = pred.copy()
        for i, (x0, y0) in enumerate(xy):
            whas, wq, wrec = orc.tu_roundtrip(depth, depth, th, tv, sw, sh, w, h, qps, intra, orig, pred, W, int(x0), int(y0))
            assert np.array_equal(lv[i].ravel(), wq), (i, th, tv)
            want_rec[y0:y0 + h, x0:x0 + w] = wrec[y0:y0 + h, x0:x0 + w]
            res = np.ascontiguousarray((orig[y0:y0 + h, x0:x0 + w].astype(np.int32) - pred[y0:y0 + h, x0:x0 + w]).astype(np.int16))
            assert np.array_equal(cf[i].ravel(), orc.tr(depth, depth, 0, th, tv, w, h, sw, sh, res.ravel())), (i, th, tv, "coef")
        assert np.array_equal(rec, want_rec), (th, tv)
