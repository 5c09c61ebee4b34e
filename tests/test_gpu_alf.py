"""HIP ALF kernels vs reference goldens and the oracle (bit-exact)."""
import numpy as np
import pytest

import helpers as H
from test_gpu_picture import dev, rand_plane
from test_oracle_alf import alf_goldens

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_vs_reference_goldens(hip, depth):
    import torch
    from uvg266_amd import api, layout
    g = alf_goldens(depth)
    for (W, Hh, shift, cs), plane, cls, coef, clip, want in g["luma"]:
        W, Hh = int(W), int(Hh)
        d = dev(plane.reshape(Hh, W))
        gcls = api.alf_classify_frame(d, W, Hh, int(shift))
        assert np.array_equal(gcls.cpu().numpy(), cls.reshape(cs, cs)[: Hh // 4, : W // 4])
        rects = api.make_rects(layout.ctu_rects(W, Hh))
        out = torch.zeros_like(d)
        idx = torch.zeros(rects.shape[0], dtype=torch.int32, device="cuda")
        api.alf_filter_batch(d, out, rects, idx, dev(coef), dev(clip), gcls)
        assert np.array_equal(out.cpu().numpy().ravel(), want)
    for (CW, CH), plane, coef, clip, want in g["chroma"]:
        CW, CH = int(CW), int(CH)
        d = dev(plane.reshape(CH, CW))
        rects = api.make_rects(layout.ctu_rects(CW, CH, 32))
        out = torch.zeros_like(d)
        idx = torch.zeros(rects.shape[0], dtype=torch.int32, device="cuda")
        api.alf_filter_batch(d, out, rects, idx, dev(coef), dev(clip), None, is_chroma=True)
        assert np.array_equal(out.cpu().numpy().ravel(), want)
    (W, Hh, shift, cs), plane, cls, _, _, _ = g["luma"][0]
    (rx, ry, rw, rh), org, ee, yv, pa = g["stats"][0]
    W, Hh = int(W), int(Hh)
    gcls = dev(np.ascontiguousarray(cls.reshape(cs, cs)[: Hh // 4, : W // 4]))
    e, y, p = api.alf_stats_batch(dev(org.reshape(Hh, W)), dev(plane.reshape(Hh, W)), api.make_rects([[rx, ry, rw, rh]]), gcls)
    assert np.array_equal(e.cpu().numpy().ravel(), ee)
    assert np.array_equal(y.cpu().numpy().ravel(), yv) and np.array_equal(p.cpu().numpy().ravel(), pa)


def textured(rng, Hh, W, depth):
    sc = 1 << (depth - 8)
    yy, xx = np.mgrid[0:Hh, 0:W]
    rid = (xx // 16) + (yy // 16) * 9
    reg, amp, per = (rid * 7) % 6, (rid * 13) % 7, 2 + (rid * 5) % 7
    ph = np.select([reg == 0, reg == 1, reg == 2, reg == 3], [xx, yy, xx + yy, xx - yy + 400], 0)
    tri = np.where((ph % (2 * per)) < per, ph % per, per - (ph % per))
    v = 128 * sc + np.where(reg < 4, tri * amp * (1 + rid % 5), 0) * sc // 2 + rng.integers(-3, 4, (Hh, W)) * sc
    return np.clip(v, 0, (1 << depth) - 1).astype(H.px_dtype(depth))


@pytest.mark.parametrize("depth", [8, 10])
def test_frame_vs_oracle(hip, orc, depth):
    """A picture with partial CTUs: classification, luma + chroma filtering with per-CTU on/off and two filter
    sets, per-CTU statistics for luma and chroma."""
    import torch
    from uvg266_amd import api, layout
    rng = np.random.default_rng(depth)
    W, Hh = 200, 136
    rec = textured(rng, Hh, W, depth)
    org = np.clip(rec.astype(np.int32) + rng.integers(-4, 5, rec.shape) * (1 << (depth - 8)), 0, (1 << depth) - 1).astype(rec.dtype)
    d = dev(rec)
    cls = api.alf_classify_frame(d, W, Hh)
    ocls = orc.alf_classify_frame(depth, rec, W, Hh, depth + 4)
    assert np.array_equal(cls.cpu().numpy(), ocls)
    assert len(np.unique(ocls & 31)) >= 8
    rects = layout.ctu_rects(W, Hh)
    coef = rng.integers(-20, 21, (2, 25, 13)).astype(np.int16); coef[:, :, 12] = 0
    clipv = np.array([1 << depth] + [1 << (7 - 2 * i + depth - 8) for i in (1, 2, 3)], np.int16)
    clip = clipv[rng.integers(0, 4, (2, 25, 13))].astype(np.int16)
    idx = rng.integers(-1, 2, len(rects)).astype(np.int32)
    out = dev(rec).clone()
    api.alf_filter_batch(d, out, api.make_rects(rects), dev(idx), dev(coef), dev(clip), cls)
    want = rec.copy()
    for (x, y, w, h), si in zip(rects, idx):
        if si >= 0:
            orc.alf_filter_rect(depth, rec, want, W, Hh, x, y, w, h, False, np.ascontiguousarray(coef[si]), np.ascontiguousarray(clip[si]), ocls)
    assert np.array_equal(out.cpu().numpy(), want)
    # statistics, luma: every CTU
    e, yv, pa = api.alf_stats_batch(dev(org), d, api.make_rects(rects), cls)
    e, yv, pa = e.cpu().numpy(), yv.cpu().numpy(), pa.cpu().numpy()
    for i, (x, y, w, h) in enumerate(rects):
        we, wy, wp = orc.alf_stats_rect(depth, org, rec, W, Hh, x, y, w, h, False, ocls)
        assert np.array_equal(e[i], we) and np.array_equal(yv[i], wy) and np.array_equal(pa[i], wp), i
    # chroma-sized plane: 5x5 filter with two alternatives + statistics
    crec, corg = np.ascontiguousarray(rec[::2, ::2]), np.ascontiguousarray(org[::2, ::2])
    CW, CH = crec.shape[1], crec.shape[0]
    crects = layout.ctu_rects(CW, CH, 32)
    ccoef = rng.integers(-30, 31, (2, 7)).astype(np.int16); ccoef[:, 6] = 0
    cclip = clipv[rng.integers(0, 4, (2, 7))].astype(np.int16)
    cidx = rng.integers(-1, 2, len(crects)).astype(np.int32)
    cout = dev(crec).clone()
    api.alf_filter_batch(dev(crec), cout, api.make_rects(crects), dev(cidx), dev(ccoef), dev(cclip), None, is_chroma=True)
    cwant = crec.copy()
    for (x, y, w, h), si in zip(crects, cidx):
        if si >= 0:
            orc.alf_filter_rect(depth, crec, cwant, CW, CH, x, y, w, h, True, np.ascontiguousarray(ccoef[si]), np.ascontiguousarray(cclip[si]), None)
    assert np.array_equal(cout.cpu().numpy(), cwant)
    e, yv, pa = api.alf_stats_batch(dev(corg), dev(crec), api.make_rects(crects), None, is_chroma=True)
    for i, (x, y, w, h) in enumerate(crects):
        we, wy, wp = orc.alf_stats_rect(depth, corg, crec, CW, CH, x, y, w, h, True, None)
        assert np.array_equal(e[i].cpu().numpy(), we) and np.array_equal(yv[i].cpu().numpy(), wy) and np.array_equal(pa[i].cpu().numpy(), wp)


def test_full_size_properties(hip):
    """1080p: an all-zero filter is the identity; the statistics' pixel-error energy equals torch's, per CTU sums
    add up to the frame total; ee is symmetric."""
    import torch
    from uvg266_amd import api, layout
    rng = np.random.default_rng(2)
    W, Hh = 1920, 1080
    rec = dev(textured(rng, Hh, W, 8))
    org = dev(np.clip(rec.cpu().numpy().astype(np.int32) + rng.integers(-3, 4, (Hh, W)), 0, 255).astype(np.uint8))
    cls = api.alf_classify_frame(rec, W, Hh)
    rects = api.make_rects(layout.ctu_rects(W, Hh))
    zero = torch.zeros((1, 25, 13), dtype=torch.int16, device="cuda")
    clip = torch.full((1, 25, 13), 256, dtype=torch.int16, device="cuda")
    out = torch.zeros_like(rec)
    api.alf_filter_batch(rec, out, rects, torch.zeros(rects.shape[0], dtype=torch.int32, device="cuda"), zero, clip, cls)
    assert torch.equal(out, rec)
    e, y, p = api.alf_stats_batch(org, rec, rects, cls)
    assert int(p.sum()) == int(((org.int() - rec.int()) ** 2).sum())
    assert torch.equal(e, e.permute(0, 1, 3, 2, 5, 4))
    # center-tap row of y: sum_px cur * (org - rec), independent of class split
    assert int(y[:, :, 12, 0].sum()) == int((rec.long() * (org.long() - rec.long())).sum())


@pytest.mark.parametrize("depth", [8, 10])
def test_cc_alf_statistics_vs_reference(hip, depth):
    """uvghip_cc_alf_stats_batch against get_blk_stats_cc_alf of the reference (vectors of tools/refcheck/rc_alf.inc; the oracle is held to the
    same ones in tests/test_oracle_alf.py): every CTU of a 136 x 136 picture incl. the partial ones and the last row's boundary rule."""
    from uvg266_amd import api
    from test_oracle_alf import cc_ctus
    recs = alf_goldens(depth)["ccstats"]
    assert len(recs) == 2
    for (W, Hh, it), luma, rec_u, rec_v, org_u, org_v, ee, yv, pix in recs:
        W, Hh = int(W), int(Hh)
        CW, CH = W // 2, Hh // 2
        rects = api.make_rects(cc_ctus(W, Hh))
        dl = dev(luma.reshape(Hh, W))
        for c, (org, rec) in enumerate(((org_u, rec_u), (org_v, rec_v))):
            e, y, p = api.cc_alf_stats_batch(dev(org.reshape(CH, CW)), dev(rec.reshape(CH, CW)), dl, rects)
            e, y, p = e.cpu().numpy(), y.cpu().numpy(), p.cpu().numpy()
            for k in range(9):
                i = k * 2 + c
                assert np.array_equal(e[k].ravel(), ee[i * 49:(i + 1) * 49]) and np.array_equal(y[k], yv[i * 7:(i + 1) * 7]) and p[k] == pix[i], (int(it), k, c)
