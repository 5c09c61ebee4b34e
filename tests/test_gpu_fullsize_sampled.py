"""BASELINE.json configs[1] (1920x1080 8-bit) and configs[3] (3840x2160 10-bit --alf full) at FULL size against the
oracle: the whole plan of uvg266_amd.pipeline runs on the GPU, then >= 512 randomly chosen units of every kernel's output
are recomputed by the oracle from that kernel's inputs (whole pictures where the oracle is fast enough: deblocking, SAO,
ALF classification).  Index-width / stride / grid-size bugs that only show at 3840-wide 10-bit planes fail here; the
property tests in test_gpu_fullsize_2160p.py cannot see them."""
import ctypes

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
SAMPLES = 512


@pytest.fixture(scope="module", params=["1080p8", "2160p10alf"])
def full(request, hip):
    """One complete step of the workload (ALF switched on for 1080p too), with the pre-deblocking reconstruction kept."""
    import torch
    from uvg266_amd import api, pipeline
    wl = dict(pipeline.WORKLOADS[request.param], alf=True)
    fr = pipeline.BandFrame(hip, wl, 5, "cuda", api.make_modes(pipeline.MODES))
    st = torch.cuda.current_stream().cuda_stream
    pipeline.run([l for c in fr.chains for l in c], st)
    torch.cuda.synchronize()
    pre = [t.cpu().numpy().copy() for t in (fr.rec_y, fr.rec_u, fr.rec_v)]
    pipeline.run(fr.filter_launches(), st)
    torch.cuda.synchronize()
    return fr, wl, pre


def np_(t):
    return t.cpu().numpy()


def pick(rng, n, k=SAMPLES):
    return np.sort(rng.permutation(n)[: min(k, n)])


def test_search_predict_tu_sampled(full, orc):
    """Per block size: best mode + cost of the fused search, the prediction of the chosen mode, RDOQ levels, has_coeffs and
    the reconstruction -- luma and (N >= 8) both chroma planes."""
    from uvg266_amd import pipeline
    fr, wl, pre = full
    d, W, Hh, qp = wl["depth"], wl["W"], wl["H"], fr.qp
    y, u, v = fr.host
    ctx = pipeline.synthetic_rdoq_ctx()
    lam = pipeline.intra_lambda(qp)
    qps = qp + 6 * (d - 8)
    mx = (1 << d) - 1
    for n in pipeline.SIZES:
        rng = np.random.default_rng(n)
        blks, tus, cnt = fr.tables[n]
        b = fr.bufs[n]
        rows = np_(blks).view(np.int32).reshape(cnt, -1)[:, :4]
        best, cost, pred, rec = np_(b["best"]), np_(b["cost"]), np_(b["pred"]), np_(b["rec"])
        if n == 4:
            rec = pre[0]                 # the finest passes' reconstructions are the planes the filters then work on in place
        lev, has = np_(fr.pool.jobs[(n, 0)]["lev"][0]), np_(fr.pool.jobs[(n, 0)]["has"][0])
        sel = pick(rng, cnt)
        assert len(sel) == SAMPLES
        nz = 0
        for i in sel:
            x, yy, at, al = (int(t) for t in rows[i])
            o = np.ascontiguousarray(y[yy:yy + n, x:x + n]).ravel()
            costs, preds = orc.intra_mode_costs(d, y, W, Hh, x, yy, n, at, al, o, pipeline.MODES, True)
            j = int(np.argmin(costs))
            assert best[i] == pipeline.MODES[j] and cost[i] == costs[j], (n, x, yy)
            p = preds.reshape(len(pipeline.MODES), n, n)[j]
            assert np.array_equal(pred[yy:yy + n, x:x + n], p), (n, x, yy)
            res = (o.astype(np.int32).reshape(n, n) - p.astype(np.int32)).astype(np.int16).ravel()
            coef = orc.tr(d, d, False, 0, 0, n, n, 0, 0, res)
            q, _ = orc.rdoq(d, coef, n, n, 0, 1, 0, 0, 0, qps, lam, ctx)
            assert np.array_equal(lev[i].ravel(), q) and has[i] == int(q.any()), (n, x, yy)
            r = orc.tr(d, d, True, 0, 0, n, n, 0, 0, orc.dequant(d, q, n, n, d, qps, 0)).reshape(n, n)
            want = np.clip(r.astype(np.int32) + p.astype(np.int32), 0, mx)
            assert np.array_equal(rec[yy:yy + n, x:x + n], want), (n, x, yy)
            nz += int(q.any())
        assert nz > SAMPLES // 4
        if n < 8:
            continue
        c = n // 2
        crows = np_(b["cblks"]).view(np.int32).reshape(cnt, -1)[:, :4]
        for ci, (name, src) in enumerate((("u", u), ("v", v))):
            pr, rc = np_(b["pred_" + name]), (pre[1 + ci] if n == 8 else np_(b["rec_" + name]))
            lev, has = np_(fr.pool.jobs[(n, 1 + ci)]["lev"][0]), np_(fr.pool.jobs[(n, 1 + ci)]["has"][0])
            for i in sel[:: 2]:
                x, yy, at, al = (int(t) for t in crows[i])
                top, left = orc.intra_build_refs(d, src, W // 2, Hh // 2, x, yy, c, c, at, al)
                ftop, fleft = orc.intra_filter_refs(d, top, left, c, c)
                p = orc.intra_predict(d, int(best[i]), True, c, c, top, left, ftop, fleft).reshape(c, c)
                assert np.array_equal(pr[yy:yy + c, x:x + c], p), (n, name, x, yy)
                o = src[yy:yy + c, x:x + c]
                res = (o.astype(np.int32) - p.astype(np.int32)).astype(np.int16).ravel()
                coef = orc.tr(d, d, False, 0, 0, c, c, 0, 0, res)
                q, _ = orc.rdoq(d, coef, c, c, 1 + ci, 1, 0, 0, 0, qps, lam * 0.9, ctx)
                assert np.array_equal(lev[i].ravel(), q) and has[i] == int(q.any()), (n, name, x, yy)
                r = orc.tr(d, d, True, 0, 0, c, c, 0, 0, orc.dequant(d, q, c, c, d, qps, 0)).reshape(c, c)
                assert np.array_equal(rc[yy:yy + c, x:x + c], np.clip(r.astype(np.int32) + p.astype(np.int32), 0, mx)), (n, name, x, yy)


def test_deblocked_pictures_whole_frame(full, orc):
    """Every row of the three deblocked planes."""
    from uvg266_amd import layout
    fr, wl, pre = full
    d, W, Hh = wl["depth"], wl["W"], wl["H"]
    tab = layout.quadtree_scu_table(W, Hh, seed=5, qp=fr.qp)
    oy, ou, ov = (p.copy() for p in pre)
    orc.deblock_frame(d, oy, ou, ov, W, Hh, tab.view(np.uint8).reshape(tab.shape[0], -1), tab.shape[1], 0, 0, False, fr.qp, None)
    assert (oy != pre[0]).mean() > 0.01
    for got, want in ((fr.rec_y, oy), (fr.rec_u, ou), (fr.rec_v, ov)):
        assert np.array_equal(np_(got), want)


def test_sao_every_ctu(full, orc):
    """Statistics, chosen offsets and the applied picture: all CTUs of the three planes."""
    fr, wl, _ = full
    d, W, Hh = wl["depth"], wl["W"], wl["H"]
    for k, org, rec, out, rects, pw, ph in (("y", fr.host[0], fr.rec_y, fr.sao_y, fr.rects, W, Hh),
                                             ("u", fr.host[1], fr.rec_u, fr.sao_u, fr.crects, W // 2, Hh // 2),
                                             ("v", fr.host[2], fr.rec_v, fr.sao_v, fr.crects, W // 2, Hh // 2)):
        rec_np, rc = np_(rec), np_(rects).view(np.int32).reshape(-1, 4)
        assert len(rc) >= SAMPLES - 2
        we, wb = orc.sao_stats_rects(d, org, rec_np, [list(map(int, r)) for r in rc])
        assert np.array_equal(np_(fr.edge[k]), we) and np.array_equal(np_(fr.bandst[k]), wb), k
        n = len(rc)
        wp, wd = np.zeros((n, 8), np.int32), np.zeros(n, np.int32)
        orc.lib.orc_sao_edge_offsets(H.ptr(np.ascontiguousarray(we)), None, n, H.ptr(wp), H.ptr(wd))
        params = np_(fr.params[k])
        assert np.array_equal(params, wp), k
        want = rec_np.copy()
        for (x, y, w, h), (typ, eo, bp, *offs) in zip(rc, params):
            orc.sao_reconstruct_rect(d, rec_np, want, pw, ph, int(x), int(y), int(w), int(h), int(typ), int(eo), [int(bp)] * 2,
                                     [int(o) for o in offs] * 2, False)
        assert np.array_equal(np_(out), want), k
        assert (want != rec_np).any()


def test_alf_classes_stats_filter_sampled(full, orc):
    """Classification of the whole plane; covariance records (compact triangle format) of 512 CTUs x every class present;
    the frame sums; 7x7 luma / 5x5 chroma filter output of 512 CTUs."""
    import torch
    from uvg266_amd import lib
    fr, wl, _ = full
    d, W, Hh = wl["depth"], wl["W"], wl["H"]
    so = np_(fr.sao_y)
    ocls = orc.alf_classify_frame(d, so, W, Hh, d + 4)
    assert np.array_equal(np_(fr.alf_cls), ocls)
    rc = np_(fr.rects).view(np.int32).reshape(-1, 4)
    n = len(rc)
    rng = np.random.default_rng(77)
    sel = pick(rng, n)
    # expand the compact records through the library's own expander (its equality with the full-layout kernel is
    # tests/test_gpu_bands.py::test_compact_covariances), then compare with the oracle's full layout
    st = torch.cuda.current_stream().cuda_stream
    ee = torch.empty((n, 25, 13, 13, 4, 4), dtype=torch.int64, device="cuda")
    yv = torch.empty((n, 25, 13, 4), dtype=torch.int32, device="cuda")
    pa = torch.empty((n, 25), dtype=torch.int64, device="cuda")
    lib.check(fr.L.uvghip_alf_cov_expand(fr.alf_rec.data_ptr(), fr.alf_present.data_ptr(), n, 0, ee.data_ptr(), yv.data_ptr(), pa.data_ptr(), st), "expand")
    torch.cuda.synchronize()
    org = fr.host[0]
    for i in sel:
        x, y, w, h = (int(t) for t in rc[i])
        we, wy, wp = orc.alf_stats_rect(d, org, so, W, Hh, x, y, w, h, False, ocls)
        assert np.array_equal(np_(ee[i]), we) and np.array_equal(np_(yv[i]), wy) and np.array_equal(np_(pa[i]), wp), (x, y)
    # frame sums: pixel energy per class against numpy over the whole picture
    sums = np_(fr.alf_sums)
    err = (org.astype(np.int64) - so.astype(np.int64)) ** 2
    cls_px = np.repeat(np.repeat(ocls[: Hh // 4, : W // 4] & 31, 4, 0), 4, 1)
    for c in range(25):
        assert sums[c, 1508] == int(err[: cls_px.shape[0], : cls_px.shape[1]][cls_px == c].sum()), c
    # filters
    coef, clip = np_(fr.alf_coefs)[0], np_(fr.alf_clips)[0]
    got = np_(fr.alf_y)
    want = so.copy()
    for i in sel:
        x, y, w, h = (int(t) for t in rc[i])
        orc.alf_filter_rect(d, so, want, W, Hh, x, y, w, h, False, np.ascontiguousarray(coef), np.ascontiguousarray(clip), ocls)
        assert np.array_equal(got[y:y + h, x:x + w], want[y:y + h, x:x + w]), (x, y)
    crc = np_(fr.crects).view(np.int32).reshape(-1, 4)
    ccoef, cclip = np_(fr.alf_ccoefs)[0], np_(fr.alf_cclips)[0]
    for src, out in ((fr.sao_u, fr.alf_u), (fr.sao_v, fr.alf_v)):
        s, g = np_(src), np_(out)
        want = s.copy()
        for i in sel:
            x, y, w, h = (int(t) for t in crc[i])
            orc.alf_filter_rect(d, s, want, W // 2, Hh // 2, x, y, w, h, True, np.ascontiguousarray(ccoef), np.ascontiguousarray(cclip), None)
            assert np.array_equal(g[y:y + h, x:x + w], want[y:y + h, x:x + w]), (x, y)


@pytest.mark.parametrize("depth,count", [(8, 510 * 4), (10, 2040 * 4)])
def test_mts_and_lfnst_at_picture_batch_sizes(hip, orc, depth, count):
    """All 32x32 TUs of a picture in one launch through the explicit-MTS kernels (DST-7 / DCT-8 with the 16-column zero-out)
    and the LFNST kernels (forward + inverse, every intra mode class), 512 sampled blocks each vs the oracle."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(depth)
    x = rng.integers(-(1 << depth), 1 << depth, (count, 32, 32)).astype(np.int16)
    xd = torch.from_numpy(x).cuda()
    sel = pick(rng, count)
    combos = set()
    for tr_idx in (2, 3, 4, 5):                                  # MTS_DST7_DST7 .. MTS_DCT8_DCT8, cfg.mts = both
        hor, ver, sw, sh = api.mts_select(32, 32, 0, 1, 0, 0, 0, tr_idx, 3)
        assert hor in (1, 2) and ver in (1, 2) and (sw, sh) == (16, 16)
        combos.add((hor, ver))
        fwd = api.transform_batch(xd, depth, False, hor, ver, sw, sh)
        inv = api.transform_batch(fwd, depth, True, hor, ver, sw, sh)
        f, iv = np_(fwd), np_(inv)
        for i in sel[:: 2]:
            wf = orc.tr(depth, depth, False, hor, ver, 32, 32, sw, sh, np.ascontiguousarray(x[i]).ravel())
            assert np.array_equal(f[i].ravel(), wf), (hor, ver, i)
            assert np.array_equal(iv[i].ravel(), orc.tr(depth, depth, True, hor, ver, 32, 32, sw, sh, wf)), (hor, ver, i)
    assert len(combos) == 4
    # LFNST on 16x16 TUs (the 8x8 kernel, 16 outputs) and 4x4 TUs (the 4x4 kernel): in place, forward then inverse
    for n in (16, 4):
        cnt = count * (4 if n == 16 else 16)
        c = rng.integers(-2000, 2001, (cnt, n, n)).astype(np.int16)
        tus = np.zeros((cnt, 4), np.int8)
        tus[:, 0] = rng.integers(0, 67, cnt); tus[:, 1] = rng.integers(1, 3, cnt)
        tus[:, 2] = tus[:, 3] = n.bit_length() - 1
        cd, td = torch.from_numpy(c).cuda(), torch.from_numpy(tus).cuda()
        st = torch.cuda.current_stream().cuda_stream
        from uvg266_amd import lib
        lib.check(hip.uvghip_lfnst_batch(0, cd.data_ptr(), n, n, td.data_ptr(), cnt, st), "lfnst fwd")
        f = np_(cd).copy()
        lib.check(hip.uvghip_lfnst_batch(1, cd.data_ptr(), n, n, td.data_ptr(), cnt, st), "lfnst inv")
        iv = np_(cd)
        lw = n.bit_length() - 1
        for i in pick(rng, cnt):
            w = np.ascontiguousarray(c[i]).ravel().copy()
            orc.lib.orc_lfnst_fwd(H.ptr(w), n, n, int(tus[i, 0]), lw, lw, int(tus[i, 1]))
            assert np.array_equal(f[i].ravel(), w), (n, i)
            orc.lib.orc_lfnst_inv(H.ptr(w), n, n, int(tus[i, 0]), lw, lw, int(tus[i, 1]))
            assert np.array_equal(iv[i].ravel(), w), (n, i)
