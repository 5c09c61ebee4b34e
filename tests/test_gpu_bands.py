"""CTU-row sharding on one GPU: N emulated ranks (own buffers, rows they do not own poisoned, halos moved by explicit
copies) must produce the same pictures and statistics, bit for bit, as the whole-frame kernels; plus the RCCL entry
points driven with a one-rank communicator (send-to-self), and the compact ALF covariance format."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(frames, stage):
    import torch
    from uvg266_amd import pipeline
    st = torch.cuda.current_stream().cuda_stream
    for f in frames:
        pipeline.run(stage(f), st)
    torch.cuda.synchronize()


def _band_run(hip, wl_name, n, t=3):
    """-> (whole-frame BandFrame, [emulated rank BandFrames]) after a complete step."""
    import torch
    from uvg266_amd import api, bands, pipeline
    wl = pipeline.WORKLOADS[wl_name]
    modes = api.make_modes(pipeline.MODES)
    whole = pipeline.BandFrame(hip, wl, t, "cuda", modes)
    _run([whole], lambda f: f.all_launches())
    ranks = [pipeline.BandFrame(hip, wl, t, "cuda", modes, rank=r, nranks=n, poison=True) for r in range(n)]
    _run(ranks, lambda f: [l for c in f.chains for l in c])
    _run(ranks, lambda f: f.stage_a)
    bands.emulate([f.spec_dbk for f in ranks])
    _run(ranks, lambda f: f.stage_b)
    bands.emulate([f.spec_alf for f in ranks])
    _run(ranks, lambda f: f.stage_c)
    if wl["alf"]:
        total = sum(f.alf_sums for f in ranks)          # what ncclAllReduce leaves on every rank
        for f in ranks:
            f.alf_sums.copy_(total)
    bands.emulate([f.spec_gather for f in ranks])
    torch.cuda.synchronize()
    return whole, ranks


@pytest.mark.parametrize("wl_name,n", [("test8", 2), ("test8", 5), ("test10", 3), ("test10", 4)])
def test_emulated_bands_equal_whole_frame(hip, wl_name, n):
    import torch
    whole, ranks = _band_run(hip, wl_name, n)
    rects_all = whole.rects.cpu().numpy()
    for f in ranks:
        # the gathered final picture is the whole-frame picture on every rank
        for a, b in zip(f.final, whole.final):
            assert torch.equal(a, b)
        # owned rows of the intermediate pictures
        for sub, mine, ref in ((1, f.rec_y, whole.rec_y), (2, f.rec_u, whole.rec_u), (2, f.rec_v, whole.rec_v),
                               (1, f.sao_y, whole.sao_y), (2, f.sao_u, whole.sao_u), (2, f.sao_v, whole.sao_v)):
            r0, r1 = f.band.owned(sub)
            assert torch.equal(mine[r0:r1], ref[r0:r1])
        # rows nobody gave this rank are still poison: the halos are sufficient AND nothing else leaked in
        lo = f.band.y0 - (4 if f.band.up >= 0 else 0)
        assert lo <= 0 or bool((f.rec_y[:lo] == 0x55).all())
        # per-CTU results of the owned CTUs
        sel = np.nonzero((rects_all[:, 1] >= f.band.y0) & (rects_all[:, 1] < f.band.y1))[0]
        sel_t = torch.from_numpy(sel).cuda()
        for k in "yuv":
            assert torch.equal(f.params[k], whole.params[k][sel_t])
            assert torch.equal(f.edge[k], whole.edge[k][sel_t]) and torch.equal(f.bandst[k], whole.bandst[k][sel_t])
        y4 = slice(f.band.y0 // 4, (f.band.y1 + 3) // 4)
        assert torch.equal(f.alf_cls[y4], whole.alf_cls[y4])
        assert torch.equal(f.alf_present, whole.alf_present[sel_t])
        npres = [bin(int(m) & 0xffffffff).count("1") for m in f.alf_present.cpu().numpy()]
        for i, (j, k) in enumerate(zip(sel, npres)):
            assert torch.equal(f.alf_rec[i, :k], whole.alf_rec[j, :k])
        assert torch.equal(f.alf_sums, whole.alf_sums)
    # and the pictures really were filtered
    assert not torch.equal(whole.rec_y, whole.sao_y) and not torch.equal(whole.sao_y, whole.alf_y)


def test_whole_frame_plan_equals_the_frame_entry_points(hip):
    """nranks = 1: the band calls over [0, H) are uvghip_deblock_frame / uvghip_alf_classify_frame."""
    import torch
    from uvg266_amd import api, layout, pipeline
    wl = pipeline.WORKLOADS["test8"]
    f = pipeline.BandFrame(hip, wl, 1, "cuda", api.make_modes(pipeline.MODES))
    _run([f], lambda fr: [l for c in fr.chains for l in c])
    y, u, v = f.rec_y.clone(), f.rec_u.clone(), f.rec_v.clone()
    _run([f], lambda fr: fr.filter_launches())
    api.deblock_frame(y, u, v, f.scu, wl["W"], wl["H"], frame_qp=f.qp)
    assert torch.equal(y, f.rec_y) and torch.equal(u, f.rec_u) and torch.equal(v, f.rec_v)
    assert torch.equal(api.alf_classify_frame(f.sao_y, wl["W"], wl["H"]), f.alf_cls)


@pytest.mark.parametrize("depth", [8, 10])
def test_compact_covariances(hip, depth):
    """compact records -> expand == the full-layout kernel; reduce == sum over the rectangles."""
    import torch
    from uvg266_amd import api, layout, lib
    from test_gpu_alf import textured
    from test_gpu_picture import dev
    rng = np.random.default_rng(depth)
    W, Hh = 200, 136
    rec = dev(textured(rng, Hh, W, depth))
    org = dev(textured(rng, Hh, W, depth))
    cls = api.alf_classify_frame(rec, W, Hh)
    rects = api.make_rects(layout.ctu_rects(W, Hh))
    n = rects.shape[0]
    ee, yv, pa = api.alf_stats_batch(org, rec, rects, cls)
    L, st = hip, torch.cuda.current_stream().cuda_stream
    recs = torch.full((n, 25, 1484), -7, dtype=torch.int64, device="cuda")
    pres = torch.zeros(n, dtype=torch.int32, device="cuda")
    lib.check(L.uvghip_alf_stats_compact_batch(depth, org.data_ptr(), org.stride(0), rec.data_ptr(), rec.stride(0), W, Hh, 0,
                                               rects.data_ptr(), n, cls.data_ptr(), cls.stride(0), recs.data_ptr(), pres.data_ptr(), st), "compact")
    e2, y2, p2 = torch.empty_like(ee), torch.empty_like(yv), torch.empty_like(pa)
    lib.check(L.uvghip_alf_cov_expand(recs.data_ptr(), pres.data_ptr(), n, 0, e2.data_ptr(), y2.data_ptr(), p2.data_ptr(), st), "expand")
    assert torch.equal(e2, ee) and torch.equal(y2, yv) and torch.equal(p2, pa)
    # present mask = the classes that occur; untouched slots keep their fill
    m = pres.cpu().numpy().astype(np.int64) & 0xffffffff
    c_np = cls.cpu().numpy() & 31
    for i, (x, y, w, h) in enumerate(layout.ctu_rects(W, Hh)):
        want = 0
        for c in np.unique(c_np[y // 4:(y + h) // 4, x // 4:(x + w) // 4]):
            want |= 1 << int(c)
        assert int(m[i]) == want
        k = bin(want).count("1")
        assert bool((recs[i, k:] == -7).all())
    sums = torch.empty((25, 1509), dtype=torch.int64, device="cuda")
    lib.check(L.uvghip_alf_cov_reduce(recs.data_ptr(), pres.data_ptr(), n, 0, sums.data_ptr(), st), "reduce")
    tot_e, tot_y, tot_p = ee.sum(0), yv.to(torch.int64).sum(0), pa.sum(0)
    iu = [(k, l) for k in range(13) for l in range(k, 13)]
    tri = torch.stack([tot_e[:, k, l].reshape(25, 16) for k, l in iu], 1).reshape(25, 91 * 16)
    assert torch.equal(sums[:, :1456], tri)
    assert torch.equal(sums[:, 1456:1508], tot_y.reshape(25, 52)) and torch.equal(sums[:, 1508], tot_p)


def test_rccl_entry_points_with_one_rank(hip):
    """ncclCommInitRank with one rank on the test GPU, then the product's exchange (grouped ncclSend/ncclRecv to self)
    and all-reduce calls on a side stream -- the same C entry points the multi-GPU driver issues."""
    import torch
    from uvg266_amd import bands, lib
    tr = bands.RcclTransport(0, 1, lambda raw: raw)
    try:
        g = torch.Generator().manual_seed(3)
        a = torch.randint(0, 255, (64, 256), dtype=torch.uint8, generator=g).cuda()
        b = torch.zeros_like(a)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        tr.exchange([(0, [(a, 8, 16), (a, 40, 44)], [(b, 8, 16), (b, 40, 44)])], st.cuda_stream)
        s = torch.arange(25 * 1509, dtype=torch.int64).reshape(25, 1509).cuda()
        st.wait_stream(torch.cuda.current_stream())
        fn, args = tr.allreduce_args(s)
        lib.check(fn(*args, st.cuda_stream), "allreduce")
        st.synchronize()
        assert torch.equal(b[8:16], a[8:16]) and torch.equal(b[40:44], a[40:44])
        assert int(b[:8].sum()) == 0 and int(b[16:40].sum()) == 0 and int(b[44:].sum()) == 0
        assert torch.equal(s.cpu(), torch.arange(25 * 1509, dtype=torch.int64).reshape(25, 1509))
    finally:
        tr.close()


@pytest.mark.parametrize("wl_name", ["test8", "test10"])
def test_frame_group_equals_single_pictures(hip, wl_name):
    """pipeline.FrameGroup (F pictures stacked in a GroupArena: search, predict, transforms and the quantiser once per block
    shape over the group) leaves every picture exactly as the per-picture chains of pipeline.BandFrame do."""
    import torch
    from uvg266_amd import api, pipeline
    wl = pipeline.WORKLOADS[wl_name]
    modes = api.make_modes(pipeline.MODES)
    F = 3
    grp = pipeline.FrameGroup(hip, wl, 5, F, "cuda", modes, step=2)
    st = torch.cuda.current_stream().cuda_stream
    pipeline.run(grp.all_launches(), st)
    torch.cuda.synchronize()
    for f in range(F):
        one = pipeline.BandFrame(hip, wl, 5 + 2 * f, "cuda", modes)
        _run([one], lambda fr: fr.all_launches())
        got = grp.frames[f]
        for n in pipeline.SIZES:
            assert torch.equal(got.bufs[n]["best"], one.bufs[n]["best"]) and torch.equal(got.bufs[n]["cost"], one.bufs[n]["cost"]), n
            assert torch.equal(got.bufs[n]["pred"], one.bufs[n]["pred"]), n
            for color in ((0, 1, 2) if n >= 8 else (0,)):
                a, b = grp.pool.jobs[(n, color)], one.pool.jobs[(n, color)]
                assert torch.equal(a["lev"][f], b["lev"][0]) and torch.equal(a["has"][f], b["has"][0]), (n, color)
        for a, b in zip(got.final, one.final):
            assert torch.equal(a, b)
        n_ctu = one.n_ctu                                                 # SAO ran once over the group: picture f's rectangles
        assert torch.equal(grp.sao_edge[:, f * n_ctu:(f + 1) * n_ctu], one.edge_all)
        assert torch.equal(grp.sao_params[:, f * n_ctu:(f + 1) * n_ctu], one.params_all)
        if wl["alf"]:
            assert torch.equal(got.alf_sums, one.alf_sums)


def test_frame_group_of_bands_equals_whole_pictures(hip):
    """A FrameGroup of CTU-row bands (rank 1 of 2: the stacked block lists hold only the owned rows of every picture): what it
    decides and reconstructs before the in-loop filters equals the same rows of the whole-picture group."""
    import torch
    from uvg266_amd import api, pipeline
    wl = pipeline.WORKLOADS["test8"]
    modes = api.make_modes(pipeline.MODES)
    F = 2
    st = torch.cuda.current_stream().cuda_stream
    whole = pipeline.FrameGroup(hip, wl, 3, F, "cuda", modes)
    band = pipeline.FrameGroup(hip, wl, 3, F, "cuda", modes, rank=1, nranks=2)
    assert not band.sao and whole.sao
    for g in (whole, band):
        pipeline.run(g.searches() + g.before_filters(), st)
    torch.cuda.synchronize()
    y0, y1 = band.frames[0].band.y0, band.frames[0].band.y1
    assert 0 < y0 < y1 <= wl["H"]
    for f in range(F):
        a, b = whole.frames[f], band.frames[f]
        for n in pipeline.SIZES:
            own = a.own[n][:, 1]
            sel = torch.from_numpy(((own >= y0) & (own < y1))).cuda()
            assert int(sel.sum()) == b.tables[n][2] > 0
            assert torch.equal(a.bufs[n]["best"][sel], b.bufs[n]["best"]) and torch.equal(a.bufs[n]["cost"][sel], b.bufs[n]["cost"]), n
            assert torch.equal(whole.pool.jobs[(n, 0)]["lev"][f][sel], band.pool.jobs[(n, 0)]["lev"][f]), n
            assert torch.equal(a.bufs[n]["rec"][y0:y1], b.bufs[n]["rec"][y0:y1]), n
