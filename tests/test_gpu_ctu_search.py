"""uvghip_ctu_search_intra (the closed-loop intra search of whole pictures: split / mode RD decisions, reconstruction, levels, CABAC
model adaptation, WPP wavefront across workgroups) against records of the real reference encoder (tests/golden/ref_ctu*.npz):
item by item on two small pictures, per-CTU CRCs at 1080p and 2160p, several pictures in one launch."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def run_gpu(hip, depth, prm, pictures):
    import torch
    from uvg266_amd import api
    P = api.ctu_params(prm.pic_w, prm.pic_h, prm.qp, lam=prm.lam)
    P.depth_min, P.depth_max, P.combine_intra_cus = prm.depth_min, prm.depth_max, prm.combine_intra_cus
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in yuv) for yuv in pictures]
    cs = api.CtuSearch(P, src)
    cs.run()
    torch.cuda.synchronize()
    out = []
    for i in range(len(pictures)):
        ry, ru, rv = (t.cpu().numpy() for t in cs.rec[i])
        scu = cs.cu[i].cpu().numpy().reshape(-1).view(H.SCU_NP)
        out.append(H.search_result_from_device_layout(prm.pic_w, prm.pic_h, ry, ru, rv, scu, cs.coeff[i].cpu().numpy(),
                                                      cs.models[i].cpu().numpy().view(np.uint32)))
    return out


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32"])
def test_every_ctu_equals_the_reference_run(hip, name):
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    r = run_gpu(hip, depth, prm, [(y, u, v)])[0]
    wc = (W + 63) // 64
    for k in range(len(g["models"])):
        for j, what in enumerate(("at the CTU's start", "after the search", "after the coder")):
            assert np.array_equal(r["models"][k, j], g["models"][k, j]), (k % wc, k // wc, what)
    h4, w4 = Hh // 4, W // 4
    assert np.array_equal(r["cu"][:h4, :w4], g["cu"][:h4, :w4])
    assert np.array_equal(r["trees"][:h4, :w4], g["trees"][:h4, :w4])
    for p in ("rec_y", "rec_u", "rec_v"):
        assert np.array_equal(r[p], g[p]), p
    assert np.array_equal(H.ctu_crcs(r, W, Hh)[:, 2], H.ctu_crcs(dict(r, coeff=g["coeff"]), W, Hh)[:, 2])


@pytest.mark.parametrize("name", ["ref_ctucrc_1920x1080_8_qp22", "ref_ctucrc_1920x1080_10_qp27"])
def test_1080p_equals_the_reference_run_ctu_by_ctu(hip, name):
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    r = run_gpu(hip, depth, H.search_params(W, Hh, qp), [(y, u, v)])[0]
    bad = np.argwhere((H.ctu_crcs(r, W, Hh) != g["crc"]).any(axis=1)).ravel()
    assert bad.size == 0, bad[:10]


@pytest.mark.parametrize("name", ["ref_ctu_320x192_8_qp42", "ref_ctu_832x480_8_qp22"])
def test_repeated_runs_give_the_reference_result_every_time(hip, name):
    """The four waves of a workgroup run concurrently (the depth pipeline): whatever two of them can touch at the same moment must
    be private to each.  (Two depths once shared RDOQ's per-position cost arrays in the global scratch: at QP 42 about half of
    the runs of the 320x192 picture decided one CTU differently.)  Twelve runs in one process, each checked in full."""
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    for rep in range(12):
        r = run_gpu(hip, depth, prm, [(y, u, v)])[0]
        assert np.array_equal(r["models"], g["models"]), rep
        assert np.array_equal(r["coeff"], g["coeff"]), rep
        for p in ("rec_y", "rec_u", "rec_v"):
            assert np.array_equal(r[p], g[p]), (rep, p)


@pytest.mark.parametrize("dmin,dmax,combine", [(1, 3, 1), (1, 2, 1), (1, 1, 1), (2, 4, 0), (2, 2, 0), (3, 4, 0)])
def test_other_pu_depth_ranges_equal_the_oracle(hip, orc, dmin, dmax, combine):
    """--pu-depth-intra other than 1-4 (tests/test_ctu_emulation.py has the same cases on the host): leaves above the 4x4 depth on
    the depth waves' scratch; with depth_max < 3 the 64x64 candidate is built after the walk instead of beside it."""
    wins64 = 0
    cases = []
    for name in ("ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12"):
        W, Hh, depth, qp, y, u, v = H.golden_source(H.ctu_golden(name))
        cases.append((W, Hh, depth, qp, (y, u, v)))
    cases.append((200, 136, 10, 27, H.varied_picture(200, 136, 7, 10)))
    cases.append((128, 128, 8, 45, H.varied_picture(128, 128, 1011, 8)))
    for W, Hh, depth, qp, pic in cases:
        prm = H.search_params(W, Hh, qp)
        prm.depth_min, prm.depth_max, prm.combine_intra_cus = dmin, dmax, combine
        r = run_gpu(hip, depth, prm, [pic])[0]
        o = H.oracle_search_picture(orc, depth, prm, *pic)
        assert np.array_equal(H.ctu_crcs(r, W, Hh), H.ctu_crcs(o, W, Hh)), (W, Hh, depth, qp)
        assert np.array_equal(r["models"], o["models"]), (W, Hh, depth, qp)
        wins64 += int((o["cu"][:Hh // 4:16, :W // 4:16, 1] == 6).sum())
    assert wins64 > 0 or not combine


def test_sweep_of_small_pictures_equals_the_oracle(hip, orc):
    """48 pictures over sizes (incl. 8-sample CTUs at the right / bottom edge), both bit depths, QP 0..51 and five kinds of content
    (smooth, noisy, white noise, lone impulses): the device search against the oracle, which tools/refcheck/sweep_ctu.py holds to the
    real encoder on the same grid (1000 combinations at the time of writing)."""
    for W, Hh, depth, qp, t in H.sweep_cases(48, 2024):
        prm = H.search_params(W, Hh, qp)
        pic = H.varied_picture(W, Hh, t, depth)
        r = run_gpu(hip, depth, prm, [pic])[0]
        o = H.oracle_search_picture(orc, depth, prm, *pic)
        assert np.array_equal(H.ctu_crcs(r, W, Hh), H.ctu_crcs(o, W, Hh)), (W, Hh, depth, qp, t)
        assert np.array_equal(r["models"], o["models"]), (W, Hh, depth, qp, t)


def test_several_pictures_in_one_launch(hip, orc):
    """Pictures are independent (-p 1): a launch over four of them -- their wavefronts interleaved on the device -- gives each
    the result it gets alone (checked against the reference-run golden for the first, the oracle for the others)."""
    from uvg266_amd import layout
    g = H.ctu_golden("ref_ctu_832x480_8_qp22")
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    pics = [(y, u, v)] + [layout.synthetic_yuv420(W, Hh, t, depth) for t in (1, 2, 3)]
    rs = run_gpu(hip, depth, prm, pics)
    assert np.array_equal(rs[0]["rec_y"], g["rec_y"]) and np.array_equal(rs[0]["cu"][:Hh // 4, :W // 4], g["cu"][:Hh // 4, :W // 4])
    for i in (1, 2, 3):
        o = H.oracle_search_picture(orc, depth, prm, *pics[i])
        assert np.array_equal(H.ctu_crcs(rs[i], W, Hh), H.ctu_crcs(o, W, Hh)), i


def test_plan_is_reusable_and_the_one_shot_call_agrees(hip, orc):
    """uvghip_ctu_plan_run twice on the same plan (new source samples in the same buffers in between), then the one-shot
    uvghip_ctu_search_intra on the same buffers: each pass gives the oracle's result for the samples it saw."""
    import torch
    from uvg266_amd import api, layout
    W, Hh, depth, qp = 256, 128, 8, 32
    prm = H.search_params(W, Hh, qp)
    P = api.ctu_params(W, Hh, qp, lam=prm.lam)
    pics = [layout.synthetic_yuv420(W, Hh, t, depth) for t in (0, 5)]
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pics[0])]
    cs = api.CtuSearch(P, src)

    def result():
        torch.cuda.synchronize()
        ry, ru, rv = (t.cpu().numpy() for t in cs.rec[0])
        scu = cs.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP)
        return H.ctu_crcs(H.search_result_from_device_layout(W, Hh, ry, ru, rv, scu, cs.coeff[0].cpu().numpy(),
                                                            cs.models[0].cpu().numpy().view(np.uint32)), W, Hh)
    want = [H.ctu_crcs(H.oracle_search_picture(orc, depth, prm, *p), W, Hh) for p in pics]
    cs.run()
    assert np.array_equal(result(), want[0])
    for d, s in zip(src[0], pics[1]):
        d.copy_(torch.from_numpy(np.ascontiguousarray(s)))
    cs.run()
    assert np.array_equal(result(), want[1])
    for t in cs.rec[0]:
        t.zero_()
    cs.run_oneshot()
    assert np.array_equal(result(), want[1])


def test_plan_refuses_what_the_search_does_not_implement(hip):
    """uvghip_ctu_plan_create / uvghip_loop_plan_create fail loudly (an error code and a message, never a different result) on
    configurations outside --preset medium -p 1 and on malformed descriptors."""
    import ctypes
    import torch
    from uvg266_amd import api, lib, layout
    L = lib.init(0)
    W, Hh = 128, 64
    y, u, v = layout.synthetic_yuv420(W, Hh, 0, 8)
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))]
    good = api.CtuSearch(api.ctu_params(W, Hh, 27), src)

    def create(params, pics=None, n=1, ws=None):
        plan = ctypes.c_void_p()
        return L.uvghip_ctu_plan_create(8, ctypes.byref(params), good.pics if pics is None else pics, n, api._dev(good.ws) if ws is None else ws, ctypes.byref(plan))
    for change in (dict(wpp=0), dict(depth_max=5), dict(depth_min=3, depth_max=2), dict(depth_min=2), dict(rough_levels=1), dict(qp=64), dict(pic_w=W + 4), dict(lambda_=0.0)):
        p = api.ctu_params(W, Hh, 27)
        for k, val in change.items():
            setattr(p, k, val)
        assert create(p) != 0, change
    assert create(api.ctu_params(W, Hh, 27), n=0) != 0
    assert create(api.ctu_params(W, Hh, 27), ws=None if False else 0) != 0
    bad = (lib.CtuPicture * 1)()
    ctypes.memmove(bad, good.pics, ctypes.sizeof(bad))
    bad[0].coeff = None
    assert create(api.ctu_params(W, Hh, 27), pics=bad) != 0
    for field in ("src_stride", "rec_stride", "src_stride_c", "rec_stride_c"):        # strides smaller than the picture (or negative) are refused
        for val in (W // 4, -W):
            ctypes.memmove(bad, good.pics, ctypes.sizeof(bad))
            setattr(bad[0], field, val)
            assert create(api.ctu_params(W, Hh, 27), pics=bad) != 0, field
    assert L.uvghip_ctu_plan_run(None, None) != 0
    assert L.uvghip_ctu_search_workspace_bytes(0, W, Hh) == 0
    # 12-bit samples are not a build of this library
    plan = ctypes.c_void_p()
    assert L.uvghip_ctu_plan_create(12, ctypes.byref(api.ctu_params(W, Hh, 27)), good.pics, 1, api._dev(good.ws), ctypes.byref(plan)) != 0
