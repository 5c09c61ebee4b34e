"""The in-loop filter side of the closed loop on the device -- uvghip_deblock_frame_sao_snapshot (every CTU deblocked by its own
edges only: what the reference's SAO decision reads), uvghip_sao_stats_batch on it, uvghip_sao_decide_pictures (edge / band / merge
decision with the coder's SAO models carried CTU to CTU), uvghip_deblock_frame + uvghip_sao_apply_batch -- against records of
the real encoder (tests/golden/ref_ctu*.npz): the block every decision saw, the decisions, the models, the final picture."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def filters_on_device(W, Hh, depth, qp, lam, src, rec, scu_np, n_copies=1):
    """-> dict like helpers.oracle_sao_picture for picture 0 of n_copies identical pictures decided in one call."""
    import torch
    from uvg266_amd import api, layout
    dev = "cuda"
    S = [torch.from_numpy(np.ascontiguousarray(p)).to(dev) for p in src]
    R = [torch.from_numpy(np.ascontiguousarray(p)).to(dev) for p in rec]
    scu = torch.from_numpy(np.ascontiguousarray(scu_np).view(np.uint8).reshape(scu_np.shape[0], -1)).to(dev)
    snap = [r.clone() for r in R]
    api.deblock_frame(snap[0], snap[1], snap[2], scu, W, Hh, frame_qp=qp, sao_snapshot=True)
    rects = [torch.from_numpy(layout.ctu_rects(W >> c, Hh >> c, 64 >> c)).to(dev) for c in (0, 1)]
    stats = []
    for c in range(3):
        e, b = api.sao_stats_batch(S[c], snap[c], rects[0 if c == 0 else 1])
        stats.append((e.repeat(n_copies, 1, 1, 1).contiguous(), b.repeat(n_copies, 1, 1).contiguous()))
    info, models, params = api.sao_decide_pictures(n_copies, W, Hh, depth, qp, lam, stats)
    ctus = rects[0].shape[0]
    api.deblock_frame(R[0], R[1], R[2], scu, W, Hh, frame_qp=qp)
    out = [torch.zeros_like(r) for r in R]
    for c in range(3):
        api.sao_apply_batch(R[c], out[c], rects[0 if c == 0 else 1], params[c][:ctus].contiguous())
    torch.cuda.synchronize()
    inf = info.cpu().numpy().reshape(n_copies, ctus, 2, 17)
    mod = models.cpu().numpy().view(np.uint16).reshape(n_copies, ctus, 6)
    for k in range(1, n_copies):
        assert np.array_equal(inf[k], inf[0]) and np.array_equal(mod[k], mod[0])
    names = ("y", "u", "v")
    d = dict(sao=inf[0], sao_models=mod[0])
    for c in range(3):
        d["snap_" + names[c]] = snap[c].cpu().numpy()
        d["final_" + names[c]] = out[c].cpu().numpy()
    return d


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32"])
def test_filters_and_sao_decisions_equal_the_encoder_run(hip, name):
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    r = filters_on_device(W, Hh, depth, qp, float(g["lam"][0]), (y, u, v), (g["rec_y"], g["rec_u"], g["rec_v"]), H.scu_from_cu(g["cu"], qp), n_copies=3)
    for k in ("snap_y", "snap_u", "snap_v"):
        assert np.array_equal(r[k], g[k]), k
    assert np.array_equal(H.sao_info_comparable(r["sao"]), H.sao_info_comparable(g["sao"]))
    assert np.array_equal(r["sao_models"], g["sao_models"])
    for k in ("final_y", "final_u", "final_v"):
        assert np.array_equal(r[k], g[k]), k


@pytest.mark.parametrize("name", ["ref_ctucrc_1920x1080_8_qp22", "ref_ctucrc_1920x1080_10_qp27", "ref_ctucrc_3840x2160_10_qp22"])
def test_search_then_filters_equal_the_encoder_run(hip, name):
    """End to end on the device: the CTU search's own reconstruction and side information feed the filters; the picture that comes
    out is the picture the encoder returned."""
    import torch
    from uvg266_amd import api
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    cs = api.CtuSearch(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cs.run()
    torch.cuda.synchronize()
    rec = [t.cpu().numpy() for t in cs.rec[0]]
    scu = cs.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP).reshape(cs.cu[0].shape[0], -1)
    r = filters_on_device(W, Hh, depth, qp, float(g["lam"][0]), (y, u, v), rec, scu)
    assert np.array_equal(H.sao_info_comparable(r["sao"]), H.sao_info_comparable(g["sao"]))
    assert np.array_equal(r["sao_models"], g["sao_models"])
    assert np.array_equal(H.filter_crcs(r, W, Hh), g["filter_crc"])


def test_loop_plan_is_the_same_chain_in_one_call(hip):
    """uvghip_loop_plan_run (search + filters strung together in C) on two pictures: the encoder's returned picture and its SAO
    decisions for the golden one, and the same for a second picture as the step-by-step calls give."""
    import torch
    from uvg266_amd import api, layout
    g = H.ctu_golden("ref_ctu_832x480_8_qp22")
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    pics = [(y, u, v), layout.synthetic_yuv420(W, Hh, 7, depth)]
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in yuv) for yuv in pics]
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src)
    cl.run()
    info, models = cl.results()
    out = [[t.cpu().numpy() for t in o] for o in cl.out]
    for k, n in enumerate(("final_y", "final_u", "final_v")):
        assert np.array_equal(out[0][k], g[n]), n
    assert np.array_equal(H.sao_info_comparable(info[0]), H.sao_info_comparable(g["sao"])) and np.array_equal(models[0], g["sao_models"])
    # second picture: search alone, then the filters step by step
    cs = api.CtuSearch(api.ctu_params(W, Hh, qp, lam=prm.lam), [src[1]])
    cs.run()
    torch.cuda.synchronize()
    scu = cs.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP).reshape(cs.cu[0].shape[0], -1)
    r = filters_on_device(W, Hh, depth, qp, prm.lam, pics[1], [t.cpu().numpy() for t in cs.rec[0]], scu)
    for k, n in enumerate(("final_y", "final_u", "final_v")):
        assert np.array_equal(out[1][k], r[n]), n
    assert np.array_equal(info[1], r["sao"]) and np.array_equal(models[1], r["sao_models"])


@pytest.mark.parametrize("W,Hh,depth,qp", [(8, 8, 8, 27), (64, 64, 10, 22), (72, 40, 8, 12), (136, 200, 10, 32), (328, 264, 8, 47), (192, 64, 8, 37),
                                           (64, 192, 10, 42)])
def test_ragged_sizes_and_qp_range_against_the_oracle(hip, orc, W, Hh, depth, qp):
    """Pictures that are one CTU, a fraction of one, one row / one column of CTUs, with partial CTUs on both edges, at both bit
    depths and across the QP range: uvghip_loop_plan_run (search + filters) against the oracle's search + filters -- every
    search output per CTU, every SAO decision, the SAO models and the final picture."""
    import torch
    from uvg266_amd import api, layout
    prm = H.search_params(W, Hh, qp)
    y, u, v = layout.synthetic_yuv420(W, Hh, 11, depth)
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cl.run_search()
    torch.cuda.synchronize()
    ry, ru, rv = (t.cpu().numpy() for t in cl.rec[0])          # before the filters run in place
    scu = cl.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP)
    got = H.search_result_from_device_layout(W, Hh, ry, ru, rv, scu, cl.coeff[0].cpu().numpy(), cl.models[0].cpu().numpy().view(np.uint32))
    want = H.oracle_search_picture(orc, depth, prm, y, u, v)
    assert np.array_equal(H.ctu_crcs(got, W, Hh), H.ctu_crcs(want, W, Hh))
    cl.run_filters()
    info, models = cl.results()
    o = H.oracle_sao_picture(orc, depth, W, Hh, qp, prm.lam, (y, u, v), (want["rec_y"], want["rec_u"], want["rec_v"]), H.scu_from_cu(want["cu"], qp))
    assert np.array_equal(H.sao_info_comparable(info[0]), H.sao_info_comparable(o["sao"])) and np.array_equal(models[0], o["sao_models"])
    for k, n in enumerate(("final_y", "final_u", "final_v")):
        assert np.array_equal(cl.out[0][k].cpu().numpy(), o[n]), n
