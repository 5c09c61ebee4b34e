"""The configuration bench.py times -- two uvghip_loop_plan groups of many pictures in flight on two streams, more workgroups than
the device has slots, launches of the two groups sharing the device -- checked picture by picture, repeatedly.  (The one race the
CTU kernel had was timing dependent, tests/test_gpu_ctu_search.py::test_repeated_runs_...; contention is what the bench adds.)"""
import zlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _expected(orc, g, W, Hh, depth, qp, prm, pics):
    """Per distinct picture: (search CRCs without the reconstruction column, SAO info, SAO models, final planes, row lengths + CRCs)."""
    from concurrent.futures import ThreadPoolExecutor

    def one(k):
        yuv = pics[k]
        if k == 0:              # the reference encoder's own record
            res = dict(cu=g["cu"], trees=g["trees"], rec_y=g["rec_y"], rec_u=g["rec_u"], rec_v=g["rec_v"], coeff=g["coeff"], models=g["models"])
            rows = [g["row_bytes"][g["row_off"][r]:g["row_off"][r + 1]] for r in range(len(g["row_off"]) - 1)]
            return (H.ctu_crcs(res, W, Hh), g["sao"], g["sao_models"], [g["final_y"], g["final_u"], g["final_v"]],
                    [(len(r), zlib.crc32(r.tobytes())) for r in rows])
        o = H.oracle_search_picture(orc, depth, prm, *yuv)
        f = H.oracle_sao_picture(orc, depth, W, Hh, qp, prm.lam, yuv, (o["rec_y"], o["rec_u"], o["rec_v"]), H.scu_from_cu(o["cu"], qp))
        return (H.ctu_crcs(o, W, Hh), f["sao"], f["sao_models"], [f["final_y"], f["final_u"], f["final_v"]], None)
    with ThreadPoolExecutor(len(pics)) as ex:
        return list(ex.map(one, range(len(pics))))


def test_two_groups_of_64_pictures_in_flight_every_picture_checked(hip, orc):
    import torch
    from uvg266_amd import api, layout
    g = H.ctu_golden("ref_ctu_832x480_8_qp22")
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    distinct = [(y, u, v)] + [layout.synthetic_yuv420(W, Hh, t, depth) for t in (1, 2, 3)]
    want = _expected(orc, g, W, Hh, depth, qp, prm, distinct)
    n = 64                                  # 64 x 104 CTUs = 6656 workgroups per launch, two launches at once: 13 x the device's slots
    which = [[(i * 7 + gi) % 4 if i % 2 else 0 for i in range(n)] for gi in range(2)]
    dev = [[tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in distinct[k]) for k in w] for w in which]
    groups = [api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), d) for d in dev]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        for gi in (0, 1):                   # poison the outputs: a picture the launch skipped cannot pass on stale data
            for i in range(n):
                for t in groups[gi].out[i]:
                    t.fill_(7)
                groups[gi].coeff[i].fill_(-1)
        torch.cuda.synchronize()
        for gi in (0, 1):
            with torch.cuda.stream(streams[gi]):
                groups[gi].run()
        torch.cuda.synchronize()
        for gi, cl in enumerate(groups):
            info, models = cl.results()
            rows, nbytes = cl.slice_data()
            nb_all = nbytes.cpu().numpy()
            for i in range(n):
                crc, sao, sao_models, final, rowsum = want[which[gi][i]]
                where = (rep, gi, i)
                out = [t.cpu().numpy() for t in cl.out[i]]
                scu = cl.cu[i].cpu().numpy().reshape(-1).view(H.SCU_NP)
                res = H.search_result_from_device_layout(W, Hh, out[0], out[1], out[2], scu, cl.coeff[i].cpu().numpy(), cl.models[i].cpu().numpy().view(np.uint32))
                got = H.ctu_crcs(res, W, Hh)
                for col in (0, 2, 3):       # column 1 is the reconstruction before the filters, which the loop plan deblocks in place
                    assert np.array_equal(got[:, col], crc[:, col]), (where, col)
                assert np.array_equal(H.sao_info_comparable(info[i]), H.sao_info_comparable(sao)), where
                assert np.array_equal(models[i], sao_models), where
                for a, b in zip(out, final):
                    assert np.array_equal(a, b), where
                if rowsum is not None:
                    nb = nb_all[i]
                    assert [(int(nb[r]), zlib.crc32(rows[i, r, :nb[r]].cpu().numpy().tobytes())) for r in range(len(nb))] == rowsum, where
