"""uvghip_encode_slice_rows -- the arithmetic coder on the device: from the search's outputs and the SAO decisions in device memory
to the slice data (one substream per WPP row) -- against the .266 the real encoder wrote (tests/golden/ref_ctu*.npz: row_bytes /
row_crc, row_off; tools/refcheck/ctu_dump.c records every row's stream when its substream ends)."""
import zlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32",
                                  "ref_ctucrc_1920x1080_8_qp22", "ref_ctucrc_1920x1080_10_qp27",
                                  "ref_ctucrc_3840x2160_10_qp22"])
def test_slice_data_equals_the_encoders(hip, name):
    import torch
    from uvg266_amd import api
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cl.run()                                  # search -> filters -> slice data, one call
    out, nbytes = cl.slice_data()
    torch.cuda.synchronize()
    nb = nbytes.cpu().numpy()[0]
    assert (nb <= out.shape[2]).all()
    assert np.array_equal(np.concatenate([[0], np.cumsum(nb)]), g["row_off"])
    rows = [out[0, r, :nb[r]].cpu().numpy() for r in range(len(nb))]
    if "row_bytes" in g.files:
        data = np.concatenate(rows)
        assert np.array_equal(data, g["row_bytes"])
        assert g["bitstream"].tobytes().find(data.tobytes()) > 0          # ... and that is the slice data inside the encoder's .266
    else:
        assert np.array_equal(np.array([zlib.crc32(r.tobytes()) for r in rows], np.uint32), g["row_crc"])


def test_standalone_entry_point_and_sao_off(hip, orc):
    """uvghip_encode_slice_rows called on its own (not through the loop plan), with and without SAO syntax, against the oracle's
    row coder on the same hand-over."""
    import torch
    from uvg266_amd import api, layout
    W, Hh, depth, qp = 200, 136, 8, 32
    prm = H.search_params(W, Hh, qp)
    y, u, v = layout.synthetic_yuv420(W, Hh, 5, depth)
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cl.run_search()
    torch.cuda.synchronize()
    ry, ru, rv = (t.cpu().numpy() for t in cl.rec[0])
    res = H.search_result_from_device_layout(W, Hh, ry, ru, rv, cl.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP), cl.coeff[0].cpu().numpy(),
                                             cl.models[0].cpu().numpy().view(np.uint32))
    cl.run_filters()
    info, _ = cl.results()
    for sao in (True, False):
        out, nbytes = cl.encode_rows(sao=sao)
        torch.cuda.synchronize()
        nb = nbytes.cpu().numpy()[0]
        got = np.concatenate([out[0, r, :nb[r]].cpu().numpy() for r in range(len(nb))])
        if sao:
            want, off, _ = H.oracle_encode_rows(orc, depth, prm, res, info[0])
        else:
            want, off, _ = H.oracle_encode_rows_no_sao(orc, depth, prm, res)
        assert np.array_equal(np.concatenate([[0], np.cumsum(nb)]), off) and np.array_equal(got, want), sao


def test_sweep_whole_loop_equals_the_oracle_chain(hip, orc):
    """uvghip_loop_plan_run (search -> per-CTU deblocking -> SAO decisions -> final picture -> slice data) on 32 combinations of the
    sweep grid (sizes with 8-sample edge CTUs, both depths, QP 0..51, smooth / noisy / white-noise / impulse pictures) against the
    oracle's chain on the same source: final picture, SAO decisions, slice data byte for byte.  tools/refcheck/sweep_ctu.py holds the
    oracle's chain to the real encoder on the same grid."""
    import torch
    from uvg266_amd import api
    for W, Hh, depth, qp, t in H.sweep_cases(32, 31337):
        prm = H.search_params(W, Hh, qp)
        y, u, v = H.varied_picture(W, Hh, t, depth)
        cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
        cl.run()
        out, nbytes = cl.slice_data()
        info, models = cl.results()
        nb = nbytes.cpu().numpy()[0]
        got = np.concatenate([out[0, r, :nb[r]].cpu().numpy() for r in range(len(nb))])
        final = [p.cpu().numpy() for p in cl.out[0]]
        s = H.oracle_search_picture(orc, depth, prm, y, u, v)
        f = H.oracle_sao_picture(orc, depth, W, Hh, qp, prm.lam, (y, u, v), (s["rec_y"], s["rec_u"], s["rec_v"]), H.scu_from_cu(s["cu"], qp))
        want, off, _ = H.oracle_encode_rows(orc, depth, prm, s, f["sao"])
        case = (W, Hh, depth, qp, t)
        assert np.array_equal(H.sao_info_comparable(info[0]), H.sao_info_comparable(f["sao"])), case
        for a, k in zip(final, ("final_y", "final_u", "final_v")):
            assert np.array_equal(a, f[k]), (case, k)
        assert np.array_equal(np.concatenate([[0], np.cumsum(nb)]), off) and np.array_equal(got, want), case


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_264x136_10_qp32", "ref_ctucrc_1920x1080_8_qp22", "ref_ctucrc_3840x2160_10_qp22"])
def test_overlapped_run_equals_the_encoders(hip, name):
    """uvghip_loop_plan_run_overlapped: the filter stage BESIDE the search (persistent workgroups behind its per-CTU flags), the coder behind the
    filter stage's flags -- the same slice data, SAO decisions and output pictures as the three launches one after the other, several
    pictures in the group, run after run (the flags are reset in stream order)."""
    import torch
    from uvg266_amd import api
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    n = 3
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v)) for _ in range(n)]
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src)
    cl.run()
    torch.cuda.synchronize()
    want_nals = cl.group_nals()
    want_out = [[p.clone() for p in o] for o in cl.out]
    want_info = [t.clone() for t in cl.sao_device()]
    for rep in range(3):
        out, nbytes = cl.slice_data()
        out.zero_(); nbytes.zero_()
        for o in cl.out:
            for p in o:
                p.zero_()
        for i in range(n):          # nothing may come from the run before: the coder's rows start BEFORE the search has written their start models
            cl.models[i].fill_(0x5a5a5a5a); cl.coeff[i].fill_(77); cl.cu[i].zero_()
            for p in cl.rec[i]:
                p.zero_()
        cl.run_overlapped()
        torch.cuda.synchronize()
        assert cl.group_nals() == want_nals, f"NAL units, repetition {rep}"
        assert all(torch.equal(a, b) for o, w in zip(cl.out, want_out) for a, b in zip(o, w)), "output pictures"
        assert all(torch.equal(a, b) for a, b in zip(cl.sao_device(), want_info)), "SAO decisions / models"
    nb = nbytes.cpu().numpy()[0]
    rows = [out[0, r, :nb[r]].cpu().numpy() for r in range(len(nb))]
    if "row_bytes" in g.files:
        assert np.array_equal(np.concatenate(rows), g["row_bytes"])
    else:
        assert np.array_equal(np.array([zlib.crc32(r.tobytes()) for r in rows], np.uint32), g["row_crc"])
