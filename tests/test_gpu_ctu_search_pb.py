"""uvghip_ctu_search_pb on the GPU: the closed-loop CTU search of the P / B pictures of three low-delay encodes of the reference encoder
(tests/golden/ref_inter_*), every picture of a sequence in ONE call (their references are the encoder's own output pictures, so they do
not depend on each other and their wavefronts interleave) -- decisions, motion, reconstruction, levels and the three model sets of every
CTU equal the encoder's records; then the device coder turns the device search's hand-over into the encoder's slice data."""
import ctypes
import os
import numpy as np
import pytest
import helpers as H

pytestmark = pytest.mark.gpu
GOLDENS = ["ref_inter_136x72_10_qp22_4frames", "ref_inter_192x128_8_qp17_5frames", "ref_inter_264x136_8_qp32_9frames",
           # other tools than --preset medium's: P slices (no bi-prediction) without the temporal candidate; no fractional search, no early skip
           "ref_inter_136x72_8_qp27_4frames_p_notmvp", "ref_inter_192x128_10_qp24_4frames_subme0_noskip",
           "ref_inter_136x72_8_qp27_17frames_ra16", "ref_inter_136x72_10_qp22_17frames_ra16", "ref_inter_136x72_8_qp27_9frames_ra8", "ref_inter_136x200_8_qp27_11frames_owf1", "ref_inter_136x72_8_qp27_5frames_rd1", "ref_inter_136x72_8_qp27_33frames_ra16p16"]


def device_pictures(W, Hh, depth, pics, P, repeat=1):
    """-> (list of lib.CtuPbPicture, per-picture dict of the device tensors behind them, the records in the same order)"""
    import torch
    from uvg266_amd import lib
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    n4 = hc * 16 * wc * 16
    tdt = torch.uint8 if depth == 8 else torch.uint16
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    cache = {}

    def ref_dev(key, a):
        if key not in cache:
            cache[key] = dev(a)
        return cache[key]
    descs, tens, recs = [], [], []
    for rep in range(repeat):
        for fr, d, prm, F, keep in H.iter_inter_frames(W, Hh, P):
            if int(d["meta"][6]) == 2:
                continue
            q = lib.CtuPbPicture()
            cp = H.ctu_params(prm)
            assert ctypes.sizeof(cp) == ctypes.sizeof(q.params)
            ctypes.memmove(ctypes.byref(q.params), ctypes.byref(cp), ctypes.sizeof(cp))
            t = dict(src=[dev(p) for p in pics[fr]], rec=[torch.zeros((Hh >> c, W >> c), dtype=tdt, device="cuda") for c in (0, 1, 1)],
                     scu=torch.zeros(n4 * H.SCU_NP.itemsize, dtype=torch.uint8, device="cuda"), i4=torch.zeros(n4 * 8, dtype=torch.uint8, device="cuda"),
                     trees=torch.zeros(n4, dtype=torch.int32, device="cuda"), mot=torch.zeros(n4 * 8, dtype=torch.int32, device="cuda"),
                     co=torch.zeros(wc * hc * 6144, dtype=torch.int16, device="cuda"), mo=torch.zeros(wc * hc * 3 * 257, dtype=torch.int32, device="cuda"),
                     mi=torch.zeros(wc * hc * 3 * 18, dtype=torch.int32, device="cuda"), refs=[])
            c = q.pic
            c.src_y, c.src_u, c.src_v = (a.data_ptr() for a in t["src"])
            c.rec_y, c.rec_u, c.rec_v = (a.data_ptr() for a in t["rec"])
            c.src_stride = c.rec_stride = W
            c.src_stride_c = c.rec_stride_c = W // 2
            c.cu, c.cu_stride, c.coeff, c.models = t["scu"].data_ptr(), wc * 16, t["co"].data_ptr(), t["mo"].data_ptr()
            for f in ("slice_type", "poc", "n_refs", "tmvp", "max_merge", "merge_level", "frame_qp", "bipred", "fme_level", "early_skip", "depth_inter_min",
                      "depth_inter_max"):
                setattr(q, f, getattr(F, f))
            for i in range(16):
                q.ref_pocs[i], q.l[0][i], q.l[1][i] = F.ref_pocs[i], F.l[0][i], F.l[1][i]
            q.l_size[0], q.l_size[1] = F.l_size[0], F.l_size[1]
            q.ref_stride, q.ref_stride_c, q.ref_motion_stride = W, W // 2, wc * 16
            q.inflight_margin = 1 + F.owf_margin if F.owf else 0          # (a golden of an --owf run: the vectors restricted as the encoder's were)
            for i in range(F.n_refs):
                planes = [ref_dev((F.ref_pocs[i], k), keep[4 * i + k]) for k in range(3)]
                rm = ref_dev((F.ref_pocs[i], 3), keep[4 * i + 3])
                t["refs"] += planes + [rm]
                q.ref_y[i], q.ref_u[i], q.ref_v[i], q.ref_motion[i] = planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(), rm.data_ptr()
            q.inter4, q.models_inter, q.trees, q.motion_out = t["i4"].data_ptr(), t["mi"].data_ptr(), t["trees"].data_ptr(), t["mot"].data_ptr()
            descs.append(q); tens.append(t); recs.append((fr, d))
    return descs, tens, recs


def result_of(W, Hh, t):
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    n4 = hc * 16 * wc * 16
    return H.inter_result_from_device_layout(
        W, Hh, *(a.cpu().numpy() for a in t["rec"]), t["scu"].cpu().numpy().view(H.SCU_NP), t["i4"].cpu().numpy().view(H.INTER4_NP),
        t["trees"].cpu().numpy().view(np.uint32), t["mot"].cpu().numpy().reshape(n4, 8), t["co"].cpu().numpy(), t["mo"].cpu().numpy().view(np.uint32),
        t["mi"].cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("name", GOLDENS)
def test_pb_ctu_search_equals_the_encoders_records(hip, name):
    import torch
    from uvg266_amd import api
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, pics, P = H.inter_pictures_from_golden(g)
    descs, tens, recs = device_pictures(W, Hh, depth, pics, P, repeat=2)         # every picture twice: more wavefronts in flight, same answers
    ws = api.ctu_search_pb(descs, depth)
    torch.cuda.synchronize()
    assert len(descs) >= 6
    for t, (fr, d) in zip(tens, recs):
        assert H.compare_device_inter_picture(W, Hh, d, result_of(W, Hh, t)) == [], f"frame {fr}"


@pytest.mark.parametrize("waves", [2, 3, 4])
def test_every_wave_count_of_the_kernel_gives_the_same_pictures(hip, waves, monkeypatch):
    """The kernel is one source built for one to four waves per CTU (a wave count asks for the LDS image up to its own fields:
    pb_lds_bytes); a plain uvghip_ctu_search_pb launch runs the one-wave build, pictures in flight the four-wave one.  Every build, on
    the goldens with the most varied tools, through the development override the library reads at launch."""
    import torch
    from uvg266_amd import api
    monkeypatch.setenv("UVGHIP_PB_WAVES", str(waves))
    for name in GOLDENS[:1] + [n for n in GOLDENS if "owf1" in n or "rd1" in n or "ra16" in n][:3]:
        g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
        W, Hh, depth, pics, P = H.inter_pictures_from_golden(g)
        descs, tens, recs = device_pictures(W, Hh, depth, pics, P)
        ws = api.ctu_search_pb(descs, depth)
        torch.cuda.synchronize()
        for t, (fr, d) in zip(tens, recs):
            assert H.compare_device_inter_picture(W, Hh, d, result_of(W, Hh, t)) == [], (name, f"frame {fr}")


@pytest.mark.parametrize("name", GOLDENS)
def test_device_search_then_device_coder_gives_the_encoders_slice_data(hip, name):
    """Both halves on the device, nothing of the encoder's in between: uvghip_ctu_search_pb's hand-over (side information, second table,
    levels, models) goes straight into uvghip_encode_slice_rows_pb; with the encoder's SAO decisions the rows are the slice data of
    every P / B picture inside the encoder's .266, byte for byte."""
    import torch
    from uvg266_amd import api, lib
    L = lib.init(0)
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, pics, P = H.inter_pictures_from_golden(g)
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    ctus = wc * hc
    descs, tens, recs = device_pictures(W, Hh, depth, pics, P)
    ws = api.ctu_search_pb(descs, depth)
    torch.cuda.synchronize()
    stream = g["bitstream"].tobytes()
    gw, gh = (W + 7) // 8, (Hh + 7) // 8
    total = 0
    for q0, t, (fr, d) in zip(descs, tens, recs):
        sel = [k for k in range(len(g["meta"])) if int(g["meta"][k][0]) == fr]
        order = sorted(sel, key=lambda k: (int(g["meta"][k][2]), int(g["meta"][k][1])))
        dsao = torch.from_numpy(np.ascontiguousarray(g["sao"][order].reshape(1, ctus, 34))).cuda()
        dsaom = torch.from_numpy(np.ascontiguousarray(g["sao_models"][order].reshape(1, ctus, 6)).view(np.int16)).cuda()
        refm = t["refs"][4 * q0.l[0][0] + 3]                               # the collocated picture's motion: L0[0]
        col = refm.reshape(hc * 16, wc * 16, 8)[0:2 * gh:2, 0:2 * gw:2][:gh, :gw].contiguous()
        pic = (lib.CtuPicture * 1)()
        pic[0] = lib.CtuPicture(None, None, None, 0, 0, None, None, None, 0, 0, t["scu"].data_ptr(), wc * 16, 0, t["co"].data_ptr(), t["mo"].data_ptr())
        pb = (lib.SlicePb * 1)()
        q = pb[0]
        for f in ("slice_type", "poc", "n_refs", "tmvp", "max_merge", "merge_level", "frame_qp"):
            setattr(q, f, getattr(q0, f))
        for i in range(16):
            q.ref_pocs[i], q.l[0][i], q.l[1][i] = q0.ref_pocs[i], q0.l[0][i], q0.l[1][i]
        q.l_size[0], q.l_size[1] = q0.l_size[0], q0.l_size[1]
        q.col, q.inter4, q.models_inter = col.data_ptr(), t["i4"].data_ptr(), t["mi"].data_ptr()
        wsc = torch.empty(L.uvghip_slice_rows_pb_workspace_bytes(1), dtype=torch.uint8, device="cuda")
        cap = 3 * 64 * W * 2
        out = torch.zeros((hc, cap), dtype=torch.uint8, device="cuda")
        nb = torch.zeros(hc, dtype=torch.int32, device="cuda")
        lib.check(L.uvghip_encode_slice_rows_pb(depth, ctypes.byref(q0.params), pic, pb, 1, dsao.data_ptr(), dsaom.data_ptr(), wsc.data_ptr(), out.data_ptr(), cap,
                                                nb.data_ptr(), None), "uvghip_encode_slice_rows_pb")
        torch.cuda.synchronize()
        nbh, outh = nb.cpu().numpy(), out.cpu().numpy()
        off = g["row_off"][fr * hc:fr * hc + hc + 1]
        whole = b""
        for r in range(hc):
            want = g["row_bytes"][off[r]:off[r + 1]]
            assert nbh[r] == len(want) and np.array_equal(outh[r, :nbh[r]], want), (name, fr, r, int(nbh[r]), len(want))
            whole += outh[r, :nbh[r]].tobytes()
        assert stream.find(whole) > 0, (name, fr)
        total += len(whole)
    assert total > 100
