"""uvghip_encode_slice_rows_pb: the arithmetic coder on the device for P / B pictures -- skip flag, prediction mode, merge flag / index,
inter direction, reference indices, motion vector differences against the AMVP predictor (derived on the device from the picture's side
information and the row's history table), predictor index, root cbf, the inter transform tree, intra CUs as in an I slice, SAO syntax,
the slice type's context initialisation -- from the encoder's own decisions (tests/golden/ref_inter_*.npz: side information, motion,
levels, SAO parameters) to the slice data of its .266: every WPP row of every P / B picture, byte for byte, and found inside the
encoder's bitstream."""
import ctypes

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def models_u32(m, n):
    """[.., n * 5 (+1)] bytes (state0 u16 x n, state1 u16 x n, rate u8 x n) -> uint32 state0 | state1 << 16 [.., n]"""
    s0 = np.ascontiguousarray(m[..., :2 * n]).view(np.uint16).astype(np.uint32)
    s1 = np.ascontiguousarray(m[..., 2 * n:4 * n]).view(np.uint16).astype(np.uint32)
    return np.ascontiguousarray(s0 | (s1 << 16))


@pytest.mark.parametrize("name", ["ref_inter_192x128_8_qp17_5frames", "ref_inter_136x72_10_qp22_4frames", "ref_inter_264x136_8_qp32_9frames"])
def test_slice_data_of_p_and_b_pictures_equals_the_encoders(hip, name):
    import torch
    from uvg266_amd import api, lib
    L = lib.init(0)
    g = H.ctu_golden(name)
    W, Hh, depth, pics, P = H.inter_pictures_from_golden(g)
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    ctus = wc * hc
    by_poc, seen = {}, dict(pictures=0, rows=0, bytes=0)
    stream = g["bitstream"].tobytes()
    for fr in sorted(P):
        d = P[fr]
        refs = d["refs"]
        n_refs, pocs = int(refs[0]), [int(a) for a in refs[1:17]]
        lsz, lists = [int(refs[17]), int(refs[18])], [[int(a) for a in refs[19:35]], [int(a) for a in refs[35:51]]]
        poc, slice_type = int(refs[51]), int(d["meta"][6])
        d["ref_cu"] = H.ref_cu_table(d["cu"], d["motion"], ([pocs[lists[0][i]] for i in range(lsz[0])], [pocs[lists[1][i]] for i in range(lsz[1])]))
        by_poc[poc] = d
        if slice_type == 2:
            continue
        # ---- the device's inputs from the encoder's records ----
        scu = H.inter_scu_table(g, fr)
        cu, mot = d["cu"], d["motion"]
        intra = cu[:, :, 0] == 1
        scu["mv"][:, :, 0, 0][intra] = cu[:, :, 6][intra].astype(np.int32) | (cu[:, :, 7][intra].astype(np.int32) << 8)
        i4 = np.zeros((hc * 16, wc * 16, 8), np.uint8)
        fl = mot[:, :, 7]
        i4[:, :, 0], i4[:, :, 1], i4[:, :, 2] = fl & 1, (fl >> 1) & 1, (fl >> 2) & 7
        i4[:, :, 3], i4[:, :, 4], i4[:, :, 5] = (fl >> 14) & 1, (fl >> 8) & 7, (fl >> 11) & 7
        i4[:, :, 6], i4[:, :, 7] = mot[:, :, 4], mot[:, :, 5]
        gw, gh = (W + 7) // 8, (Hh + 7) // 8
        col = np.ascontiguousarray(by_poc[pocs[lists[0][0]]]["ref_cu"][0:2 * gh:2, 0:2 * gw:2][:gh, :gw]).reshape(-1)
        sel = [k for k in range(len(g["meta"])) if int(g["meta"][k][0]) == fr]
        order = sorted(sel, key=lambda k: (int(g["meta"][k][2]), int(g["meta"][k][1])))
        m257 = models_u32(g["models"][order], 257).reshape(ctus, 3, 257)
        m18 = models_u32(g["models_inter"][order], 18).reshape(ctus, 3, 18)
        sao = np.ascontiguousarray(g["sao"][order].reshape(1, ctus, 34))
        saom = np.ascontiguousarray(g["sao_models"][order].reshape(1, ctus, 6))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        dscu, di4, dcol, dco, dm, dmi, dsao, dsaom = t(scu.view(np.uint8)), t(i4), t(col), t(d["coeff"]), t(m257.view(np.int32)), t(m18.view(np.int32)), t(sao), t(saom.view(np.int16))
        pic = (lib.CtuPicture * 1)()
        pic[0] = lib.CtuPicture(None, None, None, 0, 0, None, None, None, 0, 0, dscu.data_ptr(), wc * 16, 0, dco.data_ptr(), dm.data_ptr())
        pb = (lib.SlicePb * 1)()
        q = pb[0]
        q.slice_type, q.poc, q.n_refs, q.tmvp, q.max_merge, q.merge_level, q.frame_qp = slice_type, poc, n_refs, 1, 6, 2, int(d["meta"][7])
        for i in range(16):
            q.ref_pocs[i] = pocs[i]
            q.l[0][i], q.l[1][i] = lists[0][i], lists[1][i]
        q.l_size[0], q.l_size[1] = lsz
        q.col, q.inter4, q.models_inter = dcol.data_ptr(), di4.data_ptr(), dmi.data_ptr()
        prm = api.ctu_params(W, Hh, int(d["meta"][3]))
        ws = torch.empty(L.uvghip_slice_rows_pb_workspace_bytes(1), dtype=torch.uint8, device="cuda")
        cap = 3 * 64 * W * 2
        out = torch.zeros((hc, cap), dtype=torch.uint8, device="cuda")
        nb = torch.zeros(hc, dtype=torch.int32, device="cuda")
        lib.check(L.uvghip_encode_slice_rows_pb(depth, ctypes.byref(prm), pic, pb, 1, dsao.data_ptr(), dsaom.data_ptr(), ws.data_ptr(), out.data_ptr(), cap,
                                                nb.data_ptr(), None), "uvghip_encode_slice_rows_pb")
        torch.cuda.synchronize()
        nb, out = nb.cpu().numpy(), out.cpu().numpy()
        off = g["row_off"][fr * hc:fr * hc + hc + 1]
        whole = b""
        for r in range(hc):
            want = g["row_bytes"][off[r]:off[r + 1]]
            got = out[r, :nb[r]]
            if not (nb[r] == len(want) and np.array_equal(got, want)):
                first = int(np.argmax(got[:min(len(got), len(want))] != want[:min(len(got), len(want))])) if len(got) and len(want) else 0
                assert False, (name, fr, r, int(nb[r]), len(want), "first differing byte", first)
            whole += got.tobytes()
            seen["rows"] += 1
            seen["bytes"] += int(nb[r])
        assert stream.find(whole) > 0, (name, fr)          # ... and that is the picture's slice data inside the encoder's .266
        seen["pictures"] += 1
    assert seen["pictures"] >= 3 and seen["bytes"] > 300, seen
