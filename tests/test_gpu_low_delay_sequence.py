"""A low-delay sequence through the device, picture after picture, with NOTHING of the reference encoder in between: the I picture through
the all-intra loop (uvghip_loop_plan_*), every P / B picture through uvghip_loop_pb_run with the DEVICE's own earlier output pictures and
motion as its references.  Every output picture equals the picture the reference encoder produced (tests/golden/ref_inter_*: final_y/u/v),
every picture's slice data is in the encoder's .266 byte for byte."""
import ctypes
import os
import numpy as np
import pytest
import helpers as H

pytestmark = pytest.mark.gpu
GOLDENS = ["ref_inter_136x72_10_qp22_4frames", "ref_inter_192x128_8_qp17_5frames", "ref_inter_264x136_8_qp32_9frames",
           # other tools than --preset medium's: P slices (no bi-prediction) without the temporal candidate; no fractional search, no early skip
           "ref_inter_136x72_8_qp27_4frames_p_notmvp", "ref_inter_192x128_10_qp24_4frames_subme0_noskip",
           "ref_inter_136x72_8_qp27_17frames_ra16", "ref_inter_136x72_10_qp22_17frames_ra16", "ref_inter_136x72_8_qp27_9frames_ra8", "ref_inter_136x72_8_qp27_5frames_rd1", "ref_inter_136x72_8_qp27_33frames_ra16p16"]


@pytest.mark.parametrize("name", GOLDENS)
def test_sequence_closed_loop_on_the_device(hip, name):
    import torch
    from uvg266_amd import api, lib
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, pics, P = H.inter_pictures_from_golden(g)
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    ctus, n4 = wc * hc, hc * 16 * wc * 16
    tdt = torch.uint8 if depth == 8 else torch.uint16
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    stream = g["bitstream"].tobytes()
    by_poc = {}                       # poc -> (planes, motion table) made by the device
    keep = []
    coded = 0
    qp0 = int(g["dims"][3])
    mine = b""                        # the NAL units of every picture, from device outputs only
    irap_poc = 0
    for fr, d, prm, F, _ in H.iter_inter_frames(W, Hh, P):
        src = [dev(p) for p in pics[fr]]
        slice_type, poc = int(d["meta"][6]), int(d["refs"][51])
        off = g["row_off"][fr * hc:fr * hc + hc + 1]
        if slice_type == 2:
            cp = api.ctu_params(W, Hh, prm.qp, lam=prm.lam)
            loop = api.ClosedLoop(cp, [tuple(src)])
            loop.run()
            torch.cuda.synchronize()
            out = loop.out[0]
            mot = torch.zeros((hc * 16, wc * 16, 8), dtype=torch.int32, device="cuda")
            mot[:, :, 0] = loop.cu[0][:, :, 2].to(torch.int32)
            mot[:, :, 6:8] = -1
            rows, nb = loop.slice_data()
            rows, nb = rows[0].cpu().numpy(), nb[0].cpu().numpy()
            keep.append(loop)
        else:
            q = lib.LoopPbPicture()
            s = q.search
            cp = H.ctu_params(prm)
            ctypes.memmove(ctypes.byref(s.params), ctypes.byref(cp), ctypes.sizeof(cp))
            t = dict(rec=[torch.zeros((Hh >> c, W >> c), dtype=tdt, device="cuda") for c in (0, 1, 1)],
                     out=[torch.zeros((Hh >> c, W >> c), dtype=tdt, device="cuda") for c in (0, 1, 1)],
                     scu=torch.zeros(n4 * 32, dtype=torch.uint8, device="cuda"), i4=torch.zeros(n4 * 8, dtype=torch.uint8, device="cuda"),
                     mot=torch.zeros((hc * 16, wc * 16, 8), dtype=torch.int32, device="cuda"), co=torch.zeros(ctus * 6144, dtype=torch.int16, device="cuda"),
                     mo=torch.zeros(ctus * 3 * 257, dtype=torch.int32, device="cuda"), mi=torch.zeros(ctus * 3 * 18, dtype=torch.int32, device="cuda"))
            c = s.pic
            c.src_y, c.src_u, c.src_v = (a.data_ptr() for a in src)
            c.rec_y, c.rec_u, c.rec_v = (a.data_ptr() for a in t["rec"])
            c.src_stride = c.rec_stride = W
            c.src_stride_c = c.rec_stride_c = W // 2
            c.cu, c.cu_stride, c.coeff, c.models = t["scu"].data_ptr(), wc * 16, t["co"].data_ptr(), t["mo"].data_ptr()
            for f in ("slice_type", "poc", "n_refs", "tmvp", "max_merge", "merge_level", "frame_qp", "bipred", "fme_level", "early_skip", "depth_inter_min",
                      "depth_inter_max"):
                setattr(s, f, getattr(F, f))
            for i in range(16):
                s.ref_pocs[i], s.l[0][i], s.l[1][i] = F.ref_pocs[i], F.l[0][i], F.l[1][i]
            s.l_size[0], s.l_size[1] = F.l_size[0], F.l_size[1]
            s.ref_stride, s.ref_stride_c, s.ref_motion_stride = W, W // 2, wc * 16
            for i in range(F.n_refs):
                planes, rm = by_poc[F.ref_pocs[i]]
                s.ref_y[i], s.ref_u[i], s.ref_v[i], s.ref_motion[i] = planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(), rm.data_ptr()
            s.inter4, s.models_inter, s.trees, s.motion_out = t["i4"].data_ptr(), t["mi"].data_ptr(), None, t["mot"].data_ptr()
            q.out_y, q.out_u, q.out_v = (a.data_ptr() for a in t["out"])
            q.out_stride, q.out_stride_c = W, W // 2
            ws, info, models, rows, nb = api.loop_pb_run([q], depth)
            torch.cuda.synchronize()
            out, mot = t["out"], t["mot"]
            rows, nb = rows[0].cpu().numpy(), nb[0].cpu().numpy()
            keep += [t, ws, src]
        by_poc[poc] = (out, mot)
        for cidx, nme in enumerate(("final_y", "final_u", "final_v")):
            assert np.array_equal(out[cidx].cpu().numpy(), g[nme][fr]), (name, fr, nme)
        whole = b""
        for r in range(hc):
            want = g["row_bytes"][off[r]:off[r + 1]]
            assert nb[r] == len(want) and np.array_equal(rows[r, :nb[r]], want), (name, fr, "row", r, int(nb[r]), len(want))
            whole += rows[r, :nb[r]].tobytes()
        assert stream.find(whole) > 0, (name, fr)
        # the picture's NAL units: hash SEI from the device's checksum of its output picture, slice header from the rows' lengths
        sums = np.ascontiguousarray(api.picture_checksum(*out).cpu().numpy().view(np.uint32))
        sizes = np.ascontiguousarray(nb, np.int32)
        rows_h = np.ascontiguousarray(rows[:, :int(sizes.max())])
        cap = int(sizes.sum()) + 128 + 4 * hc
        buf, n = np.zeros(cap, np.uint8), ctypes.c_size_t(0)
        frame_qp = int(d["meta"][7])
        if slice_type == 2 and poc == 0:
            rc = hip.uvghip_write_idr_nals_ra(poc, H.poc_lsb_bits(g), frame_qp - qp0, 1, H.ptr(rows_h), rows_h.shape[1], H.ptr(sizes), hc, H.ptr(sums), H.ptr(buf), cap, ctypes.byref(n))
        elif slice_type == 2:          # the CRA picture of a later intra period: its reference buffer is in the record
            n_refs = int(d["refs"][0])
            cfg = g["cfg"] if "cfg" in g.files else (1, 6, 2, 1, 4, 1)
            rc = H.write_inter_nals(hip, g, poc, 2, [int(p) for p in d["refs"][1:1 + n_refs]], int(cfg[3]), int(cfg[0]), frame_qp - qp0, rows_h, sizes, sums, buf, n, irap_poc=irap_poc)
        else:
            rc = H.write_inter_nals(hip, g, poc, slice_type, [F.ref_pocs[i] for i in range(F.n_refs)], F.bipred, F.tmvp, frame_qp - qp0, rows_h, sizes, sums, buf, n, irap_poc=irap_poc)
        if slice_type == 2:
            irap_poc = poc
        assert rc == 0
        mine += buf[:n.value].tobytes()
        coded += 1
    assert coded == int(g["dims"][4])
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert at > 0 and stream[:at] + mine == stream, "the encoder's parameter sets + the device's pictures = the encoder's whole .266"


def _frame_rows(g):
    """Per-picture (meta, lam, refs) of a full ref_inter_* golden (its records are per CTU)."""
    frames = int(g["dims"][4])
    first = {}
    for k in range(len(g["meta"])):
        first.setdefault(int(g["meta"][k][0]), k)
    ks = [first[f] for f in range(frames)]
    return g["meta"][ks], g["lam"][ks], g["refs"][ks]


@pytest.mark.parametrize("name,n_seq,in_flight", [("ref_inter_264x136_8_qp32_9frames", 3, 1), ("ref_inter_136x72_8_qp27_17frames_ra16", 2, 1), ("ref_inter_136x72_8_qp27_17frames_ra16", 2, 8),
                                                  ("ref_intercrc_1920x1080_8_qp27_5frames", 2, 1), ("ref_intercrc_1920x1080_8_qp27_17frames_ra16", 1, 16), ("ref_inter_264x136_8_qp32_9frames", 2, 4),
                                                  ("ref_inter_136x72_8_qp27_17frames_ra16", 2, -1), ("ref_inter_136x72_10_qp22_17frames_ra16", 1, -1), ("ref_inter_136x72_8_qp27_9frames_ra8", 2, -1), ("ref_intercrc_136x72_8_qp27_65frames_ra16", 2, -1), ("ref_inter_136x72_8_qp27_33frames_ra16p16", 1, -1), ("ref_intercrc_1920x1080_8_qp27_17frames_ra16", 1, -1), ("ref_inter_264x136_8_qp32_9frames", 2, -1),
                                                  ("ref_intercrc_1920x1080_10_qp32_3frames", 1, 1), ("ref_intercrc_3840x2160_10_qp27_3frames", 1, 1),
                                                  ("ref_intercrc_3840x2160_10_qp27_17frames_ra16", 1, -1)])
def test_low_delay_loop_of_several_sequences(hip, name, n_seq, in_flight):
    """api.LowDelayLoop (what bench.py times for BASELINE configs[2]): n_seq sequences side by side, every picture group one call.  The
    random-access cases (_ra16) also run with their pictures in flight (deps from the reference lists).  The
    1080p and 2160p cases are checked through the CRCs of tests/golden/ref_intercrc_* (output pictures and every row's bytes of the
    reference's run; 2160p 10-bit is the geometry of BASELINE configs[3], and its _ra16 case that configuration's GOP -- without ALF)."""
    import zlib
    import torch
    from uvg266_amd import api
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, qp0, frames = (int(a) for a in g["dims"])
    hc = (Hh + 63) // 64
    crc_only = "final_crc" in g.files
    meta, lam, refs = (g["meta"], g["lam"], g["refs"]) if crc_only else _frame_rows(g)
    states = H.frame_states_from_records(meta, lam, refs)
    pics = H.golden_sources(g)          # coding order (random access: not the display order)
    src = [[tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pics[f]) for f in range(frames)] for _ in range(n_seq)]
    loop = api.LowDelayLoop(W, Hh, depth, n_seq, states, src, by_level=in_flight < 0)          # (in_flight -1: the pictures of a DAG level share a launch)
    in_flight = max(in_flight, 1)
    import time
    t0 = time.time()
    loop.run(in_flight=in_flight)          # (in_flight > 1: a picture waits only for the pictures it references -- a random-access GOP's layers side by side)
    torch.cuda.synchronize()
    print(f"{name}: {frames} pictures x {n_seq} sequence(s), {in_flight} in flight: {time.time() - t0:.2f} s")
    for f in range(frames):
        rows, nb = loop.rows[f].cpu().numpy(), loop.row_bytes[f].cpu().numpy()
        for s in range(n_seq):
            planes = [a.cpu().numpy() for a in loop.out[f][s]]
            if crc_only:
                assert zlib.crc32(b"".join(np.ascontiguousarray(a).tobytes() for a in planes)) == int(g["final_crc"][f]), (name, "picture", f, "sequence", s)
                for r in range(hc):
                    assert nb[s, r] == int(g["row_len"][f * hc + r]) and zlib.crc32(rows[s, r, :nb[s, r]].tobytes()) == int(g["row_crc"][f * hc + r]), (name, f, s, "row", r)
            else:
                for cidx, nme in enumerate(("final_y", "final_u", "final_v")):
                    assert np.array_equal(planes[cidx], g[nme][f]), (name, f, s, nme)
                off = g["row_off"][f * hc:f * hc + hc + 1]
                for r in range(hc):
                    want = g["row_bytes"][off[r]:off[r + 1]]
                    assert nb[s, r] == len(want) and np.array_equal(rows[s, r, :nb[s, r]], want), (name, f, s, "row", r)
