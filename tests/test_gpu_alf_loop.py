"""ALF in the picture loop (BASELINE configs[3]: --alf full) with the derivation where SURVEY.md keeps it -- on the host, behind a
callback: api.ClosedLoop.alf_stage.  The device runs the closed loop of the source pictures (search -> deblocking -> SAO), hands the
frame statistics to `decide`, reconstructs with what it returns and codes the slice data with the ALF syntax.  Here `decide` replays the
decisions a real `--alf full` run made (tests/golden/ref_stream_*_alf, ref_ctu_*_alf: tools/refcheck/ctu_dump.c around
uvg_alf_enc_process): the pictures ALF leaves, the slice data and -- with the library's host writer for the APS NAL units, the slice header
and the hash SEI over the device's checksum of its own output -- the encoder's whole .266 come out of the device."""
import ctypes
import os
import zlib

import numpy as np
import pytest

import helpers as H
from test_alf_syntax import write_alf_picture_nals

pytestmark = pytest.mark.gpu


def replay(pictures, seen=None):
    """decide(i, stats) returning picture i's recorded decisions (and, once, asking for every statistic the derivation would read)."""
    def decide(i, stats):
        p = pictures[i]
        m = p["meta"]
        if seen is not None and not seen:
            cls = stats.classification()
            ee, yv, pa = stats.luma()
            ce, cy, cp = stats.chroma(1)
            xe, xy, xp = stats.cc(2)
            n = stats.rects_y.shape[0]
            assert tuple(cls.shape) == (stats.H // 4, stats.W // 4) and tuple(ee.shape) == (n, 25, 13, 13, 4, 4) and tuple(ce.shape) == (n, 1, 13, 13, 4, 4) and tuple(xe.shape) == (n, 7, 7)
            assert int(pa.sum().item()) >= 0 and int(xp.sum().item()) >= 0
            seen.append(1)
        return dict(alf_type=int(m[3]), enabled=[int(a) for a in m[4:7]], n_luma_aps=int(m[7]), luma_aps=p["luma_aps"], chroma_aps=p["chroma_aps"],
                    cc_enabled=[int(a) for a in m[17:19]], cc_filter_count=[int(a) for a in m[19:21]], cc_coeff=p["cc_coeff"], ctu_flags=p["flags"], filter_set_idx=p["set_idx"])
    return decide


def host_rows(rows, nbytes, i):
    nb = np.ascontiguousarray(nbytes[i].cpu().numpy(), np.int32)
    return np.ascontiguousarray(rows[i, :, :int(nb.max())].cpu().numpy()), nb


def test_a_four_picture_alf_stream_through_the_loop(hip):
    """Four pictures of one -p 1 --alf full stream as ONE group of the closed loop; later pictures open their access unit with the APS, refer
    to APSs of earlier pictures and use fixed filter sets only."""
    import torch
    from uvg266_amd import api
    g = H.ctu_golden("ref_stream_192x128_8_qp27_4frames_alf")
    W, Hh, depth, qp = (int(a) for a in g["meta"])
    prm = H.search_params(W, Hh, qp)
    src = []
    for poc, t in enumerate(g["ts"]):
        y, u, v = H.varied_picture(W, Hh, int(t), depth)
        assert zlib.crc32(y.tobytes() + u.tobytes() + v.tobytes()) == int(g["src_crc"][poc])
        src.append(tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v)))
    pictures = [{k[4:]: g[k][poc] for k in ("alf_meta", "alf_flags", "alf_set_idx", "alf_luma_aps", "alf_chroma_aps", "alf_cc_coeff")} for poc in range(len(src))]
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src)
    cl.run()
    seen = []
    alf_out, rows, nbytes = cl.alf_stage(replay(pictures, seen), source=src, classification_shift=int(pictures[0]["meta"][28]) + 4)
    torch.cuda.synchronize()
    assert seen
    mine = b""
    for poc in range(len(src)):
        sums = api.picture_checksum(*alf_out[poc]).cpu().numpy().view(np.uint32)
        r, nb = host_rows(rows, nbytes, poc)
        sel = g["aps_meta"][:, 0] == poc
        one = dict(alf_meta=pictures[poc]["meta"], aps_meta=g["aps_meta"][sel], aps_luma=g["aps_luma"][sel], aps_chroma=g["aps_chroma"][sel], aps_cc=g["aps_cc"][sel])
        mine += write_alf_picture_nals(hip, one, r, nb, sums, poc=poc)
    stream = g["bitstream"].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x89")
    assert at > 0 and stream[:at] + mine == stream, "the encoder's parameter sets + the device's pictures (APS, slice, hash SEI) = the encoder's --alf full .266"


@pytest.mark.parametrize("name", ["ref_ctu_320x192_10_qp27_alf", "ref_ctu_192x128_8_qp22_alf", "ref_ctu_256x128_10_qp27_alf_nocc"])
def test_one_picture_item_by_item(hip, name):
    """A picture of an --alf full (the last: --alf no-cc) run: the picture ALF got, the picture it left, the slice data, the file."""
    import torch
    from uvg266_amd import api
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))]
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src)
    cl.run()
    for o, k in zip(cl.out[0], ("alf_pre_y", "alf_pre_u", "alf_pre_v")):
        assert np.array_equal(o.cpu().numpy(), g[k]), ("the picture ALF gets", k)
    pic = {k[4:]: g[k] for k in ("alf_meta", "alf_flags", "alf_set_idx", "alf_luma_aps", "alf_chroma_aps", "alf_cc_coeff")}
    alf_out, rows, nbytes = cl.alf_stage(replay([pic]), source=src, classification_shift=int(pic["meta"][28]) + 4)
    torch.cuda.synchronize()
    for o, k in zip(alf_out[0], ("final_y", "final_u", "final_v")):
        assert np.array_equal(o.cpu().numpy(), g[k]), ("the picture ALF leaves", k)
    r, nb = host_rows(rows, nbytes, 0)
    assert np.array_equal(np.concatenate([[0], np.cumsum(nb)]), g["row_off"])
    assert np.array_equal(np.concatenate([r[i, :nb[i]] for i in range(len(nb))]), g["row_bytes"])
    sums = api.picture_checksum(*alf_out[0]).cpu().numpy().view(np.uint32)
    nals = write_alf_picture_nals(hip, g, r, nb, sums)
    stream = g["bitstream"].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x89")
    assert at > 0 and stream[:at] + nals == stream


def test_a_2160p_10bit_picture_by_crc(hip):
    """BASELINE configs[3]'s size: one 3840x2160 10-bit --alf full picture (ref_stream_3840x2160_10_qp22_1frames_alf_crc: the run's decisions
    and APSs, of its .266 the parameter sets and length + CRC of the rest) -- the source through the device's loop, the ALF stage with the
    decisions replayed, the library's NAL writer."""
    import torch
    from uvg266_amd import api
    g = np.load(os.path.join(H.GOLDEN, "ref_stream_3840x2160_10_qp22_1frames_alf_crc.npz"))
    W, Hh, depth, qp = (int(a) for a in g["meta"])
    y, u, v = H.varied_picture(W, Hh, int(g["ts"][0]), depth)
    assert zlib.crc32(y.tobytes() + u.tobytes() + v.tobytes()) == int(g["src_crc"][0])
    prm = H.search_params(W, Hh, qp)
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))]
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src)
    cl.run()
    pic = {k[4:]: g[k][0] for k in ("alf_meta", "alf_flags", "alf_set_idx", "alf_luma_aps", "alf_chroma_aps", "alf_cc_coeff")}
    seen = []
    alf_out, rows, nbytes = cl.alf_stage(replay([pic], seen), source=src, classification_shift=int(pic["meta"][28]) + 4)
    # the frame-level luma statistics: 25 classes x (the ee triangle, y, pix_acc), summed over the 2040 CTUs on the device
    sums = api.AlfStatistics(cl.out[0], src[0], W, Hh, int(pic["meta"][28]) + 4).luma_frame()
    torch.cuda.synchronize()
    assert tuple(sums.shape) == (25, 1509) and int((sums[:, 1508] > 0).sum().item()) > 0
    r, nb = host_rows(rows, nbytes, 0)
    ck = api.picture_checksum(*alf_out[0]).cpu().numpy().view(np.uint32)
    nals = write_alf_picture_nals(hip, dict(alf_meta=pic["meta"], aps_meta=g["aps_meta"], aps_luma=g["aps_luma"], aps_chroma=g["aps_chroma"], aps_cc=g["aps_cc"]), r, nb, ck)
    assert len(nals) == int(g["bitstream_tail_len"]) and zlib.crc32(nals) == int(g["bitstream_tail_crc"])


# (the 8-bit golden is a run WITHOUT worker threads, kept for its job order: there the ALF job runs the moment it is submitted, before the last
# CTUs' deferred SAO columns are written -- the picture it got is not the finished SAO picture, so the statistics are taken on the golden's)
@pytest.mark.parametrize("name,own", [("ref_alf_320x192_10_qp27_3frames", True), ("ref_alf_192x128_8_qp27_3frames", False), ("ref_alf_192x128_10_qp23_2frames", True)])
def test_the_frame_statistics_of_the_devices_own_picture_equal_the_encoders(hip, name, own):
    """The statistics the ALF stage hands the derivation, held to a real run: the device's closed loop of the run's source pictures (search ->
    deblocking -> SAO: the picture uvg_alf_enc_process got, sample for sample), then api.AlfStatistics on that output -- classification, the
    luma covariance per class summed over the CTUs (uvghip_alf_stats_compact_batch + uvghip_alf_cov_reduce), the chroma covariances -- against
    what alf_derive_stats_for_filtering (alf.c:4227) gathered in the encoder (cov_luma / cov_chroma of the goldens, taken by ctu_dump.c while
    the process still held them).  int64 sums and pix_acc exact."""
    import torch
    from uvg266_amd import api
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, qp, frames, t0, kind = (int(a) for a in g["dims"])
    prm = H.search_params(W, Hh, qp)
    host = [H.varied_picture(W, Hh, kind * 1000 + t0 + f, depth) for f in range(frames)]
    for f in range(frames):
        assert zlib.crc32(b"".join(p.tobytes() for p in host[f])) == int(g["src_crc"][f])
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pic) for pic in host]
    if own:
        cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src)
        cl.run()
        torch.cuda.synchronize()
    checked = 0
    for f in range(frames):
        if own:
            got = cl.out[f]
            for o, k in zip(got, ("pre_y", "pre_u", "pre_v")):
                assert np.array_equal(o.cpu().numpy(), g[k][f]), (name, f, "the picture ALF gets", k)
        else:
            got = tuple(torch.from_numpy(np.ascontiguousarray(g[k][f])).cuda() for k in ("pre_y", "pre_u", "pre_v"))
        if not int(g["meta"][f][29]):
            continue                      # (no CTU of the picture was filtered: the run's statistics were freed unseen)
        st = api.AlfStatistics(got, src[f], W, Hh, int(g["meta"][f][28]) + 4)
        assert np.array_equal(st.classification().cpu().numpy(), g["cls"][f]), (name, f, "classification")
        luma = st.luma_frame().cpu().numpy()
        assert np.array_equal(luma, g["cov_luma"][f]), (name, f, "luma", np.argwhere(luma != g["cov_luma"][f])[:4].tolist())
        for c in (1, 2):
            e, yv, pa = (a.cpu().numpy() for a in st.chroma(c))
            mine = H.alf_sum_layout(e.sum(axis=0), yv.astype(np.int64).sum(axis=0), pa.sum(axis=0), 7)[0]
            assert np.array_equal(mine, g["cov_chroma"][f][c - 1]), (name, f, "chroma", c)
        checked += 1
    assert checked >= 2
