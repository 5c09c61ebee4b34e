"""uvghip_write_picture_nals (a host function of the library: slice NAL with the entry points + the rows' substreams, decoded
picture hash SEI) and uvghip_picture_checksum: behind the encoder's parameter sets they complete the .266 of a one-picture
all-intra encode.  Checked against the files the real encoder wrote (tests/golden/ref_ctu*.npz: bitstream) and, on random row
sizes / checksums incl. the zero runs that need emulation prevention, against the bit-by-bit restatement in helpers.py."""
import ctypes

import numpy as np
import pytest

import helpers as H

FULL = ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32"]


def write_nals(L, sizes, rows_2d, sums, poc=0, sao=1, cap=None):
    sizes = np.ascontiguousarray(sizes, np.int32)
    rows_2d = np.ascontiguousarray(rows_2d, np.uint8)
    cap = int(sizes.sum()) + 64 + 4 * len(sizes) if cap is None else cap
    out = np.zeros(max(cap, 1), np.uint8)
    n = ctypes.c_size_t(0)
    ck = None if sums is None else np.ascontiguousarray(sums, np.uint32)
    rc = L.uvghip_write_picture_nals(poc, sao, H.ptr(rows_2d), rows_2d.shape[1], H.ptr(sizes), len(sizes), None if ck is None else H.ptr(ck), H.ptr(out), cap, ctypes.byref(n))
    return rc, out[:min(n.value, cap)].tobytes(), n.value


def golden_rows(g):
    off = g["row_off"]
    sizes = np.diff(off).astype(np.int32)
    rows = np.zeros((len(sizes), int(sizes.max())), np.uint8)
    for r in range(len(sizes)):
        rows[r, :sizes[r]] = g["row_bytes"][off[r]:off[r + 1]]
    return sizes, rows


@pytest.mark.parametrize("name", FULL)
def test_whole_file_of_the_encoder_from_its_rows_and_final_picture(name):
    from uvg266_amd import lib
    L = lib.load_library()            # host function: no device
    g = H.ctu_golden(name)
    depth = int(g["meta"][2])
    stream = g["bitstream"].tobytes()
    sizes, rows = golden_rows(g)
    sums = [H.picture_checksum(g[k], depth) for k in ("final_y", "final_u", "final_v")]
    rc, nals, n = write_nals(L, sizes, rows, sums)
    assert rc == 0 and n == len(nals)
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert at > 0 and stream[at:] == nals, "slice NAL + hash SEI"
    assert stream[:at] + nals == stream                      # parameter sets (the encoder's) + these bytes = the whole .266
    assert nals == H.picture_nals(sizes, [rows[r, :sizes[r]].tobytes() for r in range(len(sizes))], sums)


@pytest.mark.parametrize("name", ["ref_stream_192x128_8_qp27_3frames", "ref_stream_136x72_10_qp32_18frames"])
def test_whole_multi_picture_stream_from_the_oracle_chain(orc, name):
    """Several pictures of one -p 1 stream: parameter sets once, then per picture slice NAL + hash SEI (IDR_N_LP first, IDR_W_RADL with
    a long start code afterwards, the 4-bit POC wrapping at 16).  Every picture through the oracle's chain (search -> filters -> SAO ->
    row coder), the NAL units from the library's host function: parameter sets of the encoder + these bytes = the encoder's file."""
    from uvg266_amd import lib, layout
    import zlib
    L = lib.load_library()
    g = H.ctu_golden(name)
    W, Hh, depth, qp = (int(a) for a in g["meta"])
    prm = H.search_params(W, Hh, qp)
    stream = g["bitstream"].tobytes()
    mine = b""
    for poc, t in enumerate(g["ts"]):
        y, u, v = layout.synthetic_yuv420(W, Hh, int(t), depth)
        assert zlib.crc32(y.tobytes() + u.tobytes() + v.tobytes()) == int(g["src_crc"][poc])
        s = H.oracle_search_picture(orc, depth, prm, y, u, v)
        f = H.oracle_sao_picture(orc, depth, W, Hh, qp, prm.lam, (y, u, v), (s["rec_y"], s["rec_u"], s["rec_v"]), H.scu_from_cu(s["cu"], qp))
        data, off, _ = H.oracle_encode_rows(orc, depth, prm, s, f["sao"])
        sizes = np.diff(off).astype(np.int32)
        rows = np.zeros((len(sizes), int(sizes.max())), np.uint8)
        for r in range(len(sizes)):
            rows[r, :sizes[r]] = data[off[r]:off[r + 1]]
        sums = [H.picture_checksum(f[k], depth) for k in ("final_y", "final_u", "final_v")]
        rc, nals, n = write_nals(L, sizes, rows, sums, poc=poc)
        assert rc == 0
        mine += nals
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert at > 0 and stream[:at] + mine == stream


def test_random_sizes_and_checksums_against_the_restatement():
    from uvg266_amd import lib
    L = lib.load_library()
    rng = np.random.default_rng(5)
    for trial in range(300):
        n_rows = int(rng.integers(1, 40))
        sizes = rng.choice([1, 2, 3, 255, 256, 257, 511, 512, 65535, 65536, 70000], n_rows) if trial % 3 == 0 else rng.integers(1, 3000, n_rows)
        rows = rng.integers(0, 256, (n_rows, int(max(sizes))), dtype=np.uint8)
        sums = rng.choice([0, 1, 2, 3, 0x300, 0x10000, 0x3000000, 0xffffffff, 0x00000100], 3) if trial % 2 == 0 else rng.integers(0, 2 ** 32, 3)
        ck = None if trial % 7 == 0 else sums
        poc, sao = int(rng.integers(0, 40)), int(rng.integers(0, 2))
        rc, nals, n = write_nals(L, sizes, rows, ck, poc, sao)
        assert rc == 0
        assert nals == H.picture_nals(list(sizes), [rows[r, :sizes[r]].tobytes() for r in range(n_rows)], ck, poc, bool(sao)), trial


def test_refuses_what_it_cannot_write():
    from uvg266_amd import lib
    L = lib.load_library()
    rows = np.zeros((2, 8), np.uint8)
    rc, _, n = write_nals(L, [4, 4], rows, [1, 2, 3], cap=10)          # too small: says how much it needs
    assert rc != 0 and n > 10
    assert write_nals(L, [4, 9], rows, None)[0] != 0                   # a row longer than its slot
    assert write_nals(L, [4, 0], rows, None)[0] != 0                   # an empty row cannot be a substream


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_264x136_10_qp32"])
def test_device_outputs_complete_the_encoders_file(hip, name):
    """Search -> filters -> SAO -> arithmetic coder on the device (uvghip_loop_plan_run), the picture's checksum on the device, the
    NAL units on the host: parameter sets of the encoder + these bytes = the encoder's .266, the whole file."""
    import torch
    from uvg266_amd import api
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cl.run()
    out, nbytes = cl.slice_data()
    sums = api.picture_checksum(*cl.out[0]).cpu().numpy().view(np.uint32)
    assert [int(s) for s in sums] == [H.picture_checksum(g[k], depth) for k in ("final_y", "final_u", "final_v")]
    nb = nbytes.cpu().numpy()[0]
    rc, nals, n = write_nals(hip, nb, out[0].cpu().numpy(), sums)
    assert rc == 0
    stream = g["bitstream"].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert stream[:at] + nals == stream


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_stream_192x128_8_qp27_3frames", "ref_stream_136x72_10_qp32_18frames"])
def test_device_outputs_complete_a_multi_picture_stream(hip, name):
    """All pictures of the stream in ONE uvghip_loop_plan_run (they are independent under -p 1): rows, row lengths and final pictures
    from the device, checksums from uvghip_picture_checksum, NAL units from uvghip_write_picture_nals -> the encoder's whole .266."""
    import torch
    from uvg266_amd import api, layout
    g = H.ctu_golden(name)
    W, Hh, depth, qp = (int(a) for a in g["meta"])
    prm = H.search_params(W, Hh, qp)
    pics = [layout.synthetic_yuv420(W, Hh, int(t), depth) for t in g["ts"]]
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in yuv) for yuv in pics])
    cl.run()
    out, nbytes = cl.slice_data()
    nb = nbytes.cpu().numpy()
    rows = out.cpu().numpy()
    mine = b""
    for poc in range(len(pics)):
        sums = api.picture_checksum(*cl.out[poc]).cpu().numpy().view(np.uint32)
        rc, nals, n = write_nals(hip, nb[poc], rows[poc], sums, poc=poc)
        assert rc == 0
        assert cl.picture_nals(poc, poc) == nals          # the plan's own entry point: checksum, download and NAL units in one call
        mine += nals
    stream = g["bitstream"].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert stream[:at] + mine == stream
    # ... and the whole group in one call (uvghip_loop_plan_group_nals: one download), twice (the buffers it grew are reused)
    for _ in range(2):
        grp = cl.group_nals(0)
        assert len(grp) == len(pics) and b"".join(grp) == mine
    assert cl.group_nals(3) == [cl.picture_nals(i, 3 + i) for i in range(len(pics))]          # (other picture numbers)


@pytest.mark.parametrize("name", ["ref_inter_136x72_10_qp22_4frames", "ref_inter_192x128_8_qp17_5frames", "ref_inter_264x136_8_qp32_9frames",
                                  "ref_inter_136x72_8_qp27_4frames_p_notmvp", "ref_inter_192x128_10_qp24_4frames_subme0_noskip",
           "ref_inter_136x72_8_qp27_17frames_ra16", "ref_inter_136x72_10_qp22_17frames_ra16", "ref_inter_136x72_8_qp27_9frames_ra8", "ref_inter_136x72_8_qp27_5frames_rd1", "ref_inter_136x72_8_qp27_33frames_ra16p16"])
def test_whole_low_delay_file_from_rows_and_final_pictures(name):
    """A low-delay stream (--gop lp-g4d3t1) or a random-access one (--gop 16: pictures in coding order, lists with references in the future,
    six POC bits): behind the encoder's parameter sets, the IDR picture's NAL units (with its slice QP offset)
    and every B picture's -- picture header with the inter flags, slice type, reference picture lists, collocated picture, QP offset,
    entry points, the rows, the hash SEI -- from the library's host functions make up the encoder's whole .266."""
    import os
    from uvg266_amd import lib
    L = lib.load_library()
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, qp0, frames = (int(a) for a in g["dims"])
    hc = (Hh + 63) // 64
    stream = g["bitstream"].tobytes()
    first = {}
    for k in range(len(g["meta"])):
        first.setdefault(int(g["meta"][k][0]), k)
    mine = b""
    irap_poc = 0
    for f in range(frames):
        k = first[f]
        off = g["row_off"][f * hc:f * hc + hc + 1]
        sizes = np.diff(off).astype(np.int32)
        rows = np.zeros((hc, int(sizes.max())), np.uint8)
        for r in range(hc):
            rows[r, :sizes[r]] = g["row_bytes"][off[r]:off[r + 1]]
        sums = np.ascontiguousarray([H.picture_checksum(g[nme][f], depth) for nme in ("final_y", "final_u", "final_v")], np.uint32)
        slice_type, frame_qp, poc = int(g["meta"][k][6]), int(g["meta"][k][7]), int(g["refs"][k][51])
        cap = int(sizes.sum()) + 128 + 4 * hc
        out = np.zeros(cap, np.uint8)
        n = ctypes.c_size_t(0)
        n_refs = int(g["refs"][k][0])
        cfg = g["cfg"] if "cfg" in g.files else (1, 6, 2, 1, 4, 1)
        if slice_type == 2 and poc == 0:
            rc = L.uvghip_write_idr_nals_ra(poc, H.poc_lsb_bits(g), frame_qp - qp0, 1, H.ptr(rows), rows.shape[1], H.ptr(sizes), hc, H.ptr(sums), H.ptr(out), cap, ctypes.byref(n))
        else:          # (an I picture with a POC: the CRA picture of a later intra period, its reference buffer in the lists)
            rc = H.write_inter_nals(L, g, poc, slice_type, [int(p) for p in g["refs"][k][1:1 + n_refs]], int(cfg[3]), int(cfg[0]), frame_qp - qp0, rows, sizes, sums, out, n, irap_poc=irap_poc)
        if slice_type == 2:
            irap_poc = poc
        assert rc == 0
        mine += out[:n.value].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert at > 0
    body = stream[at:]
    if body != mine:
        i = next(j for j in range(min(len(body), len(mine))) if body[j] != mine[j])
        assert False, (name, "first differing byte", i, body[max(0, i - 8):i + 8].hex(), mine[max(0, i - 8):i + 8].hex(), len(body), len(mine))
    assert stream[:at] + mine == stream


def test_the_inter_writers_refuse_what_no_gop_structure_lists():
    """uvghip_write_picture_nals_pb / _ra (host functions): reference distances are positive, ascending and do not reach before the stream's
    first picture; a copied list 1 has no references in the future; at least one reference."""
    from uvg266_amd import lib
    L = lib.load_library()
    rows = np.zeros((2, 8), np.uint8); rows[:, 0] = 1
    sizes = np.array([4, 4], np.int32)
    sums = np.zeros(3, np.uint32)
    out = np.zeros(256, np.uint8)
    n = ctypes.c_size_t(0)
    def ra(poc, neg, pos, lsb=6):
        a, b = np.ascontiguousarray(neg, np.int32), np.ascontiguousarray(pos, np.int32)
        return L.uvghip_write_picture_nals_ra(poc, lsb, 0, len(a), H.ptr(a) if len(a) else None, len(b), H.ptr(b) if len(b) else None, 1, 0, 1, H.ptr(rows), rows.shape[1], H.ptr(sizes), 2,
                                              H.ptr(sums), H.ptr(out), len(out), ctypes.byref(n))
    assert ra(8, [8], [8]) == 0 and n.value > 0
    assert ra(8, [8, 16], [8]) != 0            # POC -8 does not exist
    assert ra(8, [4, 4], [8]) != 0             # not ascending
    assert ra(8, [0], [8]) != 0                # the picture itself
    assert ra(8, [], []) != 0                  # no reference at all
    assert ra(8, [8], [8], lsb=3) != 0         # fewer POC bits than the SPS can signal
    assert ra(4, [4], [4, 12]) == 0
    neg = np.array([1, 2], np.int32)
    assert L.uvghip_write_picture_nals_pb(3, 4, 0, 2, H.ptr(neg), 1, 1, 0, 1, H.ptr(rows), rows.shape[1], H.ptr(sizes), 2, H.ptr(sums), H.ptr(out), len(out), ctypes.byref(n)) == 0
    def gop(nal, slice_type, neg):
        a = np.ascontiguousarray(neg, np.int32)
        return L.uvghip_write_picture_nals_gop(nal, 16, 6, slice_type, len(a), H.ptr(a) if len(a) else None, 0, None, 1, 0, 1, H.ptr(rows), rows.shape[1], H.ptr(sizes), 2,
                                               H.ptr(sums), H.ptr(out), len(out), ctypes.byref(n))
    assert gop(9, 2, [16]) == 0                # a CRA picture: an I slice that lists its reference buffer
    assert gop(9, 2, []) == 0                  # ... or none
    assert gop(9, 0, [16]) != 0                # CRA is an I slice
    assert gop(3, 2, [16]) != 0                # an I slice is CRA (or IDR: uvghip_write_idr_nals_ra)
    assert gop(3, 0, [16]) == 0 and gop(0, 1, [16]) == 0
    assert gop(5, 0, [16]) != 0                # not a NAL unit type of this writer
    bad = np.array([2, 1], np.int32)
    assert L.uvghip_write_picture_nals_pb(3, 4, 0, 2, H.ptr(bad), 1, 1, 0, 1, H.ptr(rows), rows.shape[1], H.ptr(sizes), 2, H.ptr(sums), H.ptr(out), len(out), ctypes.byref(n)) != 0
