"""uvghip_tiles_plan_* on the device: all-intra pictures under --tiles <cols>x<rows> --wpp, every tile an independent rectangle of the
closed loop, against the files the real encoder wrote with the same --tiles (tests/golden/ref_tiles_*.npz): the whole .266 behind the
parameter sets, every substream, the pictures the encoder returned -- at small sizes byte by byte, at BASELINE's sizes by CRC."""
import ctypes
import zlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

FULL = ["ref_tiles_264x136_8_qp27_2x2_1frames", "ref_tiles_192x192_8_qp37_1x3_1frames", "ref_tiles_320x192_8_qp22_5x1_1frames", "ref_tiles_416x240_10_qp32_3x2_2frames",
        "ref_tiles_456x264_8_qp27_3x2_1frames_split"]          # (the last: --tiles-width-split / --tiles-height-split, columns of 1 / 4 / 3 CTUs)
CRC = ["ref_tiles_1920x1080_8_qp22_2x2_2frames_crc", "ref_tiles_3840x2160_10_qp22_4x2_1frames_crc"]


def tiled_loop(g):
    import torch
    from uvg266_amd import api
    W, Hh, depth, qp, cols, rows = (int(a) for a in g["meta"])
    src = []
    for poc, t in enumerate(g["ts"]):
        y, u, v = H.varied_picture(W, Hh, int(t), depth)
        assert zlib.crc32(y.tobytes() + u.tobytes() + v.tobytes()) == int(g["src_crc"][poc]), "synthetic generator drifted from the golden's source"
        src.append(tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v)))
    prm = H.search_params(W, Hh, qp)
    grid = (g["col_ctus"], g["row_ctus"]) if "col_ctus" in g.files else (cols, rows)
    return api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src, grid)


def tile_substreams(tl, picture):
    """The substreams of one picture in the order of the bitstream, from the size classes' own loop plans (uvghip_tiles_plan_tile)."""
    import torch
    out = []
    for t in range(tl.n_tiles):
        plan, idx, rect, first = tl.tile(picture, t)
        a, b, cap, nr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        assert tl.L.uvghip_loop_plan_slice_data(plan, ctypes.byref(a), ctypes.byref(b), ctypes.byref(cap), ctypes.byref(nr)) == 0
        base = tl.ws.data_ptr()
        assert nr.value == (rect[3] + 63) // 64
        nb = tl.ws[b.value - base + 4 * idx * nr.value:b.value - base + 4 * (idx + 1) * nr.value].view(torch.int32).cpu().numpy()
        for r in range(nr.value):
            o = a.value - base + (idx * nr.value + r) * cap.value
            out.append(tl.ws[o:o + int(nb[r])].cpu().numpy())
    return out


@pytest.mark.parametrize("name", FULL)
def test_whole_file_of_the_encoder_under_tiles(hip, name):
    import torch
    g = H.ctu_golden(name)
    tl = tiled_loop(g)
    tl.run()
    nals = tl.nals()
    torch.cuda.synchronize()
    stream = g["bitstream"].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert at > 0 and stream[:at] + b"".join(nals) == stream          # parameter sets (the encoder's) + slice NAL + hash SEI per picture = the .266
    off = g["row_off"]
    for i in range(tl.n):
        assert np.array_equal(np.concatenate([p.cpu().numpy().reshape(-1) for p in tl.out[i]]), g["final"][i]), "the picture the encoder returned"
        subs = tile_substreams(tl, i)
        assert len(subs) == tl.n_substreams
        for k, s in enumerate(subs):
            assert np.array_equal(s, g["row_bytes"][off[i * tl.n_substreams + k]:off[i * tl.n_substreams + k + 1]]), f"substream {k} of picture {i}"
    # a second run of the same plan, and the pictures one by one
    tl.run()
    assert [tl.nals(first_poc=i, first=i, count=1)[0] for i in range(tl.n)] == nals


@pytest.mark.parametrize("name", CRC)
def test_baseline_sizes_under_tiles_by_crc(hip, name):
    import torch
    g = H.ctu_golden(name)
    tl = tiled_loop(g)
    tl.run()
    nals = b"".join(tl.nals())
    torch.cuda.synchronize()
    assert len(nals) == int(g["bitstream_tail_len"]) and zlib.crc32(nals) == int(g["bitstream_tail_crc"])
    for i in range(tl.n):
        assert zlib.crc32(np.concatenate([p.cpu().numpy().reshape(-1) for p in tl.out[i]]).tobytes()) == int(g["final_crc"][i])
        subs = tile_substreams(tl, i)
        assert np.array_equal(np.array([zlib.crc32(s.tobytes()) for s in subs], np.uint32), g["row_crc"][i * tl.n_substreams:(i + 1) * tl.n_substreams])


def test_one_tile_is_the_closed_loop(hip):
    """--tiles 1x1 is no tiles: the plan's bytes are uvghip_loop_plan_group_nals' (one size class, run on the caller's stream)."""
    import torch
    from uvg266_amd import api, layout
    W, Hh, depth, qp = 200, 136, 8, 32
    prm = H.search_params(W, Hh, qp)
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W, Hh, t, depth)) for t in (5, 6)]
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src)
    cl.run()
    want = cl.group_nals()
    tl = api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src, (1, 1))
    assert (tl.n_tiles, tl.n_classes, tl.n_substreams) == (1, 1, 3)
    tl.run()
    assert tl.nals() == want
    for i in range(2):
        assert all(torch.equal(a, b) for a, b in zip(tl.out[i], cl.out[i])) and torch.equal(tl.coeff[i], cl.coeff[i]) and torch.equal(tl.cu[i], cl.cu[i])


def test_tiles_in_strided_planes_and_refusals(hip):
    """The whole pictures may live in larger allocations (a stride beyond the width); what the encoder refuses is refused."""
    import torch
    from uvg266_amd import api
    g = H.ctu_golden("ref_tiles_264x136_8_qp27_2x2_1frames")
    W, Hh, depth, qp, cols, rows = (int(a) for a in g["meta"])
    y, u, v = H.varied_picture(W, Hh, int(g["ts"][0]), depth)
    big = [torch.full((p.shape[0] + 8, p.shape[1] + 64), 77, dtype=torch.uint8).cuda() for p in (y, u, v)]
    src = []
    for b, p in zip(big, (y, u, v)):
        view = b[:p.shape[0], :p.shape[1]]
        view.copy_(torch.from_numpy(np.ascontiguousarray(p)))
        src.append(view)
    prm = H.search_params(W, Hh, qp)
    tl = api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(src)], (cols, rows))
    tl.run()
    stream = g["bitstream"].tobytes()
    assert stream[stream.find(b"\x00\x00\x01\x00\x41"):] == tl.nals()[0]
    with pytest.raises(ValueError):
        api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(src)], (6, 1))          # more tile columns than CTU columns


@pytest.mark.parametrize("name,world", [("ref_tiles_264x136_8_qp27_2x2_1frames", 2), ("ref_tiles_416x240_10_qp32_3x2_2frames", 3), ("ref_tiles_320x192_8_qp22_5x1_1frames", 4),
                                        ("ref_tiles_1920x1080_8_qp22_2x2_2frames_crc", 4), ("ref_tiles_3840x2160_10_qp22_4x2_1frames_crc", 8)])
def test_tiles_over_emulated_ranks(hip, name, world):
    """The tiles of a picture over the devices of a node, emulated on one: every "rank" has its own buffers and a plan of the tiles it owns
    (uvghip_tiles_plan_create_owned), touches nothing outside them, and what the ranks contribute (uvghip_tiles_plan_substreams: lengths,
    bytes, checksum terms) adds up to the encoder's NAL units (uvg266_amd.tiles.write_nals; the exchange itself: tests/test_tiles.py, gloo)."""
    import torch
    from uvg266_amd import api, tiles
    g = H.ctu_golden(name)
    W, Hh, depth, qp, cols, rows = (int(a) for a in g["meta"])
    rects, _ = api.tile_grid(W, Hh, cols, rows)
    owner = tiles.assign(rects, world)
    src = []
    for poc, t in enumerate(g["ts"]):
        y, u, v = H.varied_picture(W, Hh, int(t), depth)
        src.append(tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v)))
    prm = H.search_params(W, Hh, qp)
    parts = []
    for rank in range(world):
        tl = api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src, (cols, rows), owned=owner == rank)
        for o in tl.out:
            for p in o:
                p.fill_(5)          # poison: a rank must not write outside its tiles
        tl.run()
        lens, data, sums = tl.substreams()
        with pytest.raises(RuntimeError):
            tl.nals()               # a part of the tiles cannot write the picture's NAL units by itself
        for i in range(tl.n):
            for t, (tx, ty, tw, th) in enumerate(tuple(int(a) for a in r) for r in rects):
                if owner[t] != rank:
                    assert all(bool((p[(ty >> c):(ty + th) >> c, (tx >> c):(tx + tw) >> c] == 5).all()) for p, c in zip(tl.out[i], (0, 1, 1))), "a tile of another rank was written"
        parts.append((lens, data, sums))
        del tl
    nals = b"".join(tiles.write_nals(np.stack([p[0] for p in parts]), [p[1] for p in parts], np.stack([p[2] for p in parts])))
    if "bitstream" in g.files:
        stream = g["bitstream"].tobytes()
        assert stream[stream.find(b"\x00\x00\x01\x00\x41"):] == nals
    else:
        assert len(nals) == int(g["bitstream_tail_len"]) and zlib.crc32(nals) == int(g["bitstream_tail_crc"])


@pytest.mark.parametrize("case", [(200, 136, 8, 32, 2, 1, 5), (328, 200, 10, 27, 3, 3, 1004), (448, 72, 8, 22, 7, 1, 2), (136, 264, 10, 37, 1, 4, 2003), (264, 264, 8, 17, 4, 4, 3),
                                  (520, 136, 8, 27, 4, 2, 4011)])
def test_other_grids_against_the_oracle(hip, orc, case):
    """Grids, sizes and content the goldens do not hold (partial CTUs inside the last tiles, one-CTU tiles in both directions, 16 tiles of
    one picture, noise and impulses): expected = the oracle's chain over every tile as a picture of its own -- the construction
    tests/test_tiles.py holds to the reference's --tiles runs -- and the library's NAL writer over its rows."""
    import torch
    from uvg266_amd import api, lib
    W, Hh, depth, qp, cols, rows, t = case
    y, u, v = H.varied_picture(W, Hh, t, depth)
    rects, first = api.tile_grid(W, Hh, cols, rows)
    final = [np.zeros_like(y), np.zeros_like(u), np.zeros_like(v)]
    want_rows, want_coeff = [], []
    for tx, ty, tw, th in (tuple(int(a) for a in r) for r in rects):
        sub = [np.ascontiguousarray(p[(ty >> c):(ty + th) >> c, (tx >> c):(tx + tw) >> c]) for p, c in ((y, 0), (u, 1), (v, 1))]
        prm = H.search_params(tw, th, qp)
        s = H.oracle_search_picture(orc, depth, prm, *sub)
        f = H.oracle_sao_picture(orc, depth, tw, th, qp, prm.lam, tuple(sub), (s["rec_y"], s["rec_u"], s["rec_v"]), H.scu_from_cu(s["cu"], qp))
        data, off, _ = H.oracle_encode_rows(orc, depth, prm, s, f["sao"])
        want_rows += [data[off[r]:off[r + 1]] for r in range(len(off) - 1)]
        want_coeff.append(s["coeff"])
        for p, k, c in ((final[0], "final_y", 0), (final[1], "final_u", 1), (final[2], "final_v", 1)):
            p[(ty >> c):(ty + th) >> c, (tx >> c):(tx + tw) >> c] = f[k]
    prm = H.search_params(W, Hh, qp)
    tl = api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))], (cols, rows))
    tl.run()
    nal = tl.nals()[0]
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(tl.out[0], final)), "output picture"
    got = tile_substreams(tl, 0)
    assert len(got) == len(want_rows) and all(np.array_equal(a, b) for a, b in zip(got, want_rows)), "substreams"
    assert np.array_equal(tl.coeff[0].cpu().numpy(), np.concatenate(want_coeff)), "levels in tile-scan order"          # (first_ctu: the tiles' CTUs one tile after the other)
    assert [int(f) for f in first] == list(np.cumsum([0] + [len(c) for c in want_coeff[:-1]]))
    sizes = np.array([len(r) for r in want_rows], np.int32)
    packed = np.zeros((len(sizes), int(sizes.max())), np.uint8)
    for k, r in enumerate(want_rows):
        packed[k, :len(r)] = r
    sums = np.array([H.picture_checksum(p, depth) for p in final], np.uint32)
    cap = int(sizes.sum()) + 64 + 4 * len(sizes)
    out = np.zeros(cap, np.uint8)
    n = ctypes.c_size_t(0)
    L = lib.load_library()
    assert L.uvghip_write_picture_nals(0, 1, H.ptr(packed), packed.shape[1], H.ptr(sizes), len(sizes), H.ptr(sums), H.ptr(out), cap, ctypes.byref(n)) == 0
    assert out[:n.value].tobytes() == nal
