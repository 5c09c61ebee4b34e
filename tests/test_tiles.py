"""Tiles (--tiles <cols>x<rows> --wpp) on the CPU: the library's host function uvghip_tile_grid against the encoder's uniform grid, and the
claim csrc/tiles.hip is built on -- a tile is a picture of its own (search, in-loop filters, context models, WPP rows), the slice data
the tiles' substreams in tile raster order, the hash the whole picture's -- checked with the oracle's chain per tile against the files the
real encoder wrote under --tiles (tests/golden/ref_tiles_*.npz, tools/refcheck/make_ctu_goldens.py::tiles)."""
import ctypes
import zlib

import numpy as np
import pytest

import helpers as H

FULL = ["ref_tiles_264x136_8_qp27_2x2_1frames", "ref_tiles_192x192_8_qp37_1x3_1frames", "ref_tiles_320x192_8_qp22_5x1_1frames", "ref_tiles_416x240_10_qp32_3x2_2frames",
        "ref_tiles_456x264_8_qp27_3x2_1frames_split"]


def golden_grid(g):
    """(cols, rows) of a uniform grid, or the columns' widths / rows' heights in CTUs of a --tiles-width-split / --tiles-height-split run."""
    return (g["col_ctus"], g["row_ctus"]) if "col_ctus" in g.files else (int(g["meta"][4]), int(g["meta"][5]))


def uniform(n_ctus, parts):
    return [(i + 1) * n_ctus // parts - i * n_ctus // parts for i in range(parts)]          # (src/encoder.c:445-451)


def expected_grid(W, Hh, cols, rows):
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    rects, first, at, y0 = [], [], 0, 0
    for th in uniform(hc, rows):
        x0 = 0
        for tw in uniform(wc, cols):
            rects.append((x0 * 64, y0 * 64, min(tw * 64, W - x0 * 64), min(th * 64, Hh - y0 * 64)))
            first.append(at)
            at += tw * th
            x0 += tw
        y0 += th
    return np.array(rects, np.int32), np.array(first, np.int32)


@pytest.mark.parametrize("case", [(264, 136, 2, 2), (1920, 1080, 2, 2), (3840, 2160, 4, 2), (1920, 1080, 7, 5), (416, 240, 3, 2), (64, 64, 1, 1), (8, 8, 1, 1)])
def test_tile_grid_is_the_encoders_uniform_grid(case):
    from uvg266_amd import api
    W, Hh, cols, rows = case
    rects, first = api.tile_grid(W, Hh, cols, rows)
    er, ef = expected_grid(W, Hh, cols, rows)
    assert np.array_equal(rects, er) and np.array_equal(first, ef)
    assert int((rects[:, 2] * rects[:, 3]).sum()) == W * Hh          # the tiles cover the picture exactly


def test_tile_grid_from_explicit_splits():
    """--tiles-width-split 64,320 --tiles-height-split 192 on 456x264 (encoder.c:452-478): columns of 1 / 4 / 3 CTUs (the last one cut by the
    picture's edge: 456 = 7.125 CTUs), rows of 3 / 2."""
    from uvg266_amd import api, lib
    rects, first = api.tile_grid(456, 264, [1, 4, 3], [3, 2])
    assert rects.tolist() == [[0, 0, 64, 192], [64, 0, 256, 192], [320, 0, 136, 192], [0, 192, 64, 72], [64, 192, 256, 72], [320, 192, 136, 72]]
    assert first.tolist() == [0, 3, 15, 24, 26, 34]
    L = lib.load_library()
    out = np.zeros((64, 4), np.int32)
    for cw, rh in [([1, 4, 2], [3, 2]), ([1, 4, 3], [3, 3]), ([0, 5, 3], [3, 2]), ([8], [5, 1])]:          # do not add up / an empty column
        cw, rh = np.array(cw, np.int32), np.array(rh, np.int32)
        assert L.uvghip_tile_grid_split(456, 264, cw.ctypes.data, len(cw), rh.ctypes.data, len(rh), out.ctypes.data, None) != 0
    g = H.ctu_golden("ref_tiles_456x264_8_qp27_3x2_1frames_split")
    assert g["col_ctus"].tolist() == [1, 4, 3] and g["row_ctus"].tolist() == [3, 2]


def test_tile_grid_refuses_what_the_encoder_refuses():
    from uvg266_amd import lib
    L = lib.load_library()
    rects = np.zeros((64 * 64, 4), np.int32)
    for W, Hh, cols, rows in [(264, 136, 6, 1), (264, 136, 1, 4), (264, 136, 0, 1), (4096, 4096, 48, 1), (0, 64, 1, 1)]:
        assert L.uvghip_tile_grid(W, Hh, cols, rows, rects.ctypes.data, None) != 0
    assert L.uvghip_tiles_workspace_bytes(8, 1, 264, 136, 6, 1) == 0


def substreams(g, picture, n_sub):
    off = g["row_off"]
    return [g["row_bytes"][off[picture * n_sub + k]:off[picture * n_sub + k + 1]] for k in range(n_sub)]


@pytest.mark.parametrize("name", FULL)
def test_a_tile_is_a_picture_of_its_own(orc, name):
    """Every tile through the oracle's chain as a picture of the tile's size: its substreams are the encoder's, the pictures stitched
    together are the picture the encoder returned, and slice NAL + hash SEI written by the library's host function from them complete the
    encoder's .266."""
    from uvg266_amd import api, lib
    L = lib.load_library()
    g = H.ctu_golden(name)
    W, Hh, depth, qp, cols, rows = (int(a) for a in g["meta"])
    rects, _ = api.tile_grid(W, Hh, *golden_grid(g))
    n_sub = int(sum((r[3] + 63) // 64 for r in rects))
    stream = g["bitstream"].tobytes()
    mine = b""
    for poc, t in enumerate(g["ts"]):
        y, u, v = H.varied_picture(W, Hh, int(t), depth)
        assert zlib.crc32(y.tobytes() + u.tobytes() + v.tobytes()) == int(g["src_crc"][poc])
        final = [np.zeros_like(y), np.zeros_like(u), np.zeros_like(v)]
        rows_b = []
        for tx, ty, tw, th in (tuple(int(a) for a in r) for r in rects):
            sub = [np.ascontiguousarray(p[(ty >> c):(ty + th) >> c, (tx >> c):(tx + tw) >> c]) for p, c in ((y, 0), (u, 1), (v, 1))]
            prm = H.search_params(tw, th, qp)
            s = H.oracle_search_picture(orc, depth, prm, *sub)
            f = H.oracle_sao_picture(orc, depth, tw, th, qp, prm.lam, tuple(sub), (s["rec_y"], s["rec_u"], s["rec_v"]), H.scu_from_cu(s["cu"], qp))
            data, off, _ = H.oracle_encode_rows(orc, depth, prm, s, f["sao"])
            rows_b += [data[off[r]:off[r + 1]] for r in range(len(off) - 1)]
            for p, k, c in ((final[0], "final_y", 0), (final[1], "final_u", 1), (final[2], "final_v", 1)):
                p[(ty >> c):(ty + th) >> c, (tx >> c):(tx + tw) >> c] = f[k]
        assert len(rows_b) == n_sub
        for k, (a, b) in enumerate(zip(rows_b, substreams(g, poc, n_sub))):
            assert np.array_equal(a, b), f"substream {k} of picture {poc}"
        assert np.array_equal(np.concatenate([p.reshape(-1) for p in final]), g["final"][poc]), "the picture the encoder returned"
        sizes = np.array([len(r) for r in rows_b], np.int32)
        packed = np.zeros((n_sub, int(sizes.max())), np.uint8)
        for k, r in enumerate(rows_b):
            packed[k, :len(r)] = r
        sums = np.array([H.picture_checksum(p, depth) for p in final], np.uint32)
        cap = int(sizes.sum()) + 64 + 4 * n_sub
        out = np.zeros(cap, np.uint8)
        n = ctypes.c_size_t(0)
        assert L.uvghip_write_picture_nals(poc, 1, H.ptr(packed), packed.shape[1], H.ptr(sizes), n_sub, H.ptr(sums), H.ptr(out), cap, ctypes.byref(n)) == 0
        mine += out[:n.value].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert at > 0 and stream[:at] + mine == stream          # parameter sets (the encoder's: its PPS carries the grid) + these bytes = the whole .266



# ---- the tiles of a picture over the ranks of a node (uvg266_amd/tiles.py) ---------------------------------------------------------------

def test_assign_is_a_balanced_partition():
    from uvg266_amd import api, tiles
    for W, Hh, cols, rows in [(1920, 1080, 2, 2), (1920, 1080, 4, 2), (3840, 2160, 4, 2), (3840, 2160, 8, 4), (416, 240, 3, 2)]:
        rects, _ = api.tile_grid(W, Hh, cols, rows)
        for world in (1, 2, 3, 4, 8):
            owner = tiles.assign(rects, world)
            assert owner.min() >= 0 and owner.max() < world and len(owner) == cols * rows
            counts = np.bincount(owner, minlength=world)
            assert counts.max() - counts.min() <= 1 or cols * rows < world          # near-equal tiles: near-equal counts
            assert np.array_equal(owner, tiles.assign(rects, world))                   # every rank computes the same table
    rects, _ = api.tile_grid(3840, 2160, 4, 2)
    assert sorted(tiles.assign(rects, 8)) == list(range(8))                           # configs[3]'s picture in 4 x 2 tiles: a tile per GPU


def golden_contribution(g, rects, owner, rank):
    """What a rank owning owner == rank would hand to tiles.gather_nals, cut out of the encoder's own run: its tiles' substreams and the
    terms of its tiles' samples in the picture checksum (position-dependent: the mask takes the sample's place in the PICTURE)."""
    W, Hh, depth = (int(a) for a in g["meta"][:3])
    n_sub = int(sum((r[3] + 63) // 64 for r in rects))
    count = len(g["ts"])
    lens = np.zeros((count, n_sub), np.int32)
    sums = np.zeros((count, 3), np.uint32)
    data = []
    off = g["row_off"]
    for i in range(count):
        planes = np.split(g["final"][i].astype(np.int64), [W * Hh, W * Hh + W * Hh // 4])
        planes = [planes[0].reshape(Hh, W), planes[1].reshape(Hh // 2, W // 2), planes[2].reshape(Hh // 2, W // 2)]
        s = 0
        for t, (tx, ty, tw, th) in enumerate(tuple(int(a) for a in r) for r in rects):
            for r in range((th + 63) // 64):
                if owner[t] == rank:
                    b = g["row_bytes"][off[i * n_sub + s]:off[i * n_sub + s + 1]]
                    lens[i, s] = len(b)
                    data.append(b)
                s += 1
            if owner[t] != rank:
                continue
            for c, p in enumerate(planes):
                sh = 1 if c else 0
                ys, xs = np.mgrid[ty >> sh:(ty + th) >> sh, tx >> sh:(tx + tw) >> sh]
                m = (xs ^ ys ^ (xs >> 8) ^ (ys >> 8)) & 0xFF
                v = p[ty >> sh:(ty + th) >> sh, tx >> sh:(tx + tw) >> sh]
                term = ((v & 0xFF) ^ m).sum() + (((v >> 8) & 0xFF) ^ m).sum() * (depth > 8)
                sums[i, c] = (int(sums[i, c]) + int(term)) & 0xFFFFFFFF
    return lens, np.concatenate(data) if data else np.zeros(0, np.uint8), sums


@pytest.mark.parametrize("name,world", [("ref_tiles_264x136_8_qp27_2x2_1frames", 2), ("ref_tiles_416x240_10_qp32_3x2_2frames", 3), ("ref_tiles_320x192_8_qp22_5x1_1frames", 4)])
def test_write_nals_from_the_ranks_contributions(name, world):
    """tiles.write_nals (no process group: the contributions side by side) = the encoder's bytes behind its parameter sets."""
    from uvg266_amd import api, tiles
    g = H.ctu_golden(name)
    W, Hh, depth, qp, cols, rows = (int(a) for a in g["meta"])
    rects, _ = api.tile_grid(W, Hh, cols, rows)
    owner = tiles.assign(rects, world)
    parts = [golden_contribution(g, rects, owner, r) for r in range(world)]
    nals = tiles.write_nals(np.stack([p[0] for p in parts]), [p[1] for p in parts], np.stack([p[2] for p in parts]))
    stream = g["bitstream"].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert stream[at:] == b"".join(nals)
    with pytest.raises(ValueError):          # a substream nobody owns
        tiles.write_nals(np.stack([p[0] for p in parts[1:]]), [p[1] for p in parts[1:]], np.stack([p[2] for p in parts[1:]]))


def _tile_rank(rank, world, port, name, q):
    import os
    import torch.distributed as dist
    from uvg266_amd import api, tiles
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = H.ctu_golden(name)
    W, Hh, depth, qp, cols, rows = (int(a) for a in g["meta"])
    rects, _ = api.tile_grid(W, Hh, cols, rows)
    owner = tiles.assign(rects, world)
    # (the device's part -- api.TiledLoop(owned=owner == rank).substreams() -- is played by the encoder's own record here: no GPU)
    nals = tiles.gather_nals(golden_contribution(g, rects, owner, rank), first_poc=0, sao=True)
    assert (nals is None) == (rank != 0)
    if rank == 0:
        q.put(b"".join(nals))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("ref_tiles_264x136_8_qp27_2x2_1frames", 2), ("ref_tiles_416x240_10_qp32_3x2_2frames", 3)])
def test_tiles_over_gloo_ranks(name, world):
    """2 and 3 processes, one per "GPU": every rank contributes its tiles, rank 0 ends up with the encoder's bytes."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_tile_rank, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    stream = H.ctu_golden(name)["bitstream"].tobytes()
    assert stream[stream.find(b"\x00\x00\x01\x00\x41"):] == got
