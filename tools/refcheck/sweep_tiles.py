#!/usr/bin/env python3
"""Dev-time parity sweep for tiles (CPU only, needs oracle/_ref like make_ctu_goldens.py): fresh runs of the real reference encoder with
--tiles <cols>x<rows> --wpp over random sizes / grids / depths / QPs / content, each compared with the construction csrc/tiles.hip is
built on -- every tile through the oracle's chain (search, in-loop filters, row coder) as a picture of its own size, the substreams in tile
raster order behind one slice header, the hash of the whole picture: the whole .266 behind the parameter sets and the output picture.
Nothing is written to tests/golden.

  python tools/refcheck/sweep_tiles.py [n_cases] [seed]"""
import os, sys, random, subprocess, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_ctu_goldens as M
import helpers as H
from uvg266_amd import api

ORC = H.load_oracle()


def one(W, Hh, depth, qp, ts, cols, rows):
    px = np.uint8 if depth == 8 else np.uint16
    yuv, out = "/tmp/sweep_tiles.yuv", "/tmp/sweep_tiles"
    pics = [H.varied_picture(W, Hh, t, depth) for t in ts]
    with open(yuv, "wb") as f:
        for p in pics:
            for pl in p:
                f.write(pl.astype(px).tobytes())
    subprocess.check_call([os.path.join(ROOT, "tools/refcheck/ctu_dump.sh"), str(depth), yuv, str(W), str(Hh), str(len(ts)), out,
                           "preset", "medium", "period", "1", "qp", str(qp), "tiles", f"{cols}x{rows}", "wpp", "1"], stderr=subprocess.DEVNULL)
    stream = open(out + ".266", "rb").read()
    F = [r for n, r in M.read_records(out + ".bin") if n == "final"]
    rects, _ = api.tile_grid(W, Hh, cols, rows)
    mine, bad = b"", []
    for poc, (y, u, v) in enumerate(pics):
        final = [np.zeros_like(y), np.zeros_like(u), np.zeros_like(v)]
        rows_b = []
        for tx, ty, tw, th in (tuple(int(a) for a in r) for r in rects):
            sub = [np.ascontiguousarray(p[(ty >> c):(ty + th) >> c, (tx >> c):(tx + tw) >> c]) for p, c in ((y, 0), (u, 1), (v, 1))]
            prm = H.search_params(tw, th, qp)
            s = H.oracle_search_picture(ORC, depth, prm, *sub)
            f = H.oracle_sao_picture(ORC, depth, tw, th, qp, prm.lam, tuple(sub), (s["rec_y"], s["rec_u"], s["rec_v"]), H.scu_from_cu(s["cu"], qp))
            data, off, _ = H.oracle_encode_rows(ORC, depth, prm, s, f["sao"])
            rows_b += [data[off[r]:off[r + 1]].tobytes() for r in range(len(off) - 1)]
            for p, k, c in ((final[0], "final_y", 0), (final[1], "final_u", 1), (final[2], "final_v", 1)):
                p[(ty >> c):(ty + th) >> c, (tx >> c):(tx + tw) >> c] = f[k]
        if not np.array_equal(np.concatenate([p.reshape(-1) for p in final]), np.concatenate([F[poc][1], F[poc][2], F[poc][3]])):
            bad.append(("output picture", poc))
        mine += H.picture_nals(np.array([len(r) for r in rows_b], np.int32), rows_b, [H.picture_checksum(p, depth) for p in final], poc=poc)
    at = stream.find(b"\x00\x00\x01\x00\x41")
    if stream[at:] != mine:
        bad.append(".266")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    fails = 0
    for _ in range(n):
        W, Hh = rng.choice([128, 136, 192, 200, 264, 320, 328, 448]), rng.choice([64, 72, 128, 136, 192, 200, 264])
        wc, hc = (W + 63) // 64, (Hh + 63) // 64
        cols, rows = rng.randint(1, wc), rng.randint(1, hc)
        depth, qp = rng.choice([8, 10]), rng.choice([12, 17, 22, 27, 32, 37, 42])
        ts = [rng.choice([0, 1000, 2000, 3000, 4000]) + rng.randint(0, 99) for _ in range(rng.choice([1, 1, 2]))]
        bad = one(W, Hh, depth, qp, ts, cols, rows)
        print(f"{W}x{Hh} {depth}-bit qp {qp} tiles {cols}x{rows} pictures {ts}: {'ok' if not bad else 'DIFFERS: ' + str(bad)}", flush=True)
        fails += bool(bad)
    print("cases that differ:", fails)
