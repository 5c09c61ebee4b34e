#!/usr/bin/env python3
"""Dev-time parity sweep (CPU only, needs the survey's reference build like make_ctu_goldens.py): runs the real reference encoder
on small synthetic pictures over a grid of sizes / bit depths / QPs / seeds and compares, CTU by CTU, the search kernel's source
built for the host (tests/emul) with the reference's records: the three model sets, cu fields, trees, reconstruction, levels.
Nothing is written to tests/golden; a combination that differs is what to turn into a golden (make_ctu_goldens.full).

  python tools/refcheck/sweep_ctu.py [n_cases] [seed] [default|small|large|inter]"""
import os, sys, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_ctu_goldens as M
import helpers as H
from uvg266_amd import layout

ORC = H.load_oracle()


def one(W, Hh, depth, qp, t):
    tag = f"sweep_{W}x{Hh}_{depth}_qp{qp}_t{t}"
    S, Cd, src_crc, bs, px = M.run(W, Hh, depth, qp, t, tag, picture=H.varied_picture)
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    y, u, v = H.varied_picture(W, Hh, t, depth)
    prm = H.search_params(W, Hh, qp)
    lam = float(S[0][1][0]) if hasattr(S[0][1], "__len__") else float(S[0][1])
    assert abs(lam - prm.lam) < 1e-12 * max(1.0, lam), (lam, prm.lam)
    bad = []
    for who, r in (("emul", H.emul_search_picture(depth, prm, y, u, v)), ("oracle", H.oracle_search_picture(ORC, depth, prm, y, u, v))):
      for s, c in zip(S, Cd):
          x, yy, hh, ww, ccu, ctr, ry, ru, rv, cy, cuv = M.items(s, c, W, Hh)
          k = (yy // 64) * wc + x // 64
          ok = (np.array_equal(r["models"][k, 0], s[2][:1286]) and np.array_equal(r["models"][k, 1], s[3][:1286]) and np.array_equal(r["models"][k, 2], c[2][:1286])
                and np.array_equal(r["cu"][yy // 4:yy // 4 + hh // 4, x // 4:x // 4 + ww // 4], ccu)
                and np.array_equal(r["trees"][yy // 4:yy // 4 + hh // 4, x // 4:x // 4 + ww // 4], ctr)
                and np.array_equal(r["rec_y"][yy:yy + hh, x:x + ww], ry) and np.array_equal(r["rec_u"][yy // 2:(yy + hh) // 2, x // 2:(x + ww) // 2], ru)
                and np.array_equal(r["rec_v"][yy // 2:(yy + hh) // 2, x // 2:(x + ww) // 2], rv)
                and np.array_equal(r["coeff"][k][:4096].reshape(64, 64)[:hh, :ww], cy)
                and np.array_equal(r["coeff"][k][4096:].reshape(2, 32, 32)[:, :hh // 2, :ww // 2], cuv))
          if not ok: bad.append((who, x // 64, yy // 64))
    return bad


def filters_and_coder(W, Hh, depth, qp, t):
    """The other half of the closed loop, oracle against the encoder's records: per-CTU deblocking + SAO decisions + final picture
    (tests/test_oracle_sao_search.py) and the slice data (tests/test_handover.py), from a golden written to /tmp."""
    M.full(W, Hh, depth, qp, t, picture=H.varied_picture, out_dir="/tmp")
    g = np.load(f"/tmp/ref_ctu_{W}x{Hh}_{depth}_qp{qp}.npz")
    _, _, _, _, y, u, v = H.golden_source(g)
    bad = []
    r = H.oracle_sao_picture(ORC, depth, W, Hh, qp, float(g["lam"][0]), (y, u, v), (g["rec_y"], g["rec_u"], g["rec_v"]), H.scu_from_cu(g["cu"], qp))
    for k in ("snap_y", "snap_u", "snap_v", "final_y", "final_u", "final_v", "sao_models"):
        if not np.array_equal(r[k], g[k]): bad.append(k)
    if not np.array_equal(H.sao_info_comparable(r["sao"]), H.sao_info_comparable(g["sao"])): bad.append("sao")
    res = dict(cu=g["cu"], trees=g["trees"], coeff=g["coeff"])
    data, off, after = H.oracle_encode_rows(ORC, depth, H.search_params(W, Hh, qp), res, g["sao"])
    if not (np.array_equal(off, g["row_off"]) and np.array_equal(data, g["row_bytes"]) and np.array_equal(after, g["models"][:, 2])): bad.append("slice data")
    elif g["bitstream"].tobytes().find(data.tobytes()) <= 0: bad.append("slice data not in the .266")
    return bad


def inter_case(W, Hh, depth, qp, frames, sao):
    """A low-delay encode (--gop lp-g4d3t1) of the moving test sequence: every inter CU reconstructed from the encoder's decisions
    (helpers.inter_reconstruct through the oracle's block functions), then the in-loop filters of every picture (deblocking with the
    B-slice rule, and with SAO on the decisions and the output picture)."""
    import zlib
    tag = M.inter(W, Hh, depth, qp, frames, extra=() if sao else ("sao", "off"), suffix="" if sao else "_nosao", out_dir="/tmp")
    g = {k: v for k, v in np.load(f"/tmp/ref_inter_{tag}.npz").items()}
    bad = []
    try:
        seen = H.inter_reconstruct(g, H.OracleBlocks(ORC, depth))
    except AssertionError as e:
        bad.append(("reconstruction", str(e)[:120]))
        seen = {}
    for fr in range(frames):
        ks = [k for k in range(len(g["meta"])) if int(g["meta"][k][0]) == fr]
        meta = g["meta"][ks[0]]
        scu = H.inter_scu_table(g, fr)
        if sao:
            src = H.moving_picture(W, Hh, fr, depth)
            r = H.oracle_sao_picture(ORC, depth, W, Hh, int(meta[7]), float(g["lam"][ks[0]][0]), src, (g["rec_y"][fr], g["rec_u"][fr], g["rec_v"][fr]), scu,
                                     slice_type=int(meta[6]))
            if not np.array_equal(H.sao_info_comparable(r["sao"]), H.sao_info_comparable(np.stack([g["sao"][k] for k in ks]))): bad.append(("sao", fr))
            if not all(np.array_equal(r[k], g[k][fr]) for k in ("final_y", "final_u", "final_v")): bad.append(("final", fr))
        else:
            y, u, v = (np.ascontiguousarray(g[k][fr]).copy() for k in ("rec_y", "rec_u", "rec_v"))
            ORC.deblock_frame(depth, y, u, v, W, Hh, scu.view(np.uint8).reshape(scu.shape[0], -1), scu.shape[1], 0, 0, int(meta[6]) == 0, -1, None)
            if not (np.array_equal(y, g["final_y"][fr]) and np.array_equal(u, g["final_u"][fr]) and np.array_equal(v, g["final_v"][fr])): bad.append(("deblocked", fr))
    return bad, seen


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "inter":
        import random
        rng = random.Random(int(sys.argv[2]))
        fails = 0
        for _ in range(int(sys.argv[1])):
            W, Hh = rng.choice([64, 72, 128, 136, 192, 264]), rng.choice([64, 72, 128, 136])
            depth, qp, frames, sao = rng.choice([8, 10]), rng.choice([12, 17, 22, 27, 32, 37]), rng.choice([3, 4, 6]), rng.choice([True, False])
            bad, seen = inter_case(W, Hh, depth, qp, frames, sao)
            print(f"inter {W}x{Hh} {depth}-bit qp {qp} {frames} frames sao {int(sao)}: {'ok' if not bad else 'DIFFERS: ' + str(bad[:4])} {seen}", flush=True)
            fails += bool(bad)
        print("cases that differ:", fails)
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    fails = 0
    grid = sys.argv[3] if len(sys.argv) > 3 else "default"          # default | small (sides 8..56 too) | large (up to 640x384)
    kw = dict(small=dict(widths=(8, 16, 24, 40, 56, 64, 104), heights=(8, 16, 32, 48, 56, 64, 88)),
              large=dict(widths=(384, 448, 520, 640), heights=(264, 320, 384))).get(grid, {})
    for W, Hh, depth, qp, t in H.sweep_cases(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1, **kw):
        bad = one(W, Hh, depth, qp, t) + filters_and_coder(W, Hh, depth, qp, t)
        print(f"{W}x{Hh} {depth}-bit qp {qp} t {t}: {'ok' if not bad else 'DIFFERS: ' + str(bad[:6])}", flush=True)
        fails += bool(bad)
    print("cases that differ:", fails)
