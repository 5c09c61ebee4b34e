#!/usr/bin/env python3
"""Dev-time: a low-delay encode of the REAL reference (oracle/_ref through ctu_dump.sh, CTU_DUMP_CU_INTER=1) against the oracle's inter
search (orcN_search_inter_picture), picture by picture and CTU by CTU: uvg_search_cu_inter's calls (costs and decided motion), the
side information, motion, levels, reconstruction and the three model sets.

    python tools/refcheck/inter_check.py W H depth qp frames [first_t] [option value]...
"""
import os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as Hh
from make_ctu_goldens import read_records


def sequence_picture(W, H, depth, t0, t, kind):
    """moving: helpers.moving_picture (quarter-sample motion, two halves); fast: three steps per picture (vectors of 2-5 samples, blocks
    leaving the picture); noise: moving + fresh noise of +-6 per picture (residuals everywhere); still: one picture repeated (skips)."""
    if kind == "fast":
        return Hh.moving_picture(W, H, (t0 + 3 * t) % 15, depth)
    if kind == "still":
        return Hh.moving_picture(W, H, t0, depth)
    p = Hh.moving_picture(W, H, t0 + t, depth)
    if kind == "noise":
        rng = np.random.default_rng(1000 * t0 + t)
        s = 1 if depth == 8 else 4
        p = tuple(np.clip(a.astype(np.int32) + rng.integers(-6 * s, 6 * s + 1, a.shape), 0, (1 << depth) - 1).astype(a.dtype) for a in p)
    return p


def encode(W, H, depth, qp, frames, t0=0, extra=(), kind="moving"):
    px = np.uint8 if depth == 8 else np.uint16
    tag = f"ic_{W}x{H}_{depth}_{qp}_{frames}_{t0}_{os.getpid()}"
    yuv = f"/tmp/{tag}.yuv"
    pics = [sequence_picture(W, H, depth, t0, t, kind) for t in range(frames)]
    with open(yuv, "wb") as f:
        for p3 in pics:
            for p in p3:
                f.write(p.astype(px).tobytes())
    out = f"/tmp/{tag}"
    subprocess.check_call([os.path.join(ROOT, "tools/refcheck/ctu_dump.sh"), str(depth), yuv, str(W), str(H), str(frames), out,
                           "preset", "medium", "gop", "lp-g4d3t1", "qp", str(qp)] + list(extra), stderr=subprocess.DEVNULL,
                          env=dict(os.environ, CTU_DUMP_CU_INTER="1"))
    recs = read_records(out + ".bin")
    for e in (".bin", ".266"):
        os.remove(out + e)
    os.remove(yuv)
    return pics, recs


def pictures_from_records(W, H, depth, frames, recs):
    """-> per frame number: dict(meta, lam, refs, cu [h16, w16, 12], trees, motion [h16, w16, 8], rec, coeff [ctus, 6144], models [ctus, 3, ..],
    models_inter [ctus, 3, 90], final (y, u, v), cuinter [(ints, doubles)])"""
    px = Hh.px_dtype(depth)
    wc, hc = (W + 63) // 64, (H + 63) // 64
    P = {}

    def pic(fr):
        return P.setdefault(fr, dict(cu=np.zeros((hc * 16, wc * 16, 12), np.uint8), trees=np.zeros((hc * 16, wc * 16, 2), np.uint32),
                                     motion=np.zeros((hc * 16, wc * 16, 8), np.int32), rec=[np.zeros((H, W), px), np.zeros((H // 2, W // 2), px), np.zeros((H // 2, W // 2), px)],
                                     coeff=np.zeros((wc * hc, 6144), np.int16), models=np.zeros((wc * hc, 3, Hh.MODELS_BYTES), np.uint8),
                                     models_inter=np.zeros((wc * hc, 3, Hh.MODELS_INTER_BYTES), np.uint8), cuinter=[]))
    coded = {(int(r[0][0]), int(r[0][1]), int(r[0][2])): r for n, r in recs if n == "coded"}
    for n, r in recs:
        if n == "search":
            fr, x, y = int(r[0][0]), int(r[0][1]), int(r[0][2])
            d = pic(fr)
            d["meta"], d["lam"], d["refs"] = r[0], r[1], r[12]
            k = (y // 64) * wc + x // 64
            hh, ww = min(64, H - y), min(64, W - x)
            d["cu"][y // 4:y // 4 + 16, x // 4:x // 4 + 16] = r[4].reshape(16, 16, 12)
            d["trees"][y // 4:y // 4 + 16, x // 4:x // 4 + 16] = r[5].reshape(16, 16, 2)
            d["motion"][y // 4:y // 4 + 16, x // 4:x // 4 + 16] = r[11].reshape(16, 16, 8)
            d["rec"][0][y:y + hh, x:x + ww] = r[6].reshape(64, 64)[:hh, :ww]
            d["rec"][1][y // 2:(y + hh) // 2, x // 2:(x + ww) // 2] = r[7].reshape(32, 32)[:hh // 2, :ww // 2]
            d["rec"][2][y // 2:(y + hh) // 2, x // 2:(x + ww) // 2] = r[8].reshape(32, 32)[:hh // 2, :ww // 2]
            d["coeff"][k, :4096] = r[9]; d["coeff"][k, 4096:] = r[10]
            d["models"][k, 0] = r[2]; d["models"][k, 1] = r[3]
            d["models_inter"][k, 0] = r[13]; d["models_inter"][k, 1] = r[14]
            c = coded[(fr, x, y)]
            d["models"][k, 2] = c[2]; d["models_inter"][k, 2] = c[7]
        elif n == "cuinter":
            pic(int(r[0][0]))["cuinter"].append((r[0].copy(), r[1].copy()))
        elif n == "final":
            pic(int(r[0][0]))["final"] = tuple(r[1 + c].reshape(H >> (c > 0), W >> (c > 0)).copy() for c in range(3))
    return P


def check(W, H, depth, qp, frames, t0=0, extra=(), verbose=True, kind="moving"):
    orc = Hh.load_oracle()
    pics, recs = encode(W, H, depth, qp, frames, t0, extra, kind)
    P = pictures_from_records(W, H, depth, frames, recs)
    opt = dict(zip(extra[0::2], extra[1::2]))            # the tools of the run, for the oracle's frame state (helpers.iter_inter_frames)
    cfg = (int(opt.get("tmvp", 1)), int(opt.get("max-merge", 6)), 2, int(opt.get("bipred", 1)), int(opt.get("subme", 4)), int(opt.get("early-skip", 1)))
    for d in P.values():
        d["cfg"] = cfg
    wc, hc = (W + 63) // 64, (H + 63) // 64
    bad = 0
    for fr, d, r, buf, ntr in Hh.run_inter_oracle(orc, W, H, depth, pics, P):
        poc, slice_type, ref_pocs, l0, l1 = d["info"]
        msgs = Hh.compare_inter_picture(W, H, d, r, buf, ntr)
        if verbose or msgs:
            print(f"frame {fr} poc {poc} slice {slice_type} qp {int(d['meta'][3])} refs {ref_pocs} L0 {l0} L1 {l1} "
                  f"inter calls {len(d['cuinter'])}: {'OK' if not msgs else ''}")
        for m in msgs:
            print("   ", m)
        bad += bool(msgs)
    if os.environ.get("CHECK_KERNEL_SOURCE"):      # ... and the device kernel's source built for the host (tests/emul) against the same records
        for fr, d, prm, F, keep in Hh.iter_inter_frames(W, H, P):
            if int(d["meta"][6]) == 2:
                continue
            msgs = Hh.compare_device_inter_picture(W, H, d, Hh.emul_search_inter_picture(depth, prm, F, *pics[fr]))
            if verbose or msgs:
                print(f"frame {fr} kernel source on the host: {'OK' if not msgs else msgs[:3]}")
            bad += bool(msgs)
    return bad


if __name__ == "__main__":
    a = sys.argv[1:]
    W, H, depth, qp, frames = (int(v) for v in a[:5])
    t0 = int(a[5]) if len(a) > 5 else 0
    sys.exit(1 if check(W, H, depth, qp, frames, t0, tuple(a[6:])) else 0)
