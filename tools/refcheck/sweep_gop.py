#!/usr/bin/env python3
"""Dev-time: N encodes of the real reference over GOP structures and the round-5 options -- random access (--gop 8 / 16, with and without
a short intra period: CRA / RASL pictures), low delay, --owf 0 / 1, rd 0 / 1, sizes, depths, QPs, three kinds of content -- through
make_ctu_goldens.inter (records into a scratch directory) and the oracle's chain, as tests/test_oracle_inter_search.py does for the
committed goldens; SWEEP_EMUL=1: also the P / B kernel's host emulation (tests/emul) against the same records.  Nothing is kept.
    python tools/refcheck/sweep_gop.py N seed [procs]"""
import os, sys, tempfile
import numpy as np
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def one(case):
    import io, contextlib
    import make_ctu_goldens as M
    import helpers as H
    W, Hh, depth, qp, frames, extra, clip = case
    tmp = tempfile.mkdtemp()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            tag = M.inter(W, Hh, depth, qp, frames, extra=extra, suffix="_s", out_dir=tmp, clip=clip)
        g = np.load(os.path.join(tmp, f"ref_inter_{tag}.npz"))
        orc = H.load_oracle()
        W2, H2, d2, pics, P = H.inter_pictures_from_golden(g)
        calls = 0
        for fr, d, r, buf, ntr in H.run_inter_oracle(orc, W2, H2, d2, pics, P):
            msgs = H.compare_inter_picture(W2, H2, d, r, buf, ntr)
            if msgs:
                return case, f"picture {fr}: {msgs[:2]}"
            calls += ntr
        if os.environ.get("SWEEP_EMUL"):          # ... and the P / B kernel's source on the host (tests/emul), against the same records
            for fr, d, prm, F, keep in H.iter_inter_frames(W2, H2, P):
                if int(d["meta"][6]) == 2:
                    continue
                got = H.emul_search_inter_picture(d2, prm, F, *pics[fr])
                msgs = H.compare_device_inter_picture(W2, H2, d, got)
                if msgs:
                    return case, f"kernel emulation, picture {fr}: {msgs[:2]}"
        return case, None
    except Exception as e:          # noqa: BLE001
        return case, f"{type(e).__name__}: {e}"
    finally:
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)


if __name__ == "__main__":
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        W = int(rng.choice([64, 72, 136, 192, 200, 264]))
        Hh = int(rng.choice([64, 72, 136, 200]))
        gop = str(rng.choice(["lp", "8", "16", "16"]))
        extra = () if gop == "lp" else ("gop", gop)
        frames = int(rng.integers(3, 9)) if gop == "lp" else int(gop) + 1 + int(rng.integers(0, int(gop) + 1))
        if gop != "lp" and rng.random() < 0.4:
            extra += ("period", gop)          # a second intra period: CRA + RASL pictures
        if rng.random() < 0.4:
            extra += ("owf", "1")
        if rng.random() < 0.3:
            extra += ("rd", "1")
        clip = int(rng.choice([1, 1, 2, 3]))
        if clip == 3:
            frames = min(frames, 12)
        cases.append((W, Hh, int(rng.choice([8, 10])), int(rng.integers(17, 40)), frames, extra, clip))
    bad = 0
    with ProcessPoolExecutor(procs) as ex:
        for case, err in ex.map(one, cases):
            print(case, "OK" if err is None else "MISMATCH " + err, flush=True)
            bad += err is not None
    print(f"{n} encodes, {bad} with a mismatch")
    sys.exit(1 if bad else 0)
