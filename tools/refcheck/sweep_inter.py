#!/usr/bin/env python3
"""Dev-time: N low-delay encodes of the real reference (sizes, depths, QPs, picture counts, content offsets drawn from `seed`) against
the oracle's inter search (tools/refcheck/inter_check.py).  Nothing is written.   python tools/refcheck/sweep_inter.py N seed [procs] [tools]
("tools": bi-prediction, the temporal candidate, the fractional search, early skip on / off and 5 / 6 merge candidates drawn per encode)"""
import os, sys
import numpy as np
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def one(case):
    import inter_check, io, contextlib
    W, H, depth, qp, frames, t0, kind, extra = case
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bad = inter_check.check(W, H, depth, qp, frames, t0, extra=extra, verbose=False, kind=kind)
    return case, bad, buf.getvalue()


if __name__ == "__main__":
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        W = int(rng.choice([64, 72, 128, 136, 192, 200, 256, 264, 320]))
        H = int(rng.choice([64, 72, 96, 128, 136, 192]))
        extra = ()
        if len(sys.argv) > 4 and sys.argv[4] == "tools":          # other tools than the preset's, drawn per encode
            for name, values in (("bipred", (0, 1)), ("tmvp", (0, 1)), ("subme", (0, 4)), ("early-skip", (0, 1)), ("max-merge", (5, 6))):
                extra += (name, str(int(rng.choice(values))))
        cases.append((W, H, int(rng.choice([8, 10])), int(rng.integers(10, 45)), int(rng.integers(2, 11)), int(rng.integers(0, 5)), str(rng.choice(['moving', 'fast', 'noise', 'still'])), extra))
    bad = 0
    with ProcessPoolExecutor(procs) as ex:
        for case, b, out in ex.map(one, cases):
            print(case, "OK" if not b else "MISMATCH", flush=True)
            if b:
                print(out)
                bad += 1
    print(f"{n} encodes, {bad} with a mismatch")
    sys.exit(1 if bad else 0)
