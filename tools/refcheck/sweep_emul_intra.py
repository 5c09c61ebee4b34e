#!/usr/bin/env python3
"""Dev-time: N random small pictures (tests/helpers.sweep_cases: sizes with 8-sample CTU edges, both bit depths, QP 0..51, five kinds of
content) through the I-picture CTU kernel's HOST EMULATION (tests/emul: csrc/ctu_core.h with one lane) against the oracle, which
tools/refcheck/sweep_ctu.py holds to the real encoder on the same grid.  Counts the CTUs the oracle codes as one 64x64 CU (the
candidate of combine_intra_cus: ctu_core.h post64 / finish64).  Nothing is kept.
    python tools/refcheck/sweep_emul_intra.py N seed"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import helpers as H

n, seed = int(sys.argv[1]), int(sys.argv[2])
orc = H.load_oracle()
bad = wins = 0
t0 = time.time()
for W, Hh, depth, qp, t in H.sweep_cases(n, seed):
    prm = H.search_params(W, Hh, qp)
    pic = H.varied_picture(W, Hh, t, depth)
    r = H.emul_search_picture(depth, prm, *pic)
    o = H.oracle_search_picture(orc, depth, prm, *pic)
    ok = np.array_equal(H.ctu_crcs(r, W, Hh), H.ctu_crcs(o, W, Hh)) and np.array_equal(r["models"], o["models"])
    wins += int((o["cu"][:Hh // 4:16, :W // 4:16, 1] == 6).sum())
    if not ok:
        bad += 1
        print("MISMATCH", W, Hh, depth, qp, t, flush=True)
print(f"{n} cases, {bad} mismatches, {wins} CTUs coded as 64x64 CUs, {time.time() - t0:.0f} s")
