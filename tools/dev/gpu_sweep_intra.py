"""dev: N random small pictures (tests/helpers.sweep_cases) through the I-picture CTU search kernel on the GPU against the oracle (per-CTU
CRCs of side information / reconstruction / levels, all three model sets), and a few 1080p pictures several times over (the cross-wave
hand-overs of the depth pipeline and of the 64x64 candidate under load).  usage: gpu_sweep_intra.py N seed [reps_1080p]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import helpers as H
import test_gpu_ctu_search as T
from uvg266_amd import lib, layout
hip = lib.init(0)
orc = H.load_oracle()
n, seed = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = wins = 0
t0 = time.time()
for W, Hh, depth, qp, t in H.sweep_cases(n, seed):
    prm = H.search_params(W, Hh, qp)
    pic = H.varied_picture(W, Hh, t, depth)
    r = T.run_gpu(hip, depth, prm, [pic])[0]
    o = H.oracle_search_picture(orc, depth, prm, *pic)
    ok = np.array_equal(H.ctu_crcs(r, W, Hh), H.ctu_crcs(o, W, Hh)) and np.array_equal(r["models"], o["models"])
    wins += int((o["cu"][:Hh // 4:16, :W // 4:16, 1] == 6).sum())
    if not ok:
        bad += 1
        print("MISMATCH", W, Hh, depth, qp, t, flush=True)
print(f"{n} small cases, {bad} mismatches, {wins} CTUs coded as 64x64 CUs, {time.time() - t0:.0f} s", flush=True)
if reps:
    for name in ("ref_ctucrc_1920x1080_8_qp22", "ref_ctucrc_1920x1080_10_qp27"):
        g = H.ctu_golden(name)
        W, Hh, depth, qp, y, u, v = H.golden_source(g)
        prm = H.search_params(W, Hh, qp)
        nb = 0
        for rep in range(reps):
            rs = T.run_gpu(hip, depth, prm, [(y, u, v)] * 8)          # eight copies in one launch: 4080 CTUs in flight
            for r in rs:
                nb += int((H.ctu_crcs(r, W, Hh) != g["crc"]).any(axis=1).sum())
        print(f"{name}: {reps} x 8 pictures, {nb} CTUs differ from the reference run", flush=True)
