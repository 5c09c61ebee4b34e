"""dev: the 1080p 8-bit golden through the CTU search N times; which CTUs (and which of their CRCs) ever differ from the reference run."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import helpers as H
import test_gpu_ctu_search as T
from uvg266_amd import lib
hip = lib.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "ref_ctucrc_1920x1080_8_qp22"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = H.ctu_golden(name)
W, Hh, depth, qp, y, u, v = H.golden_source(g)
prm = H.search_params(W, Hh, qp)
wc = (W + 63) // 64
for rep in range(n):
    r = T.run_gpu(hip, depth, prm, [(y, u, v)])[0]
    c = H.ctu_crcs(r, W, Hh)
    bad = np.argwhere((c != g["crc"]).any(axis=1)).ravel()
    print(rep, "bad CTUs:", [(int(b) % wc, int(b) // wc, (c[b] != g["crc"][b]).astype(int).tolist()) for b in bad[:6]], len(bad), flush=True)
