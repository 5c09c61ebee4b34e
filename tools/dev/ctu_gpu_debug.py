"""dev: run the CTU search on the GPU against a full golden and list every mismatching item per CTU."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, helpers as H, torch, time
from uvg266_amd import lib
import test_gpu_ctu_search as T
hip = lib.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "ref_ctu_832x480_8_qp22"
g = H.ctu_golden(name)
W, Hh, depth, qp, y, u, v = H.golden_source(g)
prm = H.search_params(W, Hh, qp)
for rep in range(2):
    t = time.time(); r = T.run_gpu(hip, depth, prm, [(y, u, v)])[0]; print("gpu run", time.time() - t)
    wc = (W + 63) // 64
    nb = 0
    for k in range(len(g["models"])):
        oks = [np.array_equal(r["models"][k, j], g["models"][k, j]) for j in range(3)]
        cx, cy = k % wc, k // wc; x, yy = cx * 64, cy * 64; hh, ww = min(64, Hh - yy), min(64, W - x)
        sl = (slice(yy // 4, (yy + hh) // 4), slice(x // 4, (x + ww) // 4))
        okc = np.array_equal(r["cu"][sl], g["cu"][sl]); okt = np.array_equal(r["trees"][sl], g["trees"][sl])
        okr = np.array_equal(r["rec_y"][yy:yy + hh, x:x + ww], g["rec_y"][yy:yy + hh, x:x + ww])
        okco = np.array_equal(r["coeff"][k, :4096].reshape(64, 64)[:hh, :ww], g["coeff"][k, :4096].reshape(64, 64)[:hh, :ww])
        if not (all(oks) and okc and okt and okr and okco):
            nb += 1
            if nb <= 4:
                print("CTU", cx, cy, "models", oks, "cu", okc, "trees", okt, "rec", okr, "coef", okco)
                a = r["models"][k, 2][:1028].view(np.uint16).reshape(2, 257); b = g["models"][k, 2][:1028].view(np.uint16).reshape(2, 257)
                print("  coder models differing:", np.argwhere((a != b).any(axis=0)).ravel().tolist()[:20])
                print("  log2 map\n", g["cu"][sl][:, :, 1])
    print("bad", nb, "of", len(g["models"]))
