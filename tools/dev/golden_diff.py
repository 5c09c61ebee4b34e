"""dev: where does the device search differ from a ref_ctu golden?  usage: golden_diff.py <golden name>"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import helpers as H
from uvg266_amd import api, lib
if os.environ.get("UVGHIP_LIB"): lib.LIB_PATH = os.path.abspath(os.environ["UVGHIP_LIB"])
lib.init(0)
g = H.ctu_golden(sys.argv[1])
W, Hh, depth, qp, y, u, v = H.golden_source(g)
prm = H.search_params(W, Hh, qp)
FILL = int(os.environ.get('FILL', '-1'))
for rep in range(int(os.environ.get("REPS", "3"))):
    if FILL >= 0:
        nb = lib.load_library().uvghip_ctu_search_workspace_bytes(1, W, Hh)
        junk = torch.full((nb,), FILL, dtype=torch.uint8, device='cuda'); torch.cuda.synchronize(); del junk
    cs = api.CtuSearch(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cs.run(); torch.cuda.synchronize()
    ry, ru, rv = (t.cpu().numpy() for t in cs.rec[0])
    r = H.search_result_from_device_layout(W, Hh, ry, ru, rv, cs.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP), cs.coeff[0].cpu().numpy(), cs.models[0].cpu().numpy().view(np.uint32))
    wc = (W + 63) // 64
    bad = [(k % wc, k // wc, j) for k in range(len(g["models"])) for j in range(3) if not np.array_equal(r["models"][k, j], g["models"][k, j])]
    print("rep", rep, "model sets that differ (cx, cy, set):", bad[:6])
    d = np.argwhere((r["cu"][:Hh // 4, :W // 4] != g["cu"][:Hh // 4, :W // 4]).any(axis=2)); print(" cu units differ:", len(d), d[:4].tolist())
    for p in ("rec_y", "rec_u", "rec_v"):
        dd = np.argwhere(r[p] != g[p]); print(" ", p, len(dd), dd[:3].tolist())
    c = np.argwhere(r["coeff"] != g["coeff"]); print("  coeff", len(c), c[:4].tolist())
    if bad:
        k = bad[0][1] * wc + bad[0][0]; j = bad[0][2]
        a, b = r["models"][k, j], g["models"][k, j]
        idx = np.argwhere(a != b).ravel(); print("  first differing bytes of the model set:", idx[:10].tolist(), a[idx[:6]].tolist(), b[idx[:6]].tolist())
    if bad:
        cx, cy = bad[0][0], bad[0][1]
        for nm, src in (("device", r), ("golden", g)):
            blk = src["cu"][cy * 16:cy * 16 + 16, cx * 16:cx * 16 + 16]
            print("  ", nm, "log2_width / mode (low byte) / cbf of the CTU's 4x4 units:")
            for row in range(16):
                print("    ", " ".join("%d:%02d:%d" % (blk[row, c, 1], blk[row, c, 6], blk[row, c, 5]) for c in range(16)))
        break
