"""dev: one sweep case through the loop plan vs the oracle chain; prints the SAO records that differ.  usage: sao_case.py W H depth qp t"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import helpers as H
from uvg266_amd import api, lib
lib.init(0)
orc = H.load_oracle()
W, Hh, depth, qp, t = map(int, sys.argv[1:6])
prm = H.search_params(W, Hh, qp)
y, u, v = H.varied_picture(W, Hh, t, depth)
cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
cl.run()
info, models = cl.results()
s = H.oracle_search_picture(orc, depth, prm, y, u, v)
f = H.oracle_sao_picture(orc, depth, W, Hh, qp, prm.lam, (y, u, v), (s["rec_y"], s["rec_u"], s["rec_v"]), H.scu_from_cu(s["cu"], qp))
a, b = H.sao_info_comparable(info[0]), H.sao_info_comparable(f["sao"])
for k in range(a.shape[0]):
    for c in range(2):
        if not np.array_equal(a[k, c], b[k, c]):
            print("CTU", k, "luma" if c == 0 else "chroma", "\n  device", info[0][k, c].tolist(), "\n  oracle", f["sao"][k, c].tolist())
print("sao models equal:", np.array_equal(models[0], f["sao_models"].astype(models.dtype)))
final = [p.cpu().numpy() for p in cl.out[0]]
print("final equal:", [bool(np.array_equal(x, f[k])) for x, k in zip(final, ("final_y", "final_u", "final_v"))])
