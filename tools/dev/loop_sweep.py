"""dev: the loop plan (search -> filters -> SAO -> slice data) on N combinations of the sweep grid against the oracle chain.  usage: loop_sweep.py N seed [small|large]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import helpers as H
from uvg266_amd import api, lib
lib.init(0)
orc = H.load_oracle()
fails = 0
kw = dict(small=dict(widths=(8, 16, 24, 40, 56, 64, 104), heights=(8, 16, 32, 48, 56, 64, 88)), large=dict(widths=(384, 448, 520, 640), heights=(264, 320, 384))).get(sys.argv[3] if len(sys.argv) > 3 else '', {})
for W, Hh, depth, qp, t in H.sweep_cases(int(sys.argv[1]), int(sys.argv[2]), **kw):
    prm = H.search_params(W, Hh, qp)
    y, u, v = H.varied_picture(W, Hh, t, depth)
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cl.run()
    out, nbytes = cl.slice_data()
    info, models = cl.results()
    nb = nbytes.cpu().numpy()[0]
    got = np.concatenate([out[0, r, :nb[r]].cpu().numpy() for r in range(len(nb))])
    final = [p.cpu().numpy() for p in cl.out[0]]
    ry, ru, rv = (p.cpu().numpy() for p in cl.rec[0])
    d = H.search_result_from_device_layout(W, Hh, ry, ru, rv, cl.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP), cl.coeff[0].cpu().numpy(), cl.models[0].cpu().numpy().view(np.uint32))
    s = H.oracle_search_picture(orc, depth, prm, y, u, v)
    f = H.oracle_sao_picture(orc, depth, W, Hh, qp, prm.lam, (y, u, v), (s["rec_y"], s["rec_u"], s["rec_v"]), H.scu_from_cu(s["cu"], qp))
    want, off, _ = H.oracle_encode_rows(orc, depth, prm, s, f["sao"])
    bad = []
    cd, cs = H.ctu_crcs(d, W, Hh), H.ctu_crcs(s, W, Hh)          # (column 1 is the reconstruction: the plan has deblocked it in place by now)
    if not (np.array_equal(cd[:, [0, 2, 3]], cs[:, [0, 2, 3]]) and np.array_equal(d["models"], s["models"])): bad.append("search")
    if not np.array_equal(H.sao_info_comparable(info[0]), H.sao_info_comparable(f["sao"])): bad.append("sao")
    if not all(np.array_equal(a, f[k]) for a, k in zip(final, ("final_y", "final_u", "final_v"))): bad.append("final")
    if not (np.array_equal(np.concatenate([[0], np.cumsum(nb)]), off) and np.array_equal(got, want)): bad.append("slice data")
    if bad: fails += 1; print((W, Hh, depth, qp, t), bad, flush=True)
    del cl
print("cases that differ:", fails, "of", sys.argv[1])
