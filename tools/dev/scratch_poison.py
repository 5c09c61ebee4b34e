"""dev: which field of the per-workgroup global scratch does the search read before writing it?  Poisons one field at a time."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import helpers as H
from uvg266_amd import api, lib
lib.init(0)
g = H.ctu_golden(sys.argv[1])
W, Hh, depth, qp, y, u, v = H.golden_source(g)
prm = H.search_params(W, Hh, qp)
SZ = 59328
F = [("cost_coeff", 0, 8192), ("cost_sig", 8192, 8192), ("cost_coeff0", 16384, 8192), ("save_px", 24576, 12288), ("save_co", 36864, 12288), ("save_cu", 49152, 2048),
     ("cand_co", 51200, 4032), ("save_tree", 55232, 2048), ("tree", 57280, 512), ("mtt", 57792, 512), ("prof", 58304, 1024), ("all", 0, SZ), ("none", 0, 0)]
wc, hc = (W + 63) // 64, (Hh + 63) // 64
total = wc * hc
al = lambda v, a: (v + a - 1) // a * a
done = 512; order = al(done + total * 4, 256); pics = al(order + total * 4, 256); scr = al(pics + 96, 256)
for name, off, ln in F:
    cs = api.CtuSearch(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    n_slots = (cs.ws.numel() - scr) // SZ
    assert scr + n_slots * SZ == cs.ws.numel(), (cs.ws.numel(), scr, n_slots)
    view = cs.ws[scr:].view(n_slots, SZ)
    view.zero_()
    if ln: view[:, off:off + ln] = int(os.environ.get("FILL", "171"))
    torch.cuda.synchronize()
    cs.run(); torch.cuda.synchronize()
    ry, ru, rv = (t.cpu().numpy() for t in cs.rec[0])
    r = H.search_result_from_device_layout(W, Hh, ry, ru, rv, cs.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP), cs.coeff[0].cpu().numpy(), cs.models[0].cpu().numpy().view(np.uint32))
    bad = [(k % wc, k // wc, j) for k in range(len(g["models"])) for j in range(3) if not np.array_equal(r["models"][k, j], g["models"][k, j])]
    print(name, "differs" if bad else "ok", bad[:3])
