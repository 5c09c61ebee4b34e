"""Development: per-CTU timestamps of an in-flight run (UVGHIP_LIB=uvg266_amd/libuvg266hip_prof.so, built with EXTRA=-DCTU_PROFILE) ->
gpurun_out/flight_times.npz [picture][ctu][4] (s_memtime ticks at 100 MHz: ticket, search start, search end, filters end)."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from uvg266_amd import api, lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = np.load(os.path.join(ROOT, "tests", "golden", "ref_intercrc_1920x1080_8_qp27_120frames_owf1.npz"))
W, H, depth, qp, total = (int(a) for a in g["dims"])
states = Hh.frame_states_from_records(g["meta"], g["lam"], g["refs"])
pics = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in Hh.clip_picture(W, H, t, depth)) for t in range(n)]
loop = api.LowDelayLoop(W, H, depth, 1, states[:n], [pics], inflight=True, inflight_margin=11)
loop.run(); torch.cuda.synchronize()
t0 = time.perf_counter(); loop.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{n} pictures: {dt * 1e3:.1f} ms")
L = lib.init(0)
L.uvghip_ctu_search_pb_debug_times.restype = ctypes.c_size_t
fr, step = loop.order[-1]
ws = step[2]
npb = len(fr)
ctus = ((W + 63) // 64) * ((H + 63) // 64)
off = L.uvghip_ctu_search_pb_debug_times(npb, W, H)
t = ws[off:off + npb * ctus * 32].cpu().numpy().view(np.uint64).reshape(npb, ctus, 4).astype(np.int64)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "flight_times.npz"), t=t, frames=np.array(fr), qp=np.array([states[f]["qp"] for f in fr]))
s = (t[:, :, 2] - t[:, :, 1]) / 1e5          # 100 MHz wall clock -> ms
f = (t[:, :, 3] - t[:, :, 2]) / 1e5
w = (t[:, :, 1] - t[:, :, 0]) / 1e5
qps = np.array([states[f_]["qp"] for f_ in fr])
for q in sorted(set(qps.tolist())):
    m = s[qps == q]
    print("qp %d: %d pictures, CTU search ms mean %.2f median %.2f p90 %.2f max %.2f" % (q, (qps == q).sum(), m.mean(), np.median(m), np.percentile(m, 90), m.max()))
print("search ms: mean %.2f median %.2f p90 %.2f max %.2f | filters ms: mean %.3f max %.3f | wait ms: mean %.1f" % (s.mean(), np.median(s), np.percentile(s, 90), s.max(), f.mean(), f.max(), w.mean()))
st = t[:, 0, 1]
print("picture start-to-start ms:", np.round(np.diff(st) / 1e5, 1).tolist()[:40])
