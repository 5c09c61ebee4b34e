"""Development probe: wall time of the in-flight low-delay loop over prefixes of the 120-picture clip (the increment per picture = the lag
between dependent pictures), and of the I pictures alone."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from uvg266_amd import api

g = np.load(os.path.join(ROOT, "tests", "golden", "ref_intercrc_1920x1080_8_qp27_120frames_owf1.npz"))
W, H, depth, qp, total = (int(a) for a in g["dims"])
states = Hh.frame_states_from_records(g["meta"], g["lam"], g["refs"])
pics = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in Hh.clip_picture(W, H, t, depth)) for t in range(64)]
for n in [int(a) for a in (sys.argv[1:] or "1 2 3 5 9 17 33 64".split())]:
    loop = api.LowDelayLoop(W, H, depth, 1, states[:n], [pics[:n]], inflight=True, inflight_margin=11)
    loop.run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n} pictures: {dt * 1e3:.1f} ms", flush=True)
    del loop
