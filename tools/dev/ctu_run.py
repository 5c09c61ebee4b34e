"""dev: run the closed-loop CTU search on synthetic pictures (the workload the PC-sampling / PMC scripts profile).
usage: ctu_run.py W H depth n_pictures [repeats]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from uvg266_amd import lib, api, layout
lib.init(0)
W, H, depth, n = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
P = api.ctu_params(W, H, 22)
src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W, H, t, depth)) for t in range(n)]
cs = api.CtuSearch(P, src)
for rep in range(reps):
    torch.cuda.synchronize(); t = time.time(); cs.run(); torch.cuda.synchronize(); dt = time.time() - t
    print(f"{n} pictures {W}x{H} {depth}-bit: {dt*1e3:.1f} ms -> {n/dt:.2f} pictures/s", flush=True)
del cs
