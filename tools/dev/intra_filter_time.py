"""Development (UVGHIP_LIB = a -DCTU_PROFILE build): cycles per CTU of the in-kernel filter stage of the all-intra loop."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uvg266_amd import api, layout
W, H, depth, n = 1920, 1080, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 32
src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W, H, t, depth)) for t in range(n)]
cl = api.ClosedLoop(api.ctu_params(W, H, 22), src)
cl.run(); torch.cuda.synchronize()
t0 = time.perf_counter(); cl.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
pf = cl.loop_ws[16:32].cpu().numpy().view(np.uint64)
print(f"{n} pictures {dt * 1e3:.1f} ms", None if pf is None else f"filter stage: {pf[0] / max(1, pf[1]):.0f} cycles per CTU over {pf[1]} CTUs")
