"""dev: wall-clock phases of bench.py's c2_clip (upload / run / NAL assembly), synchronising between them."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from uvg266_amd import lib, api, layout
hip = lib.init(0)
W, H, depth, n = 1920, 1080, 8, 60
if len(sys.argv) > 4: W, H, depth, n = (int(a) for a in sys.argv[1:5])
P = api.ctu_params(W, H, 22)
host = [tuple(np.ascontiguousarray(p) for p in layout.synthetic_yuv420(W, H, t, depth)) for t in range(n)]
src = [tuple(torch.empty(p.shape, dtype=torch.uint8 if depth == 8 else torch.uint16, device="cuda") for p in yuv) for yuv in host]
cs = api.ClosedLoop(P, src)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for yuv, dst in zip(host, src):
        for p, d in zip(yuv, dst):
            d.copy_(torch.from_numpy(p), non_blocking=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    cs.run()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out = [cs.picture_nals(i, i) for i in range(n)]
    t3 = time.perf_counter()
    print(f"upload {1e3*(t1-t0):.1f} ms, run {1e3*(t2-t1):.1f} ms, nals {1e3*(t3-t2):.1f} ms, total {1e3*(t3-t0):.1f} ms -> {n/(t3-t0):.1f} frames/s; {sum(len(b) for b in out)} bytes", flush=True)
    if hasattr(cs, "group_nals"):
        t4 = time.perf_counter(); g = cs.group_nals(0); t5 = time.perf_counter()
        print(f"   group_nals {1e3*(t5-t4):.1f} ms, equal: {b''.join(out) == b''.join(g)}", flush=True)
