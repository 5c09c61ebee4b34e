"""dev: the slice coder alone (uvghip_loop_plan_run_coder) on n pictures after one closed-loop run; ms per launch.
usage: coder_time.py W H depth n [reps]   (UVGHIP_LIB selects an experimental build)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from uvg266_amd import lib, api, layout
hip = lib.init(0)
W, H, depth, n = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
P = api.ctu_params(W, H, 22)
src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W, H, t, depth)) for t in range(n)]
cl = api.ClosedLoop(P, src)
cl.run()
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
for r in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    lib.check(cl.L.uvghip_loop_plan_run_coder(cl.loop, st), "run_coder")
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    rows, nb = cl.slice_data()
    print(f"{n} pictures {W}x{H} {depth}-bit: coder {dt * 1e3:.1f} ms, {int(nb.sum().item())} bytes of slice data", flush=True)
