import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from uvg266_amd import api, layout, lib
lib.init(0)
W,H,depth,n = 1920,1080,8,64
P = api.ctu_params(W,H,22)
src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W,H,t,depth)) for t in range(n)]
cl = api.ClosedLoop(P, src)
cl.run(); torch.cuda.synchronize()
for rep in range(2):
    t=time.time(); out, nb = cl.encode_rows(); torch.cuda.synchronize(); dt=time.time()-t
    print(f"slice rows of {n} pictures: {dt*1e3:.1f} ms = {dt*1e3/n:.2f} ms per picture; bytes per picture {int(nb.sum())//n}")
