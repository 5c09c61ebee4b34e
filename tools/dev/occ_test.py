import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from uvg266_amd import api, layout, lib
lib.init(0)
for (W,H,depth,n) in ((3840,2160,10,40),(3840,2160,10,80),(3840,2160,8,40),(1920,1080,10,160)):
    P = api.ctu_params(W,H,22)
    base = tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W,H,0,depth))
    src = [tuple(t.clone() for t in base) for _ in range(n)]
    cs = api.CtuSearch(P, src)
    cs.run(); torch.cuda.synchronize()
    t=time.time(); cs.run(); torch.cuda.synchronize(); dt=time.time()-t
    ctus=((W+63)//64)*((H+63)//64)*n
    print(W,H,depth,n, f"{n/dt:.2f} pictures/s {ctus/dt:.0f} CTU/s")
    del cs, src
