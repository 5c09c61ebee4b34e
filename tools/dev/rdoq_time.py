import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from uvg266_amd import api, lib, pipeline
L = lib.init(0)
rng = np.random.default_rng(1)
ctx = pipeline.synthetic_rdoq_ctx()
for (w, n) in ((4, 129600), (8, 32400), (16, 8100), (32, 2025)):
    coef = torch.from_numpy((rng.normal(0, 40, (n, w, w))).astype(np.int16)).cuda()
    for _ in range(2):
        api.rdoq_batch(coef, 8, 0, 1, 0, 0, 0, 22, 5.7, ctx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ws = torch.empty(64, dtype=torch.float64, device="cuda")
    e0.record()
    for _ in range(5):
        lv, s, h = api.rdoq_batch(coef, 8, 0, 1, 0, 0, 0, 22, 5.7, ctx, ws)
    e1.record(); torch.cuda.synchronize()
    print(w, n, "%.3f ms" % (e0.elapsed_time(e1) / 5), "nz frac %.2f" % float((lv != 0).float().mean()))
