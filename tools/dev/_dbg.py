import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from uvg266_amd import api, lib, pipeline
L = lib.init(0)
wl = pipeline.WORKLOADS["test8"]
modes = api.make_modes(pipeline.MODES)
F = 3
grp = pipeline.FrameGroup(L, wl, 5, F, "cuda", modes, step=2)
st = torch.cuda.current_stream().cuda_stream
pipeline.run(grp.all_launches(), st)
torch.cuda.synchronize()
for f in range(F):
    one = pipeline.BandFrame(L, wl, 5 + 2 * f, "cuda", modes)
    pipeline.run(one.all_launches(), st); torch.cuda.synchronize()
    got = grp.frames[f]
    print("f", f, "src equal", torch.equal(got.y, one.y))
    for n in pipeline.SIZES:
        a, b = got.bufs[n]["best"].cpu().numpy(), one.bufs[n]["best"].cpu().numpy()
        ca, cb = got.bufs[n]["cost"].cpu().numpy(), one.bufs[n]["cost"].cpu().numpy()
        bad = np.nonzero((a != b) | (ca != cb))[0]
        print(" n", n, "cnt", len(a), "bad", len(bad), bad[:10], one.own[n][bad[:6]].tolist() if len(bad) else "")
