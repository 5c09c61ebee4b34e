"""Developer probe: the RDOQ launches of the bench (1080p8, F pictures per group) on the real transformed residuals,
timed one by one; level statistics of what they produce."""
import sys, torch
sys.path.insert(0, '/root/repo')
from uvg266_amd import api, lib, pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = lib.init(0)
wl = pipeline.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "1080p8"]
g = pipeline.FrameGroup(L, wl, 0, F, "cuda", api.make_modes(pipeline.MODES))
st = torch.cuda.current_stream().cuda_stream
pipeline.run(g.searches() + g.heads_rest(), st)
torch.cuda.synchronize()
for k in range(len(g.mid)):
    name, fn, args = g.mid[k]
    if not name.startswith("rdoq"):
        continue
    for _ in range(2):
        pipeline.run([g.mid[k]], st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pipeline.run([g.mid[k]], st)
    e1.record(); torch.cuda.synchronize()
    key = [kk for kk in g.pool.jobs if (f"{kk[0]}" if kk[1] == 0 else f"chroma_{kk[0]}") == name[5:]]
    stats = ""
    for kk in key[:1]:
        lev = g.pool.jobs[kk]["lev"].abs().float()
        stats = "nz %.3f  >=2 %.3f  mean|l| %.2f  n %d x %d" % (float((lev > 0).float().mean()), float((lev >= 2).float().mean()), float(lev.mean()),
                                                                 lev.shape[0] * lev.shape[1], lev.shape[2])
    print("%-16s %.3f ms  %s" % (name, e0.elapsed_time(e1) / 5, stats))
