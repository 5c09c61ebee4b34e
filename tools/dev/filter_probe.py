"""Developer probe: the in-loop filter kernels of the 2160p10alf workload alone (deblock, SAO, ALF),
timed one by one."""
import sys, torch
sys.path.insert(0, '/root/repo')
from uvg266_amd import api, lib, pipeline
L = lib.init(0)
wl = pipeline.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "2160p10alf"]
fr = pipeline.BandFrame(L, wl, 0, "cuda", api.make_modes(pipeline.MODES))
st = torch.cuda.current_stream().cuda_stream
pipeline.run(fr.all_launches(), st)
torch.cuda.synchronize()
for l in fr.stage_a + fr.stage_b + fr.stage_c:
    for _ in range(2):
        pipeline.run([l], st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pipeline.run([l], st)
    e1.record(); torch.cuda.synchronize()
    print("%-18s %.1f us" % (l[0], 1e3 * e0.elapsed_time(e1) / 5))
pres = fr.alf_present.cpu().numpy()
print("classes present per CTU: mean %.1f" % float(sum(bin(int(m) & 0xffffffff).count("1") for m in pres) / len(pres)))
