import sys, json
sys.path.insert(0, ".")
import bench, torch
from uvg266_amd import lib
lib.init(0)
dev = torch.device("cuda:0")
grp = bench.ClosedLoop(bench.WORKLOADS["2160p10alf"], 0, 16, dev)
grp.issue(False); torch.cuda.synchronize()
print(json.dumps(bench.alf_stage_timing(grp, reps=1))[:400])
