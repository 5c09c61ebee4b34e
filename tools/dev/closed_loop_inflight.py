import sys
sys.path.insert(0,'/root/repo')
import bench, torch
from uvg266_amd import lib, pipeline
L=lib.init(0)
wl=pipeline.WORKLOADS["1080p8"]
for k in (8,16):
    r=bench.closed_loop_probe(L, wl, "cuda", reps=2, in_flight=k)
    print(k, r['frames_per_s_in_flight'], r['frames_per_s'])
