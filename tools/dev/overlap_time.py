#!/usr/bin/env python3
"""Dev-time: uvghip_loop_plan_run against uvghip_loop_plan_run_overlapped (filters and coder beside the search) for one picture and for
configs[1]'s 60-picture clip, device time + the group's NAL units on the host.   python tools/dev/overlap_time.py [1080|2160]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from uvg266_amd import api, layout

big = len(sys.argv) > 1 and sys.argv[1] == "2160"
W, Hh, depth, qp = (3840, 2160, 10, 22) if big else (1920, 1080, 8, 22)
prm = H.search_params(W, Hh, qp)
for n in (1, 8, 30) if big else (1, 16, 60):
    src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W, Hh, t % 7, depth)) for t in range(n)]
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src)
    ref = None
    for name, fn in (("run", cl.run), ("run_overlapped", cl.run_overlapped)):
        fn(); nals = cl.group_nals(); torch.cuda.synchronize()
        ref = nals if ref is None else ref
        assert nals == ref, "the two runs differ"
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cl.group_nals()
            t2 = time.perf_counter()
            best = min(best, t2 - t0)
        print(f"{W}x{Hh} {depth}-bit {n:3d} picture(s) {name:15s}: {1e3 * best:8.1f} ms ({n / best:6.1f} pictures/s; last: device {1e3 * (t1 - t0):.1f} + NAL units {1e3 * (t2 - t1):.1f} ms)", flush=True)
    del cl
