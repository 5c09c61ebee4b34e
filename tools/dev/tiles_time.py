#!/usr/bin/env python3
"""Dev-time: what tiles buy on the device -- the time of ONE 1080p / 2160p all-intra picture (search + filters + slice data + NAL units
on the host) and of a 60-picture group, without tiles and under a few grids.  Tiles change the stream (the reference's with the same
--tiles); the comparison is of latency, not of one stream's rate.

  python tools/dev/tiles_time.py [1080|2160]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from uvg266_amd import api, layout


def main():
    big = len(sys.argv) > 1 and sys.argv[1] == "2160"
    W, Hh, depth, qp = (3840, 2160, 10, 22) if big else (1920, 1080, 8, 22)
    prm = H.search_params(W, Hh, qp)
    for n in (1, 8 if big else 60):
        src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W, Hh, t % 7, depth)) for t in range(n)]
        for grid in ((1, 1), (2, 2), (4, 2), (4, 4), (6, 4), (8, 4)):
            tl = api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src, grid)
            tl.run(); tl.nals(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                tl.run()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nals = tl.nals()
                t2 = time.perf_counter()
                best = min(best, t2 - t0)
            print(f"{W}x{Hh} {depth}-bit, {n:3d} picture(s), tiles {grid[0]}x{grid[1]} ({tl.n_classes} size classes, {tl.n_substreams} substreams): "
                  f"{1e3 * best:8.1f} ms  ({n / best:7.1f} pictures/s; last: device {1e3 * (t1 - t0):.1f} ms + NAL units {1e3 * (t2 - t1):.1f} ms; {sum(map(len, nals))} bytes)", flush=True)
            del tl


if __name__ == "__main__":
    main()
