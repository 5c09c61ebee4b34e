#!/usr/bin/env python3
"""Dev-time: what ONE rank of a tile-sharded node would do, measured on one GPU -- a projection, not a scaling measurement.  A 60-picture
3840x2160 10-bit all-intra clip under --tiles <grid> --wpp; for every rank r of `world` the plan of the tiles uvg266_amd.tiles.assign gives it
(uvghip_tiles_plan_create_owned) is run ALONE and timed (run + its contribution to the NAL units on the host).  Ranks share nothing while
they run (no halo) and meet in two all-gathers of ~1.4 MB per picture, so the node's time for the clip is the slowest rank's + the gather;
beside it the same clip with all tiles on this one GPU.

  python tools/dev/tiles_rank_time.py [world=8] [pictures=60]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from uvg266_amd import api, layout, tiles as T

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W, Hh, depth, qp = 3840, 2160, 10, 22
prm = H.search_params(W, Hh, qp)
src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W, Hh, t % 7, depth)) for t in range(n)]


def timed(tl, whole):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tl.run()
        (tl.nals() if whole else tl.substreams())
        best = min(best, time.perf_counter() - t0)
    return best


for grid in ((4, 2), (8, 4)):
    rects, _ = api.tile_grid(W, Hh, *grid)
    owner = T.assign(rects, world)
    tl = api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src, grid)
    one = timed(tl, True)
    del tl
    per = []
    for r in range(world):
        tl = api.TiledLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), src, grid, owned=owner == r)
        per.append(timed(tl, False))
        del tl
    print(f"{n} pictures {W}x{Hh} {depth}-bit, tiles {grid[0]}x{grid[1]}, {world} ranks ({int((owner == 0).sum())} tile(s) each): all tiles on one GPU {1e3 * one:.0f} ms "
          f"({n / one:.1f} pictures/s); a rank alone {1e3 * min(per):.0f} .. {1e3 * max(per):.0f} ms -> {n / max(per):.1f} pictures/s for the node before the gather "
          f"({one / max(per):.2f} x this GPU)", flush=True)
