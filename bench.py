#!/usr/bin/env python3
"""bench.py -- throughput of the uvg266 per-CTU hot-path kernels on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic 1920x1080 8-bit
yuv420p frame (BASELINE.json configs[1]: all-intra, --preset medium path:
intra prediction + DCT/quant [+ in-loop filters as they land]), with the frame
already resident in HBM.  See WORKLOAD below and DESIGN.md "Measurement" for
exactly which kernels run; serial RDOQ/CABAC are outside the hot-path scope
(SURVEY.md section 8), so this is hot-path frames/s, not .266 frames/s.

Frames are independent in all-intra coding, so ranks take disjoint frames and
no collective sits on the data path ("scaling": "weak").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from uvg266_amd import api, layout, lib  # noqa: E402

W, H, DEPTH, QP = 1920, 1080, 8, 22
SIZES = (32, 16, 8, 4)            # --pu-depth-intra 1-4 (cfg.c:769-801)
MODES = list(range(67))           # every luma mode; the reference's rough search visits a subset
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
# HBM bytes per launch from the PMC passes under profiles/ (FETCH_SIZE doubled per the guide's gfx950 note + WRITE_SIZE)
TRAFFIC = {}
try:
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic_latest.json")) as _f:
        TRAFFIC = json.load(_f)
except OSError:
    pass
WORKLOAD = ("1920x1080 8-bit yuv420p, all-intra medium hot path per frame: luma, for N in 32,16,8,4 "
            "{intra rough search 67 modes min(SATD,2SAD) on all NxN blocks with fused arg-min -> intra predict "
            "-> fused residual/DCT-2/quant/dequant/IDCT/recon}; then deblock (Y,U,V; seeded random quad-tree "
            "partition) -> SAO statistics (4 edge classes + bands per CTU) -> SAO apply; open-loop references "
            "(source picture); serial RDOQ/CABAC excluded (out of hot-path scope)")


class Frame:
    """Device-resident planes + descriptor tables of one picture."""

    def __init__(self, t, device):
        y, u, v = layout.synthetic_yuv420(W, H, t, DEPTH)
        self.y = torch.from_numpy(y).to(device)
        self.pred = torch.zeros_like(self.y)
        self.rec = torch.zeros_like(self.y)
        self.tables = {}
        for n in SIZES:
            blks = layout.intra_availability(layout.block_grid(W, H, n), n, W, H)
            self.tables[n] = (api.make_intra_blocks(blks, device), api.make_tus(blks[:, :2], device), len(blks))
        self.host_y = y
        self.u = torch.from_numpy(u).to(device)
        self.v = torch.from_numpy(v).to(device)
        self.u_rec = torch.zeros_like(self.u)
        self.v_rec = torch.zeros_like(self.v)
        self.sao_out = torch.zeros_like(self.y)
        self.scu = api.make_scu_table(layout.quadtree_scu_table(W, H, seed=t, qp=QP), device)
        rects = layout.ctu_rects(W, H)
        self.rects = api.make_rects(rects, device)
        self.n_ctu = len(rects)


class KernelClock:
    """Per-kernel HIP-event timing on the launch stream (torch's current stream).  `only` restricts the
    instrumentation to kernels whose family name is in the set (None = all)."""

    def __init__(self):
        self.spans = {}
        self.only = None

    def run(self, name, fn, enabled):
        if not enabled or (self.only is not None and name.rsplit("_", 1)[0] not in self.only):
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.spans.setdefault(name, []).append((e0, e1))
        return out

    def totals(self):
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.spans.items()}


def algorithmic_bytes(kernel, n, count):
    """SURVEY.md 8(d) per-unit figures x units per launch (b = 1 byte per 8-bit sample)."""
    b = 1
    if kernel == "intra_search":      # refs (4N+1) + original NxN read, best mode + cost written (fused arg-min)
        return count * ((4 * n + 1) * b + n * n * b + 5)
    if kernel == "intra_pred_plane":  # refs read, NxN written
        return count * ((4 * n + 1) * b + n * n * b)
    if kernel == "tu_roundtrip":      # orig + pred read, levels (int16) + recon written
        return count * n * n * (2 * b + 2 + b)
    if kernel == "select_best":
        return count * (4 * len(MODES) + 1 + 4)
    if kernel == "deblock":           # 2 * 1.5*W*H*b (read+write) + 32 B side info per 4x4 (n = 0)
        return int(2 * 1.5 * W * H * b + 32 * W * H / 16)
    if kernel == "sao_stats":         # orig + rec luma read, 104 counters per CTU written
        return W * H * 2 * b + count * 104 * 4
    if kernel == "sao_offsets":       # 40 counters read, 8 words written per CTU
        return count * (40 + 8) * 4
    if kernel == "sao_apply":         # rec read, out written (+ 32 B parameters per CTU)
        return W * H * 2 * b + count * 32
    raise KeyError(kernel)


def hot_path_step(fr, modes_dev, clock, timed):
    for n in SIZES:
        blks, tus, cnt = fr.tables[n]
        best, _ = clock.run(f"intra_search_{n}", lambda: api.intra_search_best_batch(fr.y, fr.y, blks, n, modes_dev), timed)
        clock.run(f"intra_pred_plane_{n}", lambda: api.intra_pred_plane_batch(fr.y, blks, n, best, fr.pred), timed)
        clock.run(f"tu_roundtrip_{n}", lambda: api.tu_roundtrip_batch(fr.y, fr.pred, fr.rec, tus, n, n, QP), timed)
    # in-loop filters on the reconstruction left by the last (4x4) pass
    fr.u_rec.copy_(fr.u)
    fr.v_rec.copy_(fr.v)
    clock.run("deblock_0", lambda: api.deblock_frame(fr.rec, fr.u_rec, fr.v_rec, fr.scu, W, H, 0, 0, False, QP, None), timed)
    edge, band = clock.run("sao_stats_0", lambda: api.sao_stats_batch(fr.y, fr.rec, fr.rects), timed)
    params = clock.run("sao_offsets_0", lambda: api.sao_edge_offsets_batch(edge), timed)
    clock.run("sao_apply_0", lambda: api.sao_apply_batch(fr.rec, fr.sao_out, fr.rects, params), timed)


def cpu_baseline(fr_host_y):
    """The oracle (C restatement, OpenMP over blocks) on the host cores, on a bounded sample:
    slabs of the top 256 luma rows of one frame (4 CTU rows = 23.7 % of a frame), repeated until
    about 10 s of wall time have passed; same kernels as the GPU step."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as Hh
    orc = Hh.load_oracle()
    rows = 256
    y = np.ascontiguousarray(fr_host_y[:rows])
    modes = np.asarray(MODES, np.int8)
    tables = {n: layout.intra_availability(layout.block_grid(W, rows, n), n, W, rows) for n in SIZES}

    def one_slab():
        for n in SIZES:
            blks = tables[n]
            costs = np.zeros((len(blks), len(modes)), np.uint32)
            orc.fn(DEPTH, "intra_search_frame", None)(Hh.ptr(y), W, Hh.ptr(y), W, W, rows, n, Hh.ptr(blks), len(blks),
                                                      Hh.ptr(modes), len(modes), Hh.ptr(costs))
            best = np.ascontiguousarray(modes[np.argmin(costs, 1)])
            pred = np.zeros_like(y)
            orc.fn(DEPTH, "intra_pred_plane_frame", None)(Hh.ptr(y), W, W, rows, n, Hh.ptr(blks), len(blks),
                                                          Hh.ptr(best), Hh.ptr(pred), W)
            rec = np.zeros_like(y)
            coeff = np.zeros(len(blks) * n * n, np.int16)
            tus = np.ascontiguousarray(blks[:, :2])
            orc.fn(DEPTH, "tu_roundtrip_frame", None)(DEPTH, n, n, QP, 1, Hh.ptr(y), Hh.ptr(pred), Hh.ptr(rec), W,
                                                      Hh.ptr(tus), len(tus), Hh.ptr(coeff))

    slabs = 0
    t0 = time.perf_counter()
    while slabs < 1 or (time.perf_counter() - t0 < 10.0 and slabs < 64):
        one_slab()
        slabs += 1
    dt = time.perf_counter() - t0
    frames = slabs * rows / H
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{slabs} x the top {rows} of {H} luma rows of a 1080p frame ({frames:.3f} frame), same kernels, "
                      f"oracle C -O2 + OpenMP on {cores} threads ({dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=4, help="untimed, fully instrumented steps for the per-kernel table")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    lib.init(local_rank)

    # every rank owns its own frames (frame t = rank + k*world): independent units, no exchange
    n_resident = 4
    frames = [Frame(rank + k * world, device) for k in range(n_resident)]
    modes_dev = api.make_modes(MODES, device)
    clock = KernelClock()

    for s in range(args.warmup):
        hot_path_step(frames[s % n_resident], modes_dev, clock, False)
    torch.cuda.synchronize()

    # untimed profile pass: every kernel bracketed by HIP events -> per-kernel breakdown and the dominant family
    prof = KernelClock()
    for s in range(args.profile_steps):
        hot_path_step(frames[s % n_resident], modes_dev, prof, True)
    torch.cuda.synchronize()
    prof_tot = prof.totals()
    fam_ms = {}
    for name, (ms, launches) in prof_tot.items():
        fam_ms[name.rsplit("_", 1)[0]] = fam_ms.get(name.rsplit("_", 1)[0], 0.0) + ms
    dom_family = max(fam_ms, key=fam_ms.get)

    # timed region: only the dominant family's launches carry events (live roofline timing), the rest run bare
    clock.only = {dom_family}
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        hot_path_step(frames[s % n_resident], modes_dev, clock, True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        def table(totals):
            t = {}
            for name, (ms, launches) in totals.items():
                kern, n = name.rsplit("_", 1)
                cnt = frames[0].tables[int(n)][2] if int(n) else frames[0].n_ctu
                byts = algorithmic_bytes(kern, int(n), cnt)
                avg_ms = ms / launches
                t[name] = {"avg_ms": round(avg_ms, 4), "share": 0.0, "alg_bytes": byts, "launches": launches,
                           "gbs": round(byts / (avg_ms * 1e-3) / 1e9, 2)}
            return t
        per_kernel = table(prof_tot)                 # all kernels, untimed profile pass
        tot_ms = sum(v["avg_ms"] for v in per_kernel.values())
        for v in per_kernel.values():
            v["share"] = round(v["avg_ms"] / tot_ms, 3)
        live = table(clock.totals())                 # dominant family, inside the timed region
        dom = max(live, key=lambda k: live[k]["avg_ms"])
        per_kernel_live = live
        fps = args.steps * world / elapsed
        out = {
            "metric": "hot-path fps (1080p all-intra medium kernel path; Mpixels/s in config)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": WORKLOAD, "mpixels_per_s": round(fps * W * H / 1e6, 1), "qp": QP,
                       "parallelism": f"frames sharded over {world} rank(s), no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": live[dom]["gbs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(live[dom]["gbs"] / HBM_PEAK_GBS, 5), "traffic": TRAFFIC.get(dom),
                         "avg_launch_ms": live[dom]["avg_ms"], "alg_bytes_per_launch": live[dom]["alg_bytes"],
                         "launches_timed": live[dom]["launches"],
                         "note": "HIP events around every launch of the dominant kernel family inside the timed region; "
                                 "traffic = PMC bytes per launch from profiles/ (null if not collected)"},
            "kernel_sum_ms": round(tot_ms, 4),
            "kernels_timed_region": per_kernel_live,
            "kernels": per_kernel,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames[0].host_y)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
