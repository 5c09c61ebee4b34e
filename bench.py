#!/usr/bin/env python3
"""bench.py -- throughput of the uvg266 per-CTU hot-path kernels on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic 1920x1080 8-bit
yuv420p frame (BASELINE.json configs[1]: all-intra, --preset medium path:
intra prediction + DCT/quant [+ in-loop filters as they land]), with the frame
already resident in HBM.  See workload_text() below and DESIGN.md "Measurement" for
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
HOST_DEBUG = [0.0] if os.environ.get("UVGHIP_BENCH_DEBUG") else None   # seconds the host spent waiting for a frame slot
ALF = False                       # --workload 2160p10alf adds the ALF kernels of config C4 (--alf full)
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
VALU = {}
try:
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "valu_latest.json")) as _f:
        VALU = json.load(_f)
except OSError:
    pass
N_SIMD, CLOCK_GHZ, CYC_PER_VALU = 1024, 2.4, 4     # MI355X: 256 CUs x 4 SIMDs; peak engine clock; wave64 op = 4 cycles on a 16-lane SIMD
# what a SIMD actually sustains under load (tools/dev/valu_rate.hip, profiles/r01e_microbench.txt): one full-rate wave
# instruction (v_add3 / v_perm / v_sad_u16 / v_pk_mad) per 4.85 cycles of the nominal 2.4 GHz clock, i.e. the clock under
# sustained VALU load is ~2.0 GHz; v_dot2_i32_i16 / v_mad_i32_i24 take 6.2
MEASURED_CYC_PER_VALU = 4.85
def workload_text():
    alf = ("-> ALF classification (4x4 Laplacian classes) -> ALF covariance statistics per CTU and class (i8 MFMA) -> ALF "
           "7x7 luma filter " if ALF else "")
    return (f"{W}x{H} {DEPTH}-bit yuv420p, all-intra medium hot path per frame: luma, for N in 32,16,8,4 "
            "{intra rough search 67 modes min(SATD,2SAD) on all NxN blocks with fused arg-min -> intra predict "
            "-> fused residual/DCT-2/quant/dequant/IDCT/recon}; then deblock (Y,U,V; seeded random quad-tree "
            f"partition) -> SAO statistics (4 edge classes + bands per CTU) -> SAO apply {alf}; open-loop references "
            "(source picture); serial RDOQ/CABAC excluded (out of hot-path scope)")


class Frame:
    """Device-resident planes, descriptor tables and every output buffer of one picture (nothing is allocated
    inside the step), plus the prebuilt launch plan: (name, stream slot, C entry point, argument tuple)."""

    def __init__(self, t, device, L, modes_dev):
        y, u, v = layout.synthetic_yuv420(W, H, t, DEPTH)
        dev = lambda a: torch.from_numpy(a).to(device)
        self.host_y, self.host_u, self.host_v = y, u, v
        self.y = dev(y)
        self.uv = dev(np.stack([u, v]))          # both chroma planes in one buffer: the per-frame refresh is one copy
        self.uv_rec = torch.zeros_like(self.uv)  # deblocking works in place on this copy
        self.u_rec, self.v_rec = self.uv_rec[0], self.uv_rec[1]
        self.sao_out = torch.zeros_like(self.y)
        self.scu = api.make_scu_table(layout.quadtree_scu_table(W, H, seed=t, qp=QP), device)
        rects = layout.ctu_rects(W, H)
        self.rects = api.make_rects(rects, device)
        self.n_ctu = len(rects)
        self.edge = torch.zeros((self.n_ctu, 4, 2, 5), dtype=torch.int32, device=device)
        self.band = torch.zeros((self.n_ctu, 2, 32), dtype=torch.int32, device=device)
        self.params = torch.zeros((self.n_ctu, 8), dtype=torch.int32, device=device)
        self.tables, self.bufs = {}, {}
        P = lambda t_: t_.data_ptr()
        ys = self.y.stride(0)
        nm = modes_dev.shape[0]
        self.chains = []          # one list of launches per block size (independent: own pred/rec planes)
        for n in SIZES:
            blks_np = layout.intra_availability(layout.block_grid(W, H, n), n, W, H)
            blks, tus, cnt = api.make_intra_blocks(blks_np, device), api.make_tus(blks_np[:, :2], device), len(blks_np)
            self.tables[n] = (blks, tus, cnt)
            b = {"best": torch.zeros(cnt, dtype=torch.int8, device=device), "cost": torch.zeros(cnt, dtype=torch.int32, device=device),
                 "pred": torch.zeros_like(self.y), "rec": torch.zeros_like(self.y),
                 "coeff": torch.zeros((cnt, n, n), dtype=torch.int16, device=device), "has": torch.zeros(cnt, dtype=torch.uint8, device=device)}
            self.bufs[n] = b
            self.chains.append([
                (f"intra_search_{n}", L.uvghip_intra_search_best_batch,
                 [DEPTH, P(self.y), ys, P(self.y), ys, n, P(blks), cnt, P(modes_dev), nm, P(b["best"]), P(b["cost"]), None]),
                (f"intra_pred_plane_{n}", L.uvghip_intra_pred_plane_batch,
                 [DEPTH, P(self.y), ys, n, P(blks), cnt, P(b["best"]), P(b["pred"]), ys]),
                (f"tu_roundtrip_{n}", L.uvghip_tu_roundtrip_batch,
                 [DEPTH, 0, 0, 0, 0, n, n, QP, 1, P(self.y), ys, P(b["pred"]), ys, P(b["rec"]), ys, P(tus), cnt, P(b["coeff"]), P(b["has"])]),
            ])
        rec = self.bufs[SIZES[-1]]["rec"]       # in-loop filters run on the reconstruction of the last (4x4) pass
        cs = self.u_rec.stride(0)
        self.tail = [
            ("deblock_0", L.uvghip_deblock_frame,
             [DEPTH, P(rec), ys, P(self.u_rec), P(self.v_rec), cs, W, H, P(self.scu), self.scu.shape[1] // 32, 0, 0, 0, QP, None]),
            ("sao_stats_0", L.uvghip_sao_stats_batch, [DEPTH, P(self.y), ys, P(rec), ys, P(self.rects), self.n_ctu, P(self.edge), P(self.band)]),
            ("sao_offsets_0", L.uvghip_sao_edge_offsets_batch, [P(self.edge), None, self.n_ctu, P(self.params), None]),
            ("sao_apply_0", L.uvghip_sao_apply_batch,
             [DEPTH, P(rec), ys, P(self.sao_out), ys, W, H, P(self.rects), P(self.params), self.n_ctu]),
        ]
        if ALF:
            # config C4 (--alf full): classify the SAO output, gather the per-CTU/class covariances against the source,
            # filter with a fixed coefficient set (deriving the filters from the covariances is host-side, alf.c:792-835)
            self.alf_cls = torch.zeros((H // 4, W // 4), dtype=torch.uint8, device=device)
            self.alf_ee = torch.empty((self.n_ctu, 25, 13, 13, 4, 4), dtype=torch.int64, device=device)
            self.alf_y = torch.empty((self.n_ctu, 25, 13, 4), dtype=torch.int32, device=device)
            self.alf_pix = torch.empty((self.n_ctu, 25), dtype=torch.int64, device=device)
            self.alf_out = torch.zeros_like(self.y)
            g = torch.Generator().manual_seed(7)
            coefs = torch.randint(-8, 9, (1, 25, 13), dtype=torch.int16, generator=g)
            coefs[:, :, 12] = 0
            self.alf_coefs = coefs.to(device)
            self.alf_clips = torch.full((1, 25, 13), 1 << DEPTH, dtype=torch.int16, device=device)
            self.alf_set = torch.zeros(self.n_ctu, dtype=torch.int32, device=device)
            so = self.sao_out
            self.tail += [
                ("alf_classify_0", L.uvghip_alf_classify_frame, [DEPTH, P(so), ys, W, H, DEPTH + 4, P(self.alf_cls), self.alf_cls.stride(0)]),
                ("alf_stats_0", L.uvghip_alf_stats_batch,
                 [DEPTH, P(self.y), ys, P(so), ys, W, H, 0, P(self.rects), self.n_ctu, P(self.alf_cls), self.alf_cls.stride(0),
                  P(self.alf_ee), P(self.alf_y), P(self.alf_pix)]),
                ("alf_filter_0", L.uvghip_alf_filter_batch,
                 [DEPTH, P(so), ys, P(self.alf_out), ys, W, H, 0, P(self.rects), P(self.alf_set), self.n_ctu, P(self.alf_coefs),
                  P(self.alf_clips), P(self.alf_cls), self.alf_cls.stride(0)]),
            ]
        self.ev_chain = [torch.cuda.Event() for _ in SIZES]
        self.ev_small = torch.cuda.Event()
        self.ev_done = torch.cuda.Event()
        self.ev_done.record()


class KernelClock:
    """Per-kernel HIP-event timing on the stream the kernel is launched on.  `only` restricts the
    instrumentation to kernels whose family name is in the set (None = all).  Event objects are recycled: spans
    are harvested (elapsed_time read, events returned to the pool) once their frame is known to have retired --
    a few hundred live timing events make every later HIP call slow on this runtime."""

    def __init__(self):
        self.ms = {}          # name -> [total ms, launches]
        self.pending = {}     # slot -> [(name, e0, e1)]
        self.pool = []
        self.only = None

    def _event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def launch(self, name, fn, args, stream, enabled, slot=0):
        timed = enabled and (self.only is None or name.rsplit("_", 1)[0] in self.only)
        if timed:
            e0, e1 = self._event(), self._event()
            e0.record(stream)
        rc = fn(*args, stream.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {lib.load_library().uvghip_last_error().decode()}")
        if timed:
            e1.record(stream)
            self.pending.setdefault(slot, []).append((name, e0, e1))

    def harvest(self, slot=None):
        """Fold the finished spans of `slot` (all slots if None; the caller guarantees they have completed)."""
        for k in ([slot] if slot is not None else list(self.pending)):
            for name, e0, e1 in self.pending.pop(k, []):
                t = self.ms.setdefault(name, [0.0, 0])
                t[0] += e0.elapsed_time(e1); t[1] += 1
                self.pool += [e0, e1]

    def totals(self):
        self.harvest()
        return {k: (v[0], v[1]) for k, v in self.ms.items()}


def algorithmic_bytes(kernel, n, count):
    """SURVEY.md 8(d) per-unit figures x units per launch (b = 1 byte per 8-bit sample)."""
    b = DEPTH // 8 if DEPTH % 8 == 0 else 2
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
    if kernel == "alf_classify":      # SAO output read, one class byte per 4x4 written
        return W * H * b + W * H // 16
    if kernel == "alf_stats":         # orig + rec read, 25 covariances (13x13x16 int64 + 13x4 int32 + int64) per CTU written
        return W * H * 2 * b + count * 25 * (13 * 13 * 16 * 8 + 13 * 4 * 4 + 8)
    if kernel == "alf_filter":        # read + write
        return W * H * 2 * b
    raise KeyError(kernel)


def hot_path_step(fr, clock, timed, main, side):
    """One frame.  The four block sizes are independent chains (search -> predict -> TU round trip, own planes); with
    `side` streams they run concurrently and overlap with the previous frame's in-loop filters on `main`;
    side=None runs everything in order on `main` (profile pass)."""
    # at most n_resident frames in flight: the host waits for this frame's previous use (queueing thousands of
    # launches ahead of the GPU makes the HIP runtime itself slow)
    if HOST_DEBUG is not None:
        _t = time.perf_counter()
    fr.ev_done.synchronize()
    if HOST_DEBUG is not None:
        HOST_DEBUG[0] += time.perf_counter() - _t
    clock.harvest(id(fr))
    for k, chain in enumerate(fr.chains):
        st = side[k] if side else main
        if side:
            st.wait_event(fr.ev_done)            # this frame's buffers: their previous use (4 steps ago) has retired
        for name, fn, args in chain:
            clock.launch(name, fn, args, st, timed, id(fr))
        if side:
            fr.ev_chain[k].record(st)
    if side:
        for ev in fr.ev_chain:
            main.wait_event(ev)
    fr.uv_rec.copy_(fr.uv)                       # torch copies run on the current (= main) stream
    for name, fn, args in fr.tail:
        clock.launch(name, fn, args, main, timed, id(fr))
    fr.ev_done.record(main)


def hot_path_step_split(fr, clock, timed, main, sx, sy, sz):
    """Same launches, other stream plan (--schedule split): the searches -- the only VALU-heavy kernels -- run back to back
    on two streams (32/16 on sx, 8/4 on sy: two searches on the GPU at any time, never a window where all chains are in
    their small latency-bound kernels at once), every predict / TU round trip on sz behind its search's event, the
    in-loop filters on main.  Four streams = the runtime's four hardware queues, none shared."""
    if HOST_DEBUG is not None:
        _t = time.perf_counter()
    fr.ev_done.synchronize()
    if HOST_DEBUG is not None:
        HOST_DEBUG[0] += time.perf_counter() - _t
    clock.harvest(id(fr))
    for st, ks in (((sx, (0, 1, 2, 3)),) if sy is None else ((sx, (0, 1)), (sy, (2, 3)))):
        st.wait_event(fr.ev_done)                # this frame's buffers: their previous use has retired
        for k in ks:
            name, fn, args = fr.chains[k][0]
            clock.launch(name, fn, args, st, timed, id(fr))
            fr.ev_chain[k].record(st)
    order = (0, 1, 2, 3) if sy is None else (0, 2, 1, 3)       # the order in which the searches finish
    for k in order:
        sz.wait_event(fr.ev_chain[k])
        for name, fn, args in fr.chains[k][1:]:
            clock.launch(name, fn, args, sz, timed, id(fr))
    fr.ev_small.record(sz)
    main.wait_event(fr.ev_small)
    fr.uv_rec.copy_(fr.uv)
    for name, fn, args in fr.tail:
        clock.launch(name, fn, args, main, timed, id(fr))
    fr.ev_done.record(main)


def cpu_baseline(fr_host_y, fr_host_u, fr_host_v):
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
    u0, v0 = np.ascontiguousarray(fr_host_u[:rows // 2]), np.ascontiguousarray(fr_host_v[:rows // 2])
    scu = layout.quadtree_scu_table(W, rows, seed=0, qp=QP)
    scu_bytes = np.ascontiguousarray(scu.view(np.uint8).reshape(scu.shape[0], -1))
    rects = np.ascontiguousarray(np.asarray(layout.ctu_rects(W, rows), np.int32).reshape(-1, 4))

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
        # in-loop filters on the 4x4 pass's reconstruction (deblocking in the oracle is single-threaded)
        ur, vr = u0.copy(), v0.copy()
        orc.deblock_frame(DEPTH, rec, ur, vr, W, rows, scu_bytes, scu.shape[1], 0, 0, False, QP, None)
        edge, band = orc.sao_stats_rects(DEPTH, y, rec, rects)
        params, dd = np.zeros((len(rects), 8), np.int32), np.zeros(len(rects), np.int32)
        orc.lib.orc_sao_edge_offsets(Hh.ptr(edge), None, len(rects), Hh.ptr(params), Hh.ptr(dd))
        out = rec.copy()
        for (fx, fy, w_, h_), p_ in zip(rects, params):
            orc.sao_reconstruct_rect(DEPTH, rec, out, W, rows, int(fx), int(fy), int(w_), int(h_), int(p_[0]), int(p_[1]),
                                     [0, 0], list(p_[3:]) + [0] * 5, False)

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
            "sample": f"{slabs} x the top {rows} of {H} luma rows of a 1080p frame ({frames:.3f} frame), the same kernel sequence "
                      f"(search/predict/TU round trip for four block sizes, deblock, SAO statistics/offsets/apply), "
                      f"oracle C -O2, OpenMP over blocks on {cores} threads except deblocking (serial) ({dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="one stream: no overlap between block sizes / frames")
    ap.add_argument("--stream-sets", type=int, default=2, help="sets of per-block-size streams (frames alternate between them)")
    ap.add_argument("--profile-steps", type=int, default=4, help="untimed, fully instrumented steps for the per-kernel table")
    ap.add_argument("--search-streams", type=int, default=2, choices=(1, 2), help="--schedule split: streams the searches alternate over")
    ap.add_argument("--schedule", choices=("chains", "split"), default="split",
                    help="chains: one stream per block size (x --stream-sets); split: searches on two streams, small kernels on a third")
    ap.add_argument("--workload", choices=("1080p8", "2160p10alf"), default="1080p8",
                    help="1080p8 = BASELINE.json configs[1] (the default, the judged line); 2160p10alf = configs[3]: 3840x2160 "
                         "10-bit with the ALF kernels (extra line, no cpu_baseline)")
    args = ap.parse_args()
    global W, H, DEPTH, ALF
    if args.workload == "2160p10alf":
        W, H, DEPTH, ALF = 3840, 2160, 10, True
        TRAFFIC.clear(); VALU.clear()            # the PMC figures under profiles/ belong to the default workload

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
    modes_dev = api.make_modes(MODES, device)
    L = lib.load_library()
    frames = [Frame(rank + k * world, device, L, modes_dev) for k in range(n_resident)]
    clock = KernelClock()
    main_stream = torch.cuda.current_stream()
    # two sets of side streams: consecutive frames use different sets, so the searches of frame f+1 can start while
    # the chains of frame f are still draining
    side_sets = None if args.serial else [[torch.cuda.Stream(device=device) for _ in SIZES] for _ in range(args.stream_sets)]
    side = None if args.serial else side_sets[0]
    split = None if (args.serial or args.schedule != "split") else [torch.cuda.Stream(device=device) for _ in range(3)]

    def step(fr, clk, timed, s_idx):
        if split:
            hot_path_step_split(fr, clk, timed, main_stream, split[0], split[1] if args.search_streams == 2 else None, split[2])
        else:
            hot_path_step(fr, clk, timed, main_stream, side_sets[s_idx % len(side_sets)] if side_sets else None)

    for s in range(args.warmup):
        step(frames[s % n_resident], clock, False, s)
    torch.cuda.synchronize()

    # untimed profile pass: every kernel bracketed by HIP events -> per-kernel breakdown and the dominant family
    prof = KernelClock()
    for s in range(args.profile_steps):
        hot_path_step(frames[s % n_resident], prof, True, main_stream, None)
    torch.cuda.synchronize()
    prof_tot = prof.totals()
    fam_ms = {}
    for name, (ms, launches) in prof_tot.items():
        fam_ms[name.rsplit("_", 1)[0]] = fam_ms.get(name.rsplit("_", 1)[0], 0.0) + ms
    dom_family = max(fam_ms, key=fam_ms.get)

    # timed region: only the dominant family's launches carry events (live roofline timing), the rest run bare
    clock.only = {dom_family}
    for s in range(min(4, args.warmup)):          # back to the streamed plan (and its clocks) after the serial profile pass
        step(frames[s % n_resident], clock, False, s)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    if HOST_DEBUG is not None:
        HOST_DEBUG[:] = [0.0, 0.0]
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(frames[s % n_resident], clock, True, s)
    if HOST_DEBUG is not None:
        _t = time.perf_counter()
    torch.cuda.synchronize()
    if HOST_DEBUG is not None:
        HOST_DEBUG[1] = HOST_DEBUG[0] + time.perf_counter() - _t
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if HOST_DEBUG is not None and rank == 0:
        print(f"[debug] timed region {elapsed * 1e3:.1f} ms, host blocked on frame slots {HOST_DEBUG[1] * 1e3:.1f} ms "
              f"(issue time per step {(elapsed - HOST_DEBUG[1]) / args.steps * 1e6:.0f} us)", file=sys.stderr)
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
            "metric": f"hot-path fps ({H}p all-intra medium kernel path; Mpixels/s in config)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8" if DEPTH == 8 else "u16", "data": "synthetic",
            "config": {"workload": workload_text(), "mpixels_per_s": round(fps * W * H / 1e6, 1), "qp": QP,
                       "parallelism": f"frames sharded over {world} rank(s), no data-path collective",
                       "streams": 1 if args.serial else 1 + len(SIZES) * args.stream_sets},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": live[dom]["gbs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(live[dom]["gbs"] / HBM_PEAK_GBS, 5), "traffic": TRAFFIC.get(dom),
                         "avg_launch_ms": live[dom]["avg_ms"], "alg_bytes_per_launch": live[dom]["alg_bytes"],
                         "launches_timed": live[dom]["launches"],
                         "serial_avg_launch_ms": per_kernel[dom]["avg_ms"],
                         "note": "HIP events (on the launch stream) around every launch of the dominant kernel family inside the "
                                 "timed region, where the four block sizes run on concurrent streams, so a launch shares the GPU; "
                                 "serial_avg_launch_ms = the same launch alone (profile pass); "
                                 "traffic = PMC bytes per launch from profiles/ (null if not collected)"},
            "kernel_sum_ms": round(tot_ms, 4),
            "valu": (lambda v: None if v is None else {
                "kernel": dom, "insts_per_launch": v["valu_insts"],
                "issue_util_alone": round(v["valu_insts"] * CYC_PER_VALU / (N_SIMD * per_kernel[dom]["avg_ms"] * 1e-3 * CLOCK_GHZ * 1e9), 3),
                "issue_util_vs_measured_rate": round(v["valu_insts"] * MEASURED_CYC_PER_VALU / (N_SIMD * per_kernel[dom]["avg_ms"] * 1e-3 * CLOCK_GHZ * 1e9), 3),
                "note": "SQ_INSTS_VALU (PMC, --serial) x 4 cycles / (1024 SIMDs x launch duration alone (with its event pair) x 2.4 GHz): "
                        "the limiter of the dominant kernel is integer VALU issue, not HBM.  issue_util_vs_measured_rate prices an "
                        "instruction at the 4.85 cycles a SIMD sustains for full-rate ops in tools/dev/valu_rate.hip "
                        "(profiles/r01e_microbench.txt; a quarter of the kernel's ops are v_dot2 at 6.2), i.e. the fraction of the "
                        "VALU issue rate the hardware really delivers"})(VALU.get(dom)),
            "kernels_timed_region": per_kernel_live,
            "kernels": per_kernel,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "1080p8":
            out["cpu_baseline"] = cpu_baseline(frames[0].host_y, frames[0].host_u, frames[0].host_v)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
