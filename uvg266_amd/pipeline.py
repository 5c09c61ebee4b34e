"""The per-picture hot path as a prebuilt launch plan over the batched C ABI -- the rank driver of the CTU-row
sharded encoder path and, with one rank, the whole-frame path `bench.py` times.

One picture on one rank (`BandFrame`) = every device buffer (nothing is allocated per step), the descriptor tables
of the CTU rows the rank owns, and lists of launches `(name, fn, args)` with the convention `fn(*args, stream)`:

  chains[k]    block size N = 32, 16, 8, 4 (independent of each other, own prediction / reconstruction planes):
               luma rough search (67 modes, fused arg-min) -> luma predict -> fused luma TU round trip, then for
               N >= 8 the co-located chroma blocks (N/2) with the derived mode (search_intra.c:1657): predict U, V ->
               TU round trip U, V
  stage_a      deblocking, vertical edges of the band
  xchg_dbk     halo exchange (rows around the band boundaries + per-4x4 side information)
  stage_b      deblocking, horizontal edges incl. both boundary edges; SAO statistics / offsets / apply for Y, U, V
  xchg_alf     halo exchange of SAO output rows                                   (ALF workloads only)
  stage_c      ALF classification, covariance statistics (+ per-class sums), 7x7 luma and 5x5 chroma filters
  reduce       all-reduce of the per-class covariance sums (frame-level filter derivation, alf.c:792-835)
  xchg_gather  every band of the final picture to every rank (the next frame's inter reference)

With nranks == 1 the exchanges are empty and the plan is the whole-frame hot path.  References are open-loop (the
source picture) -- the closed-loop CU dependency is search control flow, not part of these kernels.
"""
import numpy as np
import torch

from . import api, layout
from . import lib as _lib
from .bands import BandLayout, spec_bytes

SIZES = (32, 16, 8, 4)            # --pu-depth-intra 1-4 (cfg.c:769-801)
MODES = list(range(67))           # every luma mode; the reference's rough search visits a subset

WORKLOADS = {
    # BASELINE.json configs[1]: 1920x1080 8-bit all-intra --preset medium (rdoq 1, no ALF: cfg.c:769-801)
    "1080p8": dict(W=1920, H=1080, depth=8, alf=False, rdoq=True),
    # BASELINE.json configs[3]: 3840x2160 10-bit --preset medium --alf full
    "2160p10alf": dict(W=3840, H=2160, depth=10, alf=True, rdoq=True),
    # small pictures for tests
    "test8": dict(W=328, H=264, depth=8, alf=True, rdoq=True),
    "test10": dict(W=328, H=264, depth=10, alf=True, rdoq=False),
}


def synthetic_rdoq_ctx(seed=22):
    """A fixed CABAC context snapshot (uvghip_rdoq_ctx_t, 244 CTX_STATE bytes) for the synthetic workloads: moderately
    skewed probabilities.  In the encoder this is state->cabac.ctx as it stands when the CU is coded."""
    rng = np.random.default_rng(seed)
    return rng.integers(40, 216, 244).astype(np.uint8)


def intra_lambda(qp):
    """The order of magnitude of the encoder's RD lambda at this QP (0.57 * 2^((qp-12)/3))."""
    return 0.57 * 2.0 ** ((qp - 12) / 3.0)


def chroma_qp(qp):
    """uvg_get_scaled_qp for chroma with the default (identity below 30) VVC mapping table, QP < 30 only."""
    assert qp < 30
    return qp


class TuPool:
    """Coefficient buffers of a GROUP of F pictures, per (block size, plane): what sits between the per-picture plane
    kernels and the quantiser, which runs once per group (RDOQ is a long sequential walk per block: only many blocks per
    launch fill the GPU).  Picture f uses slice f of every buffer."""

    def __init__(self, device, frames):
        self.device, self.F, self.jobs = device, frames, {}

    def job(self, key, cnt, c):
        """key = (block size, plane); cnt blocks of c x c per picture."""
        if key not in self.jobs:
            F, d = self.F, self.device
            self.jobs[key] = dict(cnt=cnt, c=c,
                                  coef=torch.zeros((F, cnt, c, c), dtype=torch.int16, device=d),     # transformed residual
                                  lev=torch.zeros((F, cnt, c, c), dtype=torch.int16, device=d),      # levels = the reference's coeff_out
                                  deq=torch.zeros((F, cnt, c, c), dtype=torch.int16, device=d),      # dequantised
                                  has=torch.zeros((F, cnt), dtype=torch.uint8, device=d),
                                  ws=torch.empty((_lib.load_library().uvghip_rdoq_workspace_bytes(c, c, F * cnt) // 8 + 64,), dtype=torch.float64, device=d))
        return self.jobs[key]


class GroupArena:
    """Plane storage of a GROUP of F pictures: every named plane is one (F, rows, width) tensor, picture f owns slice f.  The
    F slices stacked are one tall plane (F * rows, width) with the same stride, so the kernels that take block lists
    (search, predict, transforms) run ONCE over the group with the lists of all pictures (y offset f * rows) -- a 1080p
    picture alone does not fill 256 CUs for these kernels."""

    def __init__(self, F, device):
        self.F, self.device, self.t = F, device, {}

    def take(self, name, slot, shape, dtype, fill=0):
        if name not in self.t:
            self.t[name] = torch.full((self.F,) + tuple(shape), fill, dtype=dtype, device=self.device)
        return self.t[name][slot]

    def stacked(self, name):
        t = self.t[name]
        return t.view(-1, *t.shape[2:])


class BandFrame:
    def __init__(self, L, wl, t, device, modes_dev, rank=0, nranks=1, qp=22, transport=None, poison=False, gather=True, pool=None,
                 slot=0, arena=None):
        W, H, depth, alf = wl["W"], wl["H"], wl["depth"], wl["alf"]
        self.W, self.H, self.depth, self.alf, self.qp = W, H, depth, alf, qp
        self.rdoq = rdoq = bool(wl.get("rdoq", False))
        self._keep = []                                                     # ctypes parameter blocks referenced by launches
        self.L, self.device = L, device
        self.pool = pool if pool is not None else TuPool(device, 1)
        self.slot = slot
        self.heads, self.tails = [], []                                     # per block size: launches before / after the quantiser
        self.band = band = BandLayout(H, nranks, rank)
        self.rank, self.nranks = rank, nranks
        y0, y1 = band.y0, band.y1
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        P = lambda t_: t_.data_ptr()
        y, u, v = layout.synthetic_yuv420(W, H, t, depth)
        self.host = (y, u, v)
        if arena is None:
            self.y, self.u, self.v = dev(y), dev(u), dev(v)                # the source picture (every rank holds all of it)
        else:
            self.y, self.u, self.v = (arena.take(k, slot, a.shape, dev(a).dtype) for k, a in (("y", y), ("u", u), ("v", v)))
            self.y.copy_(dev(y)); self.u.copy_(dev(u)); self.v.copy_(dev(v))
        ys, cs = self.y.stride(0), self.u.stride(0)
        qps = qp + 6 * (depth - 8)                                          # uvg_get_scaled_qp(0, qp, (depth-8)*6, ..) transform.c:150
        qpc = chroma_qp(qp) + 6 * (depth - 8)
        nm = modes_dev.shape[0]
        pz = 0x55 if poison else 0                                          # rows a rank does not own: a value no kernel produces by luck

        def plane(like, name=None):
            if arena is not None and name is not None:
                return arena.take(name, slot, like.shape, like.dtype, pz)
            return torch.full_like(like, pz) if poison else torch.zeros_like(like)

        def vec(name, cnt, dtype):
            return arena.take(name, slot, (cnt,), dtype) if arena is not None else torch.zeros(cnt, dtype=dtype, device=device)

        # ---- intra search / predict / TU round trip, per block size, for the blocks of the owned rows ----
        self.tables, self.bufs, self.chains, self.own = {}, {}, [], {}
        self.rec_y = None
        for n in SIZES:
            allb = layout.intra_availability(layout.block_grid(W, H, n), n, W, H)
            own = allb[(allb[:, 1] >= y0) & (allb[:, 1] < y1)]
            cnt = len(own)
            blks, tus = api.make_intra_blocks(own, device), api.make_tus(own[:, :2], device)
            b = {"best": vec(f"best{n}", cnt, torch.int8), "cost": vec(f"cost{n}", cnt, torch.int32),
                 "pred": plane(self.y, f"pred{n}"), "rec": plane(self.y, f"rec{n}")}
            self.tables[n] = (blks, tus, cnt)
            self.own[n] = own
            head = [
                (f"intra_search_{n}", L.uvghip_intra_search_best_batch,
                 [depth, P(self.y), ys, P(self.y), ys, n, P(blks), cnt, P(modes_dev), nm, P(b["best"]), P(b["cost"]), None]),
                (f"intra_pred_plane_{n}", L.uvghip_intra_pred_plane_batch,
                 [depth, P(self.y), ys, n, P(blks), cnt, P(b["best"]), P(b["pred"]), ys]),
            ]
            mid, tail = [], []
            self._tu_launches(f"{n}", 0, n, n, qps, self.y, b["pred"], b["rec"], tus, cnt, head, mid, tail)
            if n >= 8:
                c = n // 2
                cown = own // 2                                             # chroma block position and available reference counts
                cblks, ctus = api.make_intra_blocks(cown, device), api.make_tus(cown[:, :2], device)
                for name, src in (("u", self.u), ("v", self.v)):
                    b["pred_" + name], b["rec_" + name] = plane(src, f"pred{n}{name}"), plane(src, f"rec{n}{name}")
                b["cblks"], b["ctus"] = cblks, ctus
                for name, src in (("u", self.u), ("v", self.v)):
                    head.append((f"intra_pred_chroma_{n}", L.uvghip_intra_pred_plane_chroma_batch,
                                 [depth, P(src), cs, c, P(cblks), cnt, P(b["best"]), P(b["pred_" + name]), cs]))
                for ci, (name, src) in enumerate((("u", self.u), ("v", self.v))):
                    self._tu_launches(f"chroma_{n}", 1 + ci, n, c, qpc, src, b["pred_" + name], b["rec_" + name], ctus, cnt, head, mid, tail)
            self.bufs[n] = b
            # a short last band may hold no block of this size
            self.heads.append(head if cnt else []); self.tails.append(tail if cnt else [])
            self.chains.append(head + mid + tail if cnt else [])      # the stand-alone chain of this picture (quantiser on its own slice)

        # ---- in-loop filters on the reconstruction of the finest passes (luma 4x4, chroma 4x4 = the N = 8 pass) ----
        self.rec_y, self.rec_u, self.rec_v = self.bufs[4]["rec"], self.bufs[8]["rec_u"], self.bufs[8]["rec_v"]
        tab = layout.quadtree_scu_table(W, H, seed=t, qp=qp)
        if nranks > 1:                                                      # a rank only knows the side information of its own rows
            keep = np.zeros(tab.shape[0], bool)
            keep[y0 // 4:(y1 + 3) // 4] = True
            tab = tab.copy()
            tab[~keep] = np.zeros((), layout.SCU_DTYPE)
        self.scu = api.make_scu_table(tab, device)
        scu_stride = self.scu.shape[1] // 32
        rects = layout.ctu_rects(W, H)
        own_r = rects[(rects[:, 1] >= y0) & (rects[:, 1] < y1)]
        crects = own_r // 2
        self.n_ctu = n_ctu = len(own_r)
        self.rects, self.crects = api.make_rects(own_r, device), api.make_rects(crects, device)
        self.sao_y, self.sao_u, self.sao_v = plane(self.y, "sao_y"), plane(self.u, "sao_u"), plane(self.v, "sao_v")
        self.own_rects = own_r
        comp = (("y", self.y, self.rec_y, self.sao_y, self.rects, ys, W, H),
                ("u", self.u, self.rec_u, self.sao_u, self.crects, cs, W // 2, H // 2),
                ("v", self.v, self.rec_v, self.sao_v, self.crects, cs, W // 2, H // 2))
        # the three planes' per-CTU statistics / parameters side by side: the offset derivation is one launch over all of them
        self.edge_all = torch.zeros((3, n_ctu, 4, 2, 5), dtype=torch.int32, device=device)
        self.params_all = torch.zeros((3, n_ctu, 8), dtype=torch.int32, device=device)
        self.edge = {k: self.edge_all[i] for i, k in enumerate("yuv")}
        self.bandst = {k: torch.zeros((n_ctu, 2, 32), dtype=torch.int32, device=device) for k in "yuv"}
        self.params = {k: self.params_all[i] for i, k in enumerate("yuv")}
        dbk = [depth, P(self.rec_y), ys, P(self.rec_u), P(self.rec_v), cs, W, H, P(self.scu), scu_stride, 0, 0, 0, qp, None, y0, y1]
        self.stage_a = [("deblock_v_0", L.uvghip_deblock_band, dbk + [1])]
        self.stage_b = [("deblock_h_0", L.uvghip_deblock_band, dbk + [2])]
        self.stage_b_dbk = list(self.stage_b)                                # (a FrameGroup runs SAO once over the group instead)
        for k, org, rec, out, rc, st, pw, ph in comp:
            self.stage_b.append((f"sao_stats_{k}_0", L.uvghip_sao_stats_batch, [depth, P(org), st, P(rec), st, P(rc), n_ctu, P(self.edge[k]), P(self.bandst[k])]))
        self.stage_b.append(("sao_offsets_yuv_0", L.uvghip_sao_edge_offsets_batch, [P(self.edge_all), None, 3 * n_ctu, P(self.params_all), None]))
        for k, org, rec, out, rc, st, pw, ph in comp:
            self.stage_b.append((f"sao_apply_{k}_0", L.uvghip_sao_apply_batch, [depth, P(rec), st, P(out), st, pw, ph, P(rc), P(self.params[k]), n_ctu]))
        self.final = (self.sao_y, self.sao_u, self.sao_v)
        self.stage_c, self.reduce = [], []
        if alf:
            # classify the SAO output, gather the per-CTU / class covariances against the source, filter with fixed
            # coefficient sets (deriving the filters from the covariances is host-side, alf.c:792-835)
            self.alf_cls = torch.zeros((H // 4, W // 4), dtype=torch.uint8, device=device)
            self.alf_rec = torch.empty((n_ctu, 25, 1484), dtype=torch.int64, device=device)
            self.alf_present = torch.zeros(n_ctu, dtype=torch.int32, device=device)
            self.alf_sums = torch.zeros((25, 1509), dtype=torch.int64, device=device)
            self.alf_y, self.alf_u, self.alf_v = plane(self.y), plane(self.u), plane(self.v)
            g = torch.Generator().manual_seed(7)
            coefs = torch.randint(-8, 9, (1, 25, 13), dtype=torch.int16, generator=g)
            coefs[:, :, 12] = 0
            ccoefs = torch.randint(-8, 9, (1, 7), dtype=torch.int16, generator=g)
            ccoefs[:, 6] = 0
            self.alf_coefs, self.alf_ccoefs = coefs.to(device), ccoefs.to(device)
            self.alf_clips = torch.full((1, 25, 13), 1 << depth, dtype=torch.int16, device=device)
            self.alf_cclips = torch.full((1, 7), 1 << depth, dtype=torch.int16, device=device)
            self.alf_set = torch.zeros(n_ctu, dtype=torch.int32, device=device)
            so, cst = self.sao_y, self.alf_cls.stride(0)
            self.stage_c = [
                ("alf_classify_0", L.uvghip_alf_classify_band, [depth, P(so), ys, W, H, depth + 4, P(self.alf_cls), cst, y0, y1]),
                ("alf_stats_0", L.uvghip_alf_stats_compact_batch,
                 [depth, P(self.y), ys, P(so), ys, W, H, 0, P(self.rects), n_ctu, P(self.alf_cls), cst, P(self.alf_rec), P(self.alf_present)]),
                ("alf_cov_reduce_0", L.uvghip_alf_cov_reduce, [P(self.alf_rec), P(self.alf_present), n_ctu, 0, P(self.alf_sums)]),
                ("alf_filter_y_0", L.uvghip_alf_filter_batch,
                 [depth, P(so), ys, P(self.alf_y), ys, W, H, 0, P(self.rects), P(self.alf_set), n_ctu, P(self.alf_coefs), P(self.alf_clips), P(self.alf_cls), cst]),
            ]
            for k, src, dst in (("u", self.sao_u, self.alf_u), ("v", self.sao_v, self.alf_v)):
                self.stage_c.append((f"alf_filter_{k}_0", L.uvghip_alf_filter_batch,
                                     [depth, P(src), cs, P(dst), cs, W // 2, H // 2, 1, P(self.crects), P(self.alf_set), n_ctu,
                                      P(self.alf_ccoefs), P(self.alf_cclips), None, 0]))
            self.final = (self.alf_y, self.alf_u, self.alf_v)

        # ---- exchanges ----
        self.spec_dbk = band.halo_deblock(self.rec_y, self.rec_u, self.rec_v, self.scu)
        self.spec_alf = band.halo_alf(self.sao_y, self.sao_u, self.sao_v) if alf else []
        self.spec_gather = band.gather(*self.final) if gather else []
        self.xchg_dbk, self.xchg_alf, self.xchg_gather = [], [], []
        if transport is not None and nranks > 1:
            for nm_, spec, dst in (("halo_dbk_0", self.spec_dbk, self.xchg_dbk), ("halo_alf_0", self.spec_alf, self.xchg_alf),
                                   ("gather_0", self.spec_gather, self.xchg_gather)):
                if spec:
                    fn, args = transport.launch_args(spec)
                    dst.append((nm_, fn, args))
            if alf:
                fn, args = transport.allreduce_args(self.alf_sums)
                self.reduce.append(("allreduce_cov_0", fn, args))

    def _tu_launches(self, tag, color, n, c, qp_scaled, orig, pred, rec, tus, cnt, head, mid, tail):
        """The reconstruction of the c x c TUs of block size n in one plane.  Medium runs uvg_quantize_residual on its RDOQ
        branch (cfg.c:781, quant-generic.c:527): residual + transform per picture (head), RDOQ + dequantisation on the picture's
        slice of the group buffers (mid; a FrameGroup replaces these by one launch over all its pictures), inverse transform +
        reconstruction per picture (tail).  Without RDOQ: the single-launch plain-quant round trip."""
        L, P, depth = self.L, (lambda t_: t_.data_ptr()), self.depth
        st = orig.stride(0)
        if cnt == 0:
            return
        j = self.pool.job((n, color), cnt, c)
        f = self.slot
        if not self.rdoq:
            head.append((f"tu_roundtrip_{tag}", L.uvghip_tu_roundtrip_batch,
                         [depth, 0, 0, 0, 0, c, c, qp_scaled, 1, P(orig), st, P(pred), st, P(rec), st, P(tus), cnt, P(j["lev"][f]), P(j["has"][f])]))
            return
        head.append((f"tu_forward_{tag}", L.uvghip_tu_forward_batch,
                     [depth, 0, 0, 0, 0, c, c, 0, P(orig), st, P(pred), st, P(tus), cnt, P(j["coef"][f])]))
        mid += quantiser_launches(L, self, tag, color, c, qp_scaled, j, f, 1)
        tail.append((f"tu_inverse_{tag}", L.uvghip_tu_inverse_batch,
                     [depth, 0, 0, 0, 0, c, c, 0, P(j["deq"][f]), P(pred), st, P(rec), st, P(tus), cnt]))

    # -- what one step moves over xGMI for this rank --
    def comm_bytes(self):
        out = {"halo_deblock": spec_bytes(self.spec_dbk), "halo_alf": spec_bytes(self.spec_alf), "gather": spec_bytes(self.spec_gather)}
        out["allreduce_cov"] = (self.alf_sums.numel() * 8,) * 2 if (self.alf and self.nranks > 1) else (0, 0)
        return out

    def filter_launches(self):
        """The in-loop filter part in order, exchanges included."""
        return (self.stage_a + self.xchg_dbk + self.stage_b + self.xchg_alf + self.stage_c + self.reduce + self.xchg_gather)

    def all_launches(self):
        """This picture on its own (tests, smoke): complete per-size chains, then the filters."""
        return [l for c in self.chains for l in c] + self.filter_launches()


def run(launches, stream_handle, L=None):
    """Issue a list of launches on one stream."""
    from . import lib as _lib
    for name, fn, args in launches:
        rc = fn(*args, stream_handle)
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {_lib.load_library().uvghip_last_error().decode()}")


def quantiser_launches(L, fr, tag, color, c, qp_scaled, job, f0, nf):
    """RDOQ + dequantisation over pictures [f0, f0 + nf) of a group job."""
    import ctypes
    P = lambda t_: t_.data_ptr()
    if not hasattr(fr, "_rdoq_ctx"):
        fr._rdoq_ctx = (ctypes.c_uint8 * 244).from_buffer_copy(synthetic_rdoq_ctx().tobytes())
    lam = intra_lambda(fr.qp) * (1.0 if color == 0 else 0.9)
    n = nf * job["cnt"]
    off = lambda t_: t_[f0].data_ptr()
    return [(f"rdoq_{tag}", L.uvghip_rdoq_batch,
             [fr.depth, off(job["coef"]), off(job["lev"]), c, c, n, color, 1, 0, 0, 0, qp_scaled, ctypes.c_double(lam),
              ctypes.cast(fr._rdoq_ctx, ctypes.c_void_p), P(job["ws"]), job["ws"].numel() * 8, None, off(job["has"])]),
            (f"dequant_{tag}", L.uvghip_dequant_batch, [fr.depth, off(job["lev"]), off(job["deq"]), c, c, n, qp_scaled, 0])]


class FrameGroup:
    """F pictures processed together (the frame-parallel operation of an all-intra encode, uvg266 --owf).  Their planes
    live in a GroupArena, so every kernel that takes a block list -- rough search, predict, residual + forward
    transform, RDOQ + dequantisation, inverse transform + reconstruction -- runs once per block shape over all F
    pictures; the in-loop filters, which work on whole planes with picture-edge rules, run per picture."""

    def __init__(self, L, wl, t0, F, device, modes_dev, step=1, **kw):
        self.pool = TuPool(device, F)
        self.arena = GroupArena(F, device)
        self.frames = [BandFrame(L, wl, t0 + f * step, device, modes_dev, pool=self.pool, slot=f, arena=self.arena, **kw) for f in range(F)]
        fr = self.frames[0]
        self.F, self.rdoq = F, fr.rdoq
        depth, H = fr.depth, fr.H
        qps, qpc = fr.qp + 6 * (depth - 8), chroma_qp(fr.qp) + 6 * (depth - 8)
        P = lambda t_: t_.data_ptr()
        A = self.arena.stacked
        nm = modes_dev.shape[0]
        self._keep = []
        self._searches, self._heads, self.mid, self._tails = [], [], [], []

        def stack_blocks(per_frame, rows):
            allb = np.concatenate([o + np.array([0, k * rows, 0, 0]) for k, o in enumerate(per_frame)])
            blks, tus = api.make_intra_blocks(allb, device), api.make_tus(allb[:, :2], device)
            self._keep += [blks, tus]
            return blks, tus

        def tu(tag, color, n, c, qp_scaled, orig, pred, rec, tus, cnt):
            st = orig.stride(0)
            j = self.pool.job((n, color), cnt // F, c)
            if not fr.rdoq:
                self._heads.append((f"tu_roundtrip_{tag}", L.uvghip_tu_roundtrip_batch,
                                    [depth, 0, 0, 0, 0, c, c, qp_scaled, 1, P(orig), st, P(pred), st, P(rec), st, P(tus), cnt, P(j["lev"]), P(j["has"])]))
                return
            self._heads.append((f"tu_forward_{tag}", L.uvghip_tu_forward_batch,
                                [depth, 0, 0, 0, 0, c, c, 0, P(orig), st, P(pred), st, P(tus), cnt, P(j["coef"])]))
            self.mid += quantiser_launches(L, fr, tag, color, c, qp_scaled, j, 0, F)[:1]      # RDOQ; dequantisation rides on the inverse
            self._tails.append((f"tu_inverse_{tag}", L.uvghip_tu_dequant_inverse_batch,
                                [depth, 0, 0, c, c, qp_scaled, P(j["lev"]), P(pred), st, P(rec), st, P(tus), cnt]))

        for n in SIZES:
            cnt = fr.tables[n][2]
            if cnt == 0:
                continue
            assert all(f.tables[n][2] == cnt for f in self.frames)
            blks, tus = stack_blocks([f.own[n] for f in self.frames], H)
            Y, ys = A("y"), A("y").stride(0)
            best, cost = self.arena.t[f"best{n}"].view(-1), self.arena.t[f"cost{n}"].view(-1)
            self._searches.append((f"intra_search_{n}", L.uvghip_intra_search_best_stacked_batch,
                                   [depth, P(Y), ys, P(Y), ys, n, P(blks), F * cnt, P(modes_dev), nm, P(best), P(cost), None, H]))
            self._heads.append((f"intra_pred_plane_{n}", L.uvghip_intra_pred_plane_stacked_batch,
                                [depth, P(Y), ys, n, P(blks), F * cnt, P(best), P(A(f"pred{n}")), ys, H, 0]))
            tu(f"{n}", 0, n, n, qps, Y, A(f"pred{n}"), A(f"rec{n}"), tus, F * cnt)
            if n >= 8:
                c = n // 2
                cblks, ctus = stack_blocks([f.own[n] // 2 for f in self.frames], H // 2)
                cs = A("u").stride(0)
                for name in "uv":
                    self._heads.append((f"intra_pred_chroma_{n}", L.uvghip_intra_pred_plane_stacked_batch,
                                        [depth, P(A(name)), cs, c, P(cblks), F * cnt, P(best), P(A(f"pred{n}{name}")), cs, H // 2, 1]))
                for ci, name in enumerate("uv"):
                    tu(f"chroma_{n}", 1 + ci, n, c, qpc, A(name), A(f"pred{n}{name}"), A(f"rec{n}{name}"), ctus, F * cnt)

        # ---- SAO once over the group (whole pictures per rank only: with CTU-row bands the halo exchanges sit between the
        #      per-picture stages).  Statistics are per rectangle, the offsets per rectangle, and uvghip_sao_apply_batch looks
        #      at a rectangle's row inside its own picture: one launch per plane over the rectangles of all pictures. ----
        self.sao = []
        if fr.nranks == 1:
            W = fr.W
            n_ctu = fr.n_ctu
            ry = np.concatenate([f.own_rects + np.array([0, k * H, 0, 0]) for k, f in enumerate(self.frames)])
            rc = np.concatenate([f.own_rects // 2 + np.array([0, k * (H // 2), 0, 0]) for k, f in enumerate(self.frames)])
            rects_y, rects_c = api.make_rects(ry, device), api.make_rects(rc, device)
            self.sao_edge = torch.zeros((3, F * n_ctu, 4, 2, 5), dtype=torch.int32, device=device)
            self.sao_band = {k: torch.zeros((F * n_ctu, 2, 32), dtype=torch.int32, device=device) for k in "yuv"}
            self.sao_params = torch.zeros((3, F * n_ctu, 8), dtype=torch.int32, device=device)
            self._keep += [rects_y, rects_c]
            rec_names = {"y": "rec4", "u": "rec8u", "v": "rec8v"}
            comp = [(i, k, A(k), A(rec_names[k]), A("sao_" + k), rects_y if k == "y" else rects_c,
                     W if k == "y" else W // 2, H if k == "y" else H // 2) for i, k in enumerate("yuv")]
            for i, k, org, rec, out, rcts, pw, ph in comp:
                st = org.stride(0)
                self.sao.append((f"sao_stats_{k}_0", L.uvghip_sao_stats_batch,
                                 [depth, P(org), st, P(rec), st, P(rcts), F * n_ctu, P(self.sao_edge[i]), P(self.sao_band[k])]))
            self.sao.append(("sao_offsets_yuv_0", L.uvghip_sao_edge_offsets_batch, [P(self.sao_edge), None, 3 * F * n_ctu, P(self.sao_params), None]))
            for i, k, org, rec, out, rcts, pw, ph in comp:
                st = org.stride(0)
                self.sao.append((f"sao_apply_{k}_0", L.uvghip_sao_apply_batch,
                                 [depth, P(rec), st, P(out), st, pw, ph, P(rcts), P(self.sao_params[i]), F * n_ctu]))

    def filters(self):
        """The in-loop filters of the group in order (whole pictures per rank: SAO once over the group)."""
        if not self.sao:
            return [l for fr in self.frames for l in fr.filter_launches()]
        return ([l for fr in self.frames for l in fr.stage_a] + [l for fr in self.frames for l in fr.stage_b_dbk] + self.sao +
                [l for fr in self.frames for l in fr.stage_c + fr.reduce])

    def searches(self):
        return list(self._searches)

    def heads_rest(self):
        """Predict + residual / forward transform (everything before the quantiser but the searches), once per block shape."""
        return list(self._heads)

    def tails(self):
        return list(self._tails)

    def before_filters(self):
        """Everything between the searches and the in-loop filters, in an order that respects the dependencies."""
        return self.heads_rest() + self.mid + self.tails()

    def all_launches(self):
        return self.searches() + self.before_filters() + self.filters()


class Graph:
    """Launches captured once into a hipGraph (uvghip_graph_*) and replayed with one call.  `branches`: independent
    launch lists that fork from the capture stream onto `side_streams` and join again before `launches` (the four block-size
    chains of a picture: the graph then lets the long, narrow kernels of one chain run beside the others)."""

    def __init__(self, L, launches, capture_stream, branches=None, side_streams=None):
        import ctypes
        from . import lib as _lib
        self.L = L
        self.names = [l[0] for b in (branches or []) for l in b] + [l[0] for l in launches]
        self.handle = ctypes.c_void_p()
        h = capture_stream.cuda_stream
        _lib.check(L.uvghip_graph_begin(h), "uvghip_graph_begin")
        forked = []          # the side streams that were really pulled into the capture (only those may be joined back)
        try:
            if branches:
                fork = torch.cuda.Event()
                fork.record(capture_stream)
                joins = []
                for k, br in enumerate(branches):
                    if k == 0 or not side_streams:
                        run(br, h)
                        continue
                    st = side_streams[(k - 1) % len(side_streams)]
                    st.wait_event(fork)                      # pulls the side stream into the capture
                    if st not in forked:
                        forked.append(st)
                    run(br, st.cuda_stream)
                    ev = torch.cuda.Event()
                    ev.record(st)
                    joins.append(ev)
                for ev in joins:
                    capture_stream.wait_event(ev)
            run(launches, h)
        except BaseException:
            # a launch failed mid-capture: pull every forked side stream back into the capture stream before ending the capture
            # (an unjoined branch would make hipStreamEndCapture fail and hide the real error), drop whatever was captured.
            # Streams that never joined the capture are left alone (waiting on their events from a capturing stream is an error),
            # and the capture is always ended so that the original exception is the one that propagates.
            try:
                for st in forked:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    capture_stream.wait_event(ev)
            except BaseException:
                pass
            finally:
                L.uvghip_graph_end(h, ctypes.byref(self.handle))
                self.destroy()
            raise
        rc = L.uvghip_graph_end(h, ctypes.byref(self.handle))
        _lib.check(rc, "uvghip_graph_end")

    def launch(self, stream_handle):
        rc = self.L.uvghip_graph_launch(self.handle, stream_handle)
        if rc != 0:
            from . import lib as _lib
            _lib.check(rc, "uvghip_graph_launch")

    def destroy(self):
        if self.handle:
            self.L.uvghip_graph_destroy(self.handle)
            self.handle = None
