"""Test-side helpers: oracle loader (ctypes), golden-file reader, KAT patterns."""
import ctypes
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_ORC = None

c_int, c_uint, c_vp = ctypes.c_int, ctypes.c_uint, ctypes.c_void_p


def px_dtype(depth):
    return np.uint8 if depth == 8 else np.uint16


def load_oracle():
    """dlopen oracle/liborc.so (built by `make -C oracle`; rebuilt here if stale or absent)."""
    global _ORC
    if _ORC is None:
        path = os.path.join(ROOT, "oracle", "liborc.so")
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        _ORC = Oracle(ctypes.CDLL(path))
    return _ORC


def ptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_vp)


class Oracle:
    """Typed access to orc8_* / orc10_* (oracle/*.c)."""

    def __init__(self, lib):
        self.lib = lib

    def fn(self, depth, name, restype=c_int):
        f = getattr(self.lib, f"orc{depth}_{name}")
        f.restype = restype

        def call(*args):
            return f(*[int(a) if isinstance(a, np.integer) else a for a in args])
        return call

    # ---- picture ---------------------------------------------------------
    def reg_sad(self, d, a, b, w, h, sa, sb):
        return self.fn(d, "reg_sad", c_uint)(ptr(a), ptr(b), w, h, c_uint(sa), c_uint(sb))

    def image_calc_sad(self, d, pic, ref, ref_w, ref_h, px, py, rx, ry, bw, bh):
        return self.fn(d, "image_calc_sad", c_uint)(ptr(pic), pic.shape[1], ptr(ref), ref.shape[1],
                                                     ref_w, ref_h, px, py, rx, ry, bw, bh)

    def sad_nxn(self, d, a, b, n):
        return self.fn(d, "sad_nxn", c_uint)(ptr(a), ptr(b), n)

    def satd_nxn(self, d, a, b, n):
        return self.fn(d, "satd_nxn", c_uint)(ptr(a), ptr(b), n)

    def sad_nxn_dual(self, d, preds, orig, n):
        out = np.zeros(2, np.uint32)
        self.fn(d, "sad_nxn_dual", None)(ptr(preds), ptr(orig), n, ptr(out))
        return out

    def satd_nxn_dual(self, d, preds, orig, n):
        out = np.zeros(2, np.uint32)
        self.fn(d, "satd_nxn_dual", None)(ptr(preds), ptr(orig), n, ptr(out))
        return out

    def satd_any_size(self, d, w, h, a, sa, b, sb, a_off=0, b_off=0):
        es = a.itemsize
        return self.fn(d, "satd_any_size", c_uint)(w, h, c_vp(a.ctypes.data + int(a_off) * es), sa,
                                                   c_vp(b.ctypes.data + int(b_off) * es), sb)

    def satd_any_size_quad(self, d, w, h, base, offs, ps, orig, os_):
        arr = (c_vp * 4)(*[c_vp(base.ctypes.data + int(o) * base.itemsize) for o in offs])
        out = np.zeros(4, np.uint32)
        self.fn(d, "satd_any_size_quad", None)(w, h, arr, ps, ptr(orig), os_, ptr(out))
        return out

    def pixels_calc_ssd(self, d, a, b, sa, sb, w, h):
        return self.fn(d, "pixels_calc_ssd", c_uint)(ptr(a), ptr(b), sa, sb, w, h)

    def generate_residual(self, d, a, b, w, h, sa, sb):
        res = np.zeros(w * h, np.int16)
        self.fn(d, "generate_residual", None)(ptr(a), ptr(b), ptr(res), w, h, sa, sb)
        return res


    # ---- dct ---------------------------------------------------------------
    def dct_nxn(self, d, bitdepth, n, block, inverse=False):
        out = np.zeros(n * n, np.int16)
        self.fn(d, "idct_nxn" if inverse else "dct_nxn", None)(bitdepth, n, ptr(block), ptr(out))
        return out

    def tr(self, d, bitdepth, inverse, hor, ver, w, h, sw, sh, block):
        out = np.zeros(w * h, np.int16)
        self.fn(d, "tr_inverse" if inverse else "tr_forward", None)(bitdepth, hor, ver, w, h, sw, sh, ptr(block), ptr(out))
        return out

    def mts_dct(self, d, bitdepth, color, cu_intra, cu_inter, isp, lfnst, cr_lfnst, tr_idx, w, h, block, mts_type, inverse):
        out = np.zeros(w * h, np.int16)
        self.fn(d, "mts_dct", None)(bitdepth, color, cu_intra, cu_inter, isp, lfnst, cr_lfnst, tr_idx, w, h,
                                    ptr(block), ptr(out), mts_type, int(inverse))
        return out

    def mts_select(self, d, w, h, color, cu_intra, cu_inter, isp, lfnst, cr_lfnst, tr_idx, mts_type):
        hor, ver, sw, sh = (ctypes.c_int() for _ in range(4))
        self.fn(d, "get_tr_type", None)(w, h, color, cu_intra, cu_inter, isp, lfnst, cr_lfnst, tr_idx, mts_type,
                                        ctypes.byref(hor), ctypes.byref(ver))
        lf = (lfnst and color == 0) or (cr_lfnst and color != 0)
        self.fn(d, "mts_skips", None)(w, h, hor.value, ver.value, int(bool(lf)), ctypes.byref(sw), ctypes.byref(sh))
        return hor.value, ver.value, sw.value, sh.value


    # ---- quant -------------------------------------------------------------
    def quant(self, d, coef, w, h, bitdepth, qp_scaled, ts, intra):
        out = np.zeros(w * h, np.int16)
        self.fn(d, "quant", None)(ptr(coef), ptr(out), w, h, bitdepth, qp_scaled, ts, intra)
        return out

    def dequant(self, d, q, w, h, bitdepth, qp_scaled, ts):
        out = np.zeros(w * h, np.int16)
        self.fn(d, "dequant", None)(ptr(q), ptr(out), w, h, bitdepth, qp_scaled, ts)
        return out

    def coeff_abs_sum(self, d, c):
        return self.fn(d, "coeff_abs_sum", ctypes.c_uint32)(ptr(c), ctypes.c_size_t(c.size))

    def fast_coeff_cost(self, d, c, w, h, weights):
        return self.fn(d, "fast_coeff_cost", ctypes.c_uint32)(ptr(c), w, h, ctypes.c_uint64(weights))

    def tu_roundtrip(self, d, bitdepth, hor, ver, sw, sh, w, h, qp_scaled, intra, ref, pred, stride, x0=0, y0=0):
        """-> (has_coeffs, coeff, rec plane copy)"""
        rec = pred.copy()
        coeff = np.zeros(w * h, np.int16)
        es = ref.itemsize
        off = (y0 * stride + x0) * es
        has = self.fn(d, "tu_roundtrip")(bitdepth, hor, ver, sw, sh, w, h, qp_scaled, intra,
                                         c_vp(ref.ctypes.data + off), c_vp(pred.ctypes.data + off), stride,
                                         c_vp(rec.ctypes.data + off), stride, ptr(coeff))
        return has, coeff, rec


    # ---- RDOQ ----------------------------------------------------------------
    RDOQ_CTX_BYTES = 244

    def rdoq(self, d, coef, w, h, color, block_type, cbf_u, lfnst, mts, qp_scaled, lam, ctx):
        """-> (levels (h*w,) int16, abs_sum).  ctx: 244 uint8 = uvghip_rdoq_ctx_t."""
        assert ctx.dtype == np.uint8 and ctx.size == self.RDOQ_CTX_BYTES
        out = np.full(w * h, 0x33, np.int16)
        s = self.fn(d, "rdoq")(ptr(np.ascontiguousarray(coef, np.int16)), ptr(out), w, h, color, block_type, cbf_u, lfnst, mts, qp_scaled,
                               ctypes.c_double(lam), ptr(ctx))
        return out, s

    def rdoq_sh(self, d, coef, w, h, color, block_type, cbf_u, lfnst, mts, qp_scaled, lam, ctx, signhide=1):
        out = np.full(w * h, 0x33, np.int16)
        s = self.fn(d, "rdoq_sh")(ptr(np.ascontiguousarray(coef, np.int16)), ptr(out), w, h, color, block_type, cbf_u, lfnst, mts, qp_scaled,
                                  ctypes.c_double(lam), ptr(ctx), signhide)
        return out, s

    def quant_sh(self, d, coef, w, h, bitdepth, qp_scaled, ts, intra, lfnst):
        out = np.zeros(w * h, np.int16)
        self.fn(d, "quant_sh", None)(ptr(np.ascontiguousarray(coef, np.int16)), ptr(out), w, h, bitdepth, qp_scaled, ts, intra, lfnst)
        return out

    def coeff_cost(self, d, coeff, w, h, color, models):
        """-> (bits, flags, adapted models bytes).  models: 1220 bytes = uvghip_cabac_models_t."""
        m = np.ascontiguousarray(np.frombuffer(bytes(models), np.uint8)).copy()
        assert m.size == 1220
        out = np.zeros(1220, np.uint8)
        flags = ctypes.c_uint32(0)
        bits = self.fn(d, "coeff_cost", ctypes.c_double)(ptr(np.ascontiguousarray(coeff, np.int16)), w, h, color, ptr(m), ctypes.byref(flags), ptr(out))
        return bits, flags.value, out

    def get_extended_block(self, d, wrap, src, src_w, src_h, bx, by, bw, bh, pl, pr, pt, pb, pbs):
        """-> (inside, ext_off, ext_s, buf)"""
        buf = np.full((pt + bh + pb + pbs) * (pl + bw + pr), 0x5a, src.dtype)
        off, es = ctypes.c_long(), ctypes.c_int()
        inside = self.fn(d, "get_extended_block")(int(wrap), ptr(src), src_w, src_h, src.shape[1], bx, by, bw, bh, pl, pr, pt, pb, pbs,
                                                  ptr(buf), ctypes.byref(off), ctypes.byref(es))
        return inside, off.value, es.value, buf

    # ---- intra -------------------------------------------------------------
    REF_LEN = 400

    def intra_build_refs(self, d, rec, pic_w, pic_h, x, y, w, h, at, al):
        top = np.zeros(self.REF_LEN, rec.dtype); left = np.zeros(self.REF_LEN, rec.dtype)
        self.fn(d, "intra_build_refs", None)(ptr(rec), rec.shape[1], pic_w, pic_h, x, y, w, h, at, al, ptr(top), ptr(left))
        return top, left

    def intra_filter_refs(self, d, top, left, w, h):
        ft = np.zeros_like(top); fl = np.zeros_like(left)
        self.fn(d, "intra_filter_refs", None)(ptr(top), ptr(left), w, h, ptr(ft), ptr(fl))
        return ft, fl

    def intra_predict(self, d, mode, is_chroma, w, h, top, left, ftop, fleft):
        dst = np.zeros(w * h, top.dtype)
        self.fn(d, "intra_predict")(mode, int(is_chroma), w, h, ptr(top), ptr(left), ptr(ftop), ptr(fleft), ptr(dst))
        return dst

    def angular_pred(self, d, w, h, mode, is_chroma, above, left, mrl=0, isp=0):
        dst = np.zeros(w * h, above.dtype)
        self.fn(d, "angular_pred", None)(w, h, mode, int(is_chroma), ptr(above), ptr(left), ptr(dst), mrl, isp)
        return dst

    def intra_mode_costs(self, d, rec, pic_w, pic_h, x, y, n, at, al, orig, modes, want_preds=False):
        modes = np.asarray(modes, np.int8)
        costs = np.zeros(len(modes), np.uint32)
        preds = np.zeros(len(modes) * n * n, rec.dtype) if want_preds else None
        self.fn(d, "intra_mode_costs", None)(ptr(rec), rec.shape[1], pic_w, pic_h, x, y, n, at, al, ptr(orig),
                                             ptr(modes), len(modes), ptr(costs), ptr(preds) if want_preds else None)
        return (costs, preds) if want_preds else costs


    # ---- ipol --------------------------------------------------------------
    def ipol_sample(self, d, ref, pic_w, pic_h, x0, y0, w, h, fx, fy, chroma=False, hi=False):
        out = np.zeros(w * h, np.int16 if hi else ref.dtype)
        self.fn(d, "ipol_sample", None)(ptr(ref), ref.shape[1], pic_w, pic_h, x0, y0, w, h, fx, fy, int(chroma), int(hi), ptr(out), w)
        return out

    def frac_satd(self, d, cur, cx, cy, ref, pic_w, pic_h, rx, ry, w, h, cands):
        cands = np.ascontiguousarray(np.asarray(cands, np.int16).reshape(-1, 2))
        costs = np.zeros(len(cands), np.uint32)
        self.fn(d, "frac_satd", None)(ptr(cur), cur.shape[1], cx, cy, ptr(ref), ref.shape[1], pic_w, pic_h, rx, ry, w, h,
                                      ptr(cands), len(cands), ptr(costs))
        return costs

    def bipred_average(self, d, l0, l1, w, h):
        out = np.zeros(w * h, px_dtype(d))
        mode = (1 if l0.dtype == np.int16 else 0) | (2 if l1.dtype == np.int16 else 0)
        self.fn(d, "bipred_average", None)(ptr(out), w, ptr(l0), ptr(l1), mode, w, h)
        return out


    # ---- sao ---------------------------------------------------------------
    def sao_stats_rects(self, d, orig, rec, rects):
        rects = np.ascontiguousarray(np.asarray(rects, np.int32).reshape(-1, 4))
        edge = np.zeros((len(rects), 4, 2, 5), np.int32); band = np.zeros((len(rects), 2, 32), np.int32)
        self.fn(d, "sao_stats_rects", None)(ptr(orig), ptr(rec), rec.shape[1], ptr(rects), len(rects), ptr(edge), ptr(band))
        return edge, band

    def sao_reconstruct_rect(self, d, rec, out, pic_w, pic_h, fx, fy, w, h, typ, eo, band_position, offsets, is_v):
        bp = np.asarray(band_position, np.int32); of = np.asarray(offsets, np.int32)
        self.fn(d, "sao_reconstruct_rect", None)(ptr(rec), ptr(out), rec.shape[1], pic_w, pic_h, fx, fy, w, h, typ, eo,
                                                 ptr(bp), ptr(of), int(is_v))


    # ---- deblock -----------------------------------------------------------
    def deblock_frame(self, d, y, u, v, width, height, scu_bytes, scu_stride, beta_off, tc_off, slice_is_b, frame_qp, qp_map):
        qm = np.ascontiguousarray(np.asarray(qp_map, np.int8)) if qp_map is not None else None
        self.fn(d, "deblock_frame", None)(ptr(y), y.shape[1], ptr(u) if u is not None else None, ptr(v) if v is not None else None,
                                          u.shape[1] if u is not None else 0, width, height, ptr(scu_bytes), scu_stride,
                                          beta_off, tc_off, int(slice_is_b), frame_qp, ptr(qm) if qm is not None else None)


    # ---- alf ---------------------------------------------------------------
    def alf_classify_frame(self, d, rec, w, h, shift, vbh=64, vb_pos=60):
        cls = np.zeros((h // 4, w // 4), np.uint8)
        self.fn(d, "alf_classify_frame", None)(ptr(rec), rec.shape[1], w, h, shift, vbh, vb_pos, ptr(cls), cls.shape[1])
        return cls

    def alf_filter_rect(self, d, src, dst, pic_w, pic_h, x0, y0, w, h, chroma, coef, clip, cls):
        vbh, vbp = (32, 30) if chroma else (64, 60)
        self.fn(d, "alf_filter_rect", None)(ptr(src), ptr(dst), src.shape[1], pic_w, pic_h, x0, y0, w, h, int(chroma),
                                            ptr(coef), ptr(clip), ptr(cls) if cls is not None else None,
                                            cls.shape[1] if cls is not None else 0, vbh, vbp)

    def alf_stats_rect(self, d, org, rec, pic_w, pic_h, x0, y0, w, h, chroma, cls):
        C = 1 if chroma else 25
        vbh, vbp = (32, 30) if chroma else (64, 60)
        ee = np.zeros((C, 13, 13, 4, 4), np.int64); yv = np.zeros((C, 13, 4), np.int32); pa = np.zeros(C, np.int64)
        clip = np.array([1 << d] + [1 << (7 - 2 * i + d - 8) for i in (1, 2, 3)], np.int16)
        self.fn(d, "alf_stats_rect", None)(ptr(org), org.shape[1], ptr(rec), rec.shape[1], pic_w, pic_h, x0, y0, w, h, int(chroma),
                                           ptr(cls) if cls is not None else None, cls.shape[1] if cls is not None else 0,
                                           vbh, vbp, ptr(clip), ptr(ee), ptr(yv), ptr(pa))
        return ee, yv, pa


# ---- closed-loop CTU search (oracle/orc_search.c) ------------------------------
class SearchParams(ctypes.Structure):
    """orc_search_params: what the search reads from encoder_state_t / encoder_control_t."""
    _fields_ = [("pic_w", ctypes.c_int32), ("pic_h", ctypes.c_int32), ("qp", ctypes.c_int32), ("qp_c", ctypes.c_int32),
                ("depth_min", ctypes.c_int32), ("depth_max", ctypes.c_int32), ("wpp", ctypes.c_int32),
                ("combine_intra_cus", ctypes.c_int32), ("rough_levels", ctypes.c_int32), ("rd", ctypes.c_int32),
                ("lam", ctypes.c_double), ("lam_sqrt", ctypes.c_double), ("c_lam", ctypes.c_double),
                ("cw_u", ctypes.c_double), ("cw_v", ctypes.c_double)]


N_MODELS = 257
MODELS_BYTES = 1286          # uint16 state0[257], state1[257], uint8 rate[257], padded to 2


def search_params(W, H, qp, lam=None):
    """--preset medium -p 1 (pu-depth-intra 1-4, WPP, combine-intra-cus, two rough-search levels); the default chroma QP
    table is the identity, lambda = 0.57 * 2^((qp - 12) / 3) for an intra picture (rate_control.c qp_to_lambda)."""
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0) if lam is None else lam
    return SearchParams(W, H, qp, qp, 1, 4, 1, 1, 2, 0, lam, float(np.sqrt(lam)), lam, 1.0, 1.0)


def oracle_search_picture(orc, depth, prm, y, u, v):
    """-> dict(rec_y, rec_u, rec_v, cu [h16, w16, 11], trees [h16, w16, 2], coeff [ctus, 6144], models [ctus, 3, 1286])"""
    W, H = prm.pic_w, prm.pic_h
    wc, hc = (W + 63) // 64, (H + 63) // 64
    px = px_dtype(depth)
    y, u, v = (np.ascontiguousarray(a, px) for a in (y, u, v))
    ry, ru, rv = np.zeros((H, W), px), np.zeros((H // 2, W // 2), px), np.zeros((H // 2, W // 2), px)
    cu = np.zeros((hc * 16, wc * 16, 20), np.uint8)
    co = np.zeros((wc * hc, 6144), np.int16)
    mo = np.zeros((wc * hc, 3, MODELS_BYTES), np.uint8)
    rc = orc.fn(depth, "search_intra_picture")(ctypes.byref(prm), ptr(y), ptr(u), ptr(v), ptr(ry), ptr(ru), ptr(rv), ptr(cu), ptr(co), ptr(mo))
    assert rc == 0
    trees = np.ascontiguousarray(cu[:, :, 12:]).view(np.uint32).reshape(hc * 16, wc * 16, 2)
    return dict(rec_y=ry, rec_u=ru, rec_v=rv, cu=cu[:, :, :11].copy(), trees=trees, coeff=co, models=mo)


class CtuParams(ctypes.Structure):
    """ctu::params (uvg266_amd/csrc/ctu_core.h) = uvghip_ctu_params_t: orc_search_params + the chroma lambda of the transform units."""
    _fields_ = SearchParams._fields_ + [("c_lam_tu", ctypes.c_double)]


def ctu_params(prm):
    c = CtuParams()
    for f, _ in SearchParams._fields_:
        setattr(c, f, getattr(prm, f))
    c.c_lam_tu = prm.lam / 2.0 ** ((prm.qp - prm.qp_c) / 3.0)       # uvg_calculate_chroma_lambda (rate_control.c:1216-1233)
    return c


SCU_NP = np.dtype([("luma_edges", "u1"), ("chroma_edges", "u1"), ("type", "u1"), ("cbf", "u1"), ("qp", "i1"), ("log2_width", "u1"),
                   ("log2_height", "u1"), ("log2_chroma_width", "u1"), ("log2_chroma_height", "u1"), ("isp_mode", "u1"), ("mv_dir", "u1"),
                   ("reserved", "u1"), ("ref_id", "<i2", (2,)), ("mv", "<i4", (2, 2))])      # uvghip_scu_t


def search_result_from_device_layout(W, H, ry, ru, rv, scu, coeff, models):
    """The closed-loop search's device-side outputs (uvghip_scu_t table, packed models) in the layout of oracle_search_picture."""
    wc, hc = (W + 63) // 64, (H + 63) // 64
    cu = np.zeros((hc * 16, wc * 16, 11), np.uint8)
    t = scu.reshape(hc * 16, wc * 16)
    for j, f in enumerate(("type", "log2_width", "log2_height", "log2_chroma_width", "log2_chroma_height", "cbf")):
        cu[:, :, j] = t[f]
    cu[:, :, 6] = t["mv"][:, :, 0, 0] & 0xff
    cu[:, :, 7] = (t["mv"][:, :, 0, 0] >> 8) & 0xff
    cu[:, :, 8], cu[:, :, 9], cu[:, :, 10] = t["luma_edges"], t["chroma_edges"], t["qp"].astype(np.uint8)
    trees = np.stack([t["mv"][:, :, 0, 1].astype(np.uint32), t["mv"][:, :, 1, 0].astype(np.uint32)], axis=2)
    m = models.reshape(hc * wc, 3, N_MODELS)
    mb = np.zeros((hc * wc, 3, MODELS_BYTES), np.uint8)
    mb[:, :, :514] = np.ascontiguousarray((m & 0xffff).astype(np.uint16)).view(np.uint8).reshape(hc * wc, 3, 514)
    mb[:, :, 514:1028] = np.ascontiguousarray((m >> 16).astype(np.uint16)).view(np.uint8).reshape(hc * wc, 3, 514)
    mb[:, :, 1028:1028 + N_MODELS] = model_rates()
    return dict(rec_y=ry, rec_u=ru, rec_v=rv, cu=cu, trees=trees, coeff=coeff.reshape(hc * wc, 6144), models=mb)


_RATES = None


def model_rates():
    """The window byte of every model: row 3 of k_ctx_init in uvg266_amd/csrc/vvc_ctx_init.h (0 for the gaps of the index space)."""
    global _RATES
    if _RATES is None:
        import re
        txt = open(os.path.join(ROOT, "uvg266_amd", "csrc", "vvc_ctx_init.h")).read()
        rows = re.findall(r"\{([0-9,\s]+)\}", txt[txt.index("k_ctx_init"):])
        tab = [np.array([int(v) for v in r.replace("\n", " ").split(",") if v.strip()], np.uint8) for r in rows[:4]]
        assert all(len(t) == N_MODELS for t in tab)
        _RATES = np.where(tab[2] == 255, 0, tab[3]).astype(np.uint8)
    return _RATES


_EMUL = None


def load_ctu_emulation():
    """tests/emul: the CTU search kernel's source built for the host with one emulated lane (CPU tests of the device logic)."""
    global _EMUL
    if _EMUL is None:
        d = os.path.join(ROOT, "tests", "emul")
        subprocess.check_call(["make", "-s", "-C", d])
        _EMUL = ctypes.CDLL(os.path.join(d, "_build", "libctu_emul.so"))
    return _EMUL


def emul_search_picture(depth, prm, y, u, v, lazy=False):
    lib = load_ctu_emulation()
    lib.ctu_emul_set_lazy(int(lazy))        # lazy: a CU's own cost only arrives after all its children (the slowest possible depth wave)
    W, H = prm.pic_w, prm.pic_h
    wc, hc = (W + 63) // 64, (H + 63) // 64
    px = px_dtype(depth)
    y, u, v = (np.ascontiguousarray(a, px) for a in (y, u, v))
    ry, ru, rv = np.zeros((H, W), px), np.zeros((H // 2, W // 2), px), np.zeros((H // 2, W // 2), px)
    scu = np.zeros(hc * 16 * wc * 16, SCU_NP)
    co = np.zeros(wc * hc * 6144, np.int16)
    mo = np.zeros(wc * hc * 3 * N_MODELS, np.uint32)
    cp = ctu_params(prm)
    rc = lib.ctu_emul_search_picture(depth, ctypes.byref(cp), ptr(y), ptr(u), ptr(v), ptr(ry), ptr(ru), ptr(rv), ptr(scu), ptr(co), ptr(mo))
    assert rc == 0
    return search_result_from_device_layout(W, H, ry, ru, rv, scu, co, mo)


def ctu_crcs(res, W, H):
    """Per CTU CRC-32 of the in-picture cu fields + trees / reconstruction / levels / models after the coder, as
    tools/refcheck/make_ctu_goldens.py computes them from the reference's records."""
    import zlib
    wc, hc = (W + 63) // 64, (H + 63) // 64
    out = np.zeros((hc * wc, 4), np.uint32)
    c = np.ascontiguousarray
    for cy in range(hc):
        for cx in range(wc):
            x, y, k = cx * 64, cy * 64, cy * wc + cx
            hh, ww = min(64, H - y), min(64, W - x)
            out[k, 0] = zlib.crc32(c(res["cu"][y // 4:(y + hh) // 4, x // 4:(x + ww) // 4]).tobytes() +
                                   c(res["trees"][y // 4:(y + hh) // 4, x // 4:(x + ww) // 4]).tobytes())
            out[k, 1] = zlib.crc32(c(res["rec_y"][y:y + hh, x:x + ww]).tobytes() + c(res["rec_u"][y // 2:(y + hh) // 2, x // 2:(x + ww) // 2]).tobytes() +
                                   c(res["rec_v"][y // 2:(y + hh) // 2, x // 2:(x + ww) // 2]).tobytes())
            co = res["coeff"][k]
            out[k, 2] = zlib.crc32(c(co[:4096].reshape(64, 64)[:hh, :ww]).tobytes() + c(co[4096:].reshape(2, 32, 32)[:, :hh // 2, :ww // 2]).tobytes())
            out[k, 3] = zlib.crc32(c(res["models"][k, 2]).tobytes())
    return out


def scu_from_cu(cu, qp):
    """uvghip_scu_t / orc_scu table (one entry per 4x4, all rows of whole CTUs) from the compact cu fields [h4, w4, 11] of
    oracle_search_picture / the ref_ctu goldens: what the deblocking filter reads of an intra picture."""
    t = np.zeros(cu.shape[:2], SCU_NP)
    for j, f in enumerate(("type", "log2_width", "log2_height", "log2_chroma_width", "log2_chroma_height", "cbf")):
        t[f] = cu[:, :, j]
    t["luma_edges"], t["chroma_edges"] = cu[:, :, 8], cu[:, :, 9]
    t["qp"] = qp
    return t


def oracle_sao_picture(orc, depth, W, H, qp, lam, src, rec, scu, sao_type=3, slice_type=2):
    """orcN_sao_search_picture: per-CTU deblocking in the encoder's order, the SAO decision of every CTU on the block it sees at
    that moment, SAO of the deblocked picture.  rec: reconstruction before the in-loop filters (not modified).
    -> dict(sao [ctus, 2, 17], sao_models [ctus, 6], snap_y/u/v, final_y/u/v, deblocked_y/u/v)"""
    wc, hc = (W + 63) // 64, (H + 63) // 64
    px = px_dtype(depth)
    s = [np.ascontiguousarray(a, px) for a in src]
    r = [np.ascontiguousarray(a, px).copy() for a in rec]
    scu = np.ascontiguousarray(scu)
    info = np.zeros((wc * hc, 2, 17), np.int32)
    models = np.zeros((wc * hc, 6), np.uint16)
    snap = [np.zeros_like(a) for a in r]
    out = [np.zeros_like(a) for a in r]
    fn = orc.fn(depth, "sao_search_picture_slice")          # slice_type: 2 = I, 1 = P, 0 = B
    fn.restype = None
    fn(ptr(s[0]), ptr(s[1]), ptr(s[2]), ptr(r[0]), ptr(r[1]), ptr(r[2]), ctypes.c_int(W), ctypes.c_int(H), ptr(scu), ctypes.c_int(scu.shape[1]),
       ctypes.c_int(qp), ctypes.c_double(lam), ctypes.c_int(sao_type), ctypes.c_int(slice_type), ptr(info), ptr(models), ptr(snap[0]), ptr(snap[1]), ptr(snap[2]),
       ptr(out[0]), ptr(out[1]), ptr(out[2]))
    return dict(sao=info, sao_models=models, snap_y=snap[0], snap_u=snap[1], snap_v=snap[2], final_y=out[0], final_u=out[1], final_v=out[2],
                deblocked_y=r[0], deblocked_u=r[1], deblocked_v=r[2])


def sao_info_comparable(info):
    """sao_info_t records with the entries the encoder leaves undefined masked out: offsets[0] / offsets[5] of a band decision
    are copies of uninitialised stack (sao.c:453,478: temp_offsets[0], [5] are never written), the second half of a luma
    record's offsets likewise (one buffer), and merges pass both on."""
    a = info.copy()
    a[:, :, 7] = 0              # category 0 / the slot in front of the four band offsets: 0 for an edge decision, stack garbage for a
    a[:, :, 12] = 0             # band decision -- also when "nothing" then beat the band decision and only the type changed
    a[:, 0, 12:] = 0            # luma: one buffer, offsets[5..9] are whatever edge_offset[5..9] held (sao.c:373,436)
    a[:, 0, 6] = 0              # ... and band_position[1] of a luma band decision is never written (sao.c:462)
    return a


def oracle_count_bits(orc, depth, prm, res, range_in):
    """The hand-over consumer (orcN_count_picture_bits): a count-mode walk of uvg_encode_coding_tree over a search result in the
    layout of oracle_search_picture / search_result_from_device_layout -- cu fields + trees, levels, the models every CTU starts
    from -- with the arithmetic coder's range arithmetic.  range_in: the coder's range at every CTU's start.
    -> (bits [ctus], range afterwards [ctus], models afterwards [ctus, 1286])"""
    W, H = prm.pic_w, prm.pic_h
    wc, hc = (W + 63) // 64, (H + 63) // 64
    cu = np.zeros((hc * 16, wc * 16, 20), np.uint8)
    cu[:, :, :11] = res["cu"]
    cu[:, :, 12:] = np.ascontiguousarray(res["trees"].astype(np.uint32)).view(np.uint8).reshape(hc * 16, wc * 16, 8)
    co = np.ascontiguousarray(res["coeff"], np.int16)
    start = np.ascontiguousarray(res["models"][:, 0])
    rin = np.ascontiguousarray(range_in, np.int64)
    bits, rout = np.zeros(wc * hc, np.int64), np.zeros(wc * hc, np.int64)
    after = np.zeros((wc * hc, MODELS_BYTES), np.uint8)
    rc = orc.fn(depth, "count_picture_bits")(ctypes.byref(prm), ptr(cu), ptr(co), ptr(start), ptr(rin), ptr(bits), ptr(rout), ptr(after))
    assert rc == 0
    return bits, rout, after


def oracle_encode_ctus(orc, depth, prm, res, state_in):
    """orcN_encode_picture_ctus: the arithmetic coder itself over every CTU's coding tree (same inputs as oracle_count_bits;
    state_in [ctus, 5]: low, range, bits_left, num_buffered_bytes, buffered_byte when the tree begins).
    -> (state afterwards [ctus, 5], payload bytes, offsets [ctus + 1])"""
    W, H = prm.pic_w, prm.pic_h
    wc, hc = (W + 63) // 64, (H + 63) // 64
    cu = np.zeros((hc * 16, wc * 16, 20), np.uint8)
    cu[:, :, :11] = res["cu"]
    cu[:, :, 12:] = np.ascontiguousarray(res["trees"].astype(np.uint32)).view(np.uint8).reshape(hc * 16, wc * 16, 8)
    co = np.ascontiguousarray(res["coeff"], np.int16)
    start = np.ascontiguousarray(res["models"][:, 0])
    sin = np.ascontiguousarray(state_in, np.int64)
    sout = np.zeros((wc * hc, 5), np.int64)
    cap = 64 + wc * hc * 16384
    out = np.zeros(cap, np.uint8)
    off = np.zeros(wc * hc + 1, np.int64)
    fn = orc.fn(depth, "encode_picture_ctus")
    fn.restype = ctypes.c_long
    n = fn(ctypes.byref(prm), ptr(cu), ptr(co), ptr(start), ptr(sin), ptr(sout), ptr(out), ctypes.c_long(cap), ptr(off))
    assert n >= 0
    return sout, out[:n].copy(), off


def oracle_encode_rows(orc, depth, prm, res, sao):
    """orcN_encode_picture_rows: the slice data -- every WPP row's substream (SAO syntax + coding tree of every CTU, the end of the
    substream, emulation prevention) from the search's hand-over (cu fields, trees, levels) and the SAO decisions [ctus, 2, 17].
    -> (bytes, row offsets [rows + 1], the coder's models after every CTU [ctus, 1286])"""
    W, H = prm.pic_w, prm.pic_h
    wc, hc = (W + 63) // 64, (H + 63) // 64
    cu = np.zeros((hc * 16, wc * 16, 20), np.uint8)
    cu[:, :, :11] = res["cu"]
    cu[:, :, 12:] = np.ascontiguousarray(res["trees"].astype(np.uint32)).view(np.uint8).reshape(hc * 16, wc * 16, 8)
    co = np.ascontiguousarray(res["coeff"], np.int16)
    sa = np.ascontiguousarray(sao, np.int32)
    cap = 4096 + wc * hc * 20000
    out = np.zeros(cap, np.uint8)
    off = np.zeros(hc + 1, np.int64)
    after = np.zeros((wc * hc, MODELS_BYTES), np.uint8)
    fn = orc.fn(depth, "encode_picture_rows")
    fn.restype = ctypes.c_long
    n = fn(ctypes.byref(prm), ptr(cu), ptr(co), ptr(sa), ptr(out), ctypes.c_long(cap), ptr(off), ptr(after))
    assert n >= 0
    return out[:n].copy(), off, after


def oracle_encode_rows_no_sao(orc, depth, prm, res):
    """oracle_encode_rows with SAO off: no SAO syntax in the substreams."""
    W, H = prm.pic_w, prm.pic_h
    wc, hc = (W + 63) // 64, (H + 63) // 64
    cu = np.zeros((hc * 16, wc * 16, 20), np.uint8)
    cu[:, :, :11] = res["cu"]
    cu[:, :, 12:] = np.ascontiguousarray(res["trees"].astype(np.uint32)).view(np.uint8).reshape(hc * 16, wc * 16, 8)
    co = np.ascontiguousarray(res["coeff"], np.int16)
    cap = 4096 + wc * hc * 20000
    out, off, after = np.zeros(cap, np.uint8), np.zeros(hc + 1, np.int64), np.zeros((wc * hc, MODELS_BYTES), np.uint8)
    fn = orc.fn(depth, "encode_picture_rows")
    fn.restype = ctypes.c_long
    n = fn(ctypes.byref(prm), ptr(cu), ptr(co), None, ptr(out), ctypes.c_long(cap), ptr(off), ptr(after))
    assert n >= 0
    return out[:n].copy(), off, after


def filter_crcs(res, W, H):
    """Per CTU CRC-32 of (the block the SAO decision saw, the block of the final picture), Y + U + V, as
    tools/refcheck/make_ctu_goldens.py computes them (filter_crc)."""
    import zlib
    wc, hc = (W + 63) // 64, (H + 63) // 64
    out = np.zeros((wc * hc, 2), np.uint32)
    for k in range(wc * hc):
        y, x = (k // wc) * 64, (k % wc) * 64
        for j, pre in enumerate(("snap_", "final_")):
            out[k, j] = zlib.crc32(b"".join(np.ascontiguousarray(res[pre + n][(y >> c):(y >> c) + (64 >> c), (x >> c):(x >> c) + (64 >> c)]).tobytes()
                                            for n, c in (("y", 0), ("u", 1), ("v", 1))))
    return out


def ctu_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_source(g):
    """The synthetic source picture of a ref_ctu* golden; its CRC pins the generator."""
    import zlib
    from uvg266_amd import layout
    W, H, depth, qp, t = (int(a) for a in g["meta"][:5])
    y, u, v = varied_picture(W, H, t, depth)        # (t < 1000: layout.synthetic_yuv420 itself)
    assert zlib.crc32(y.tobytes() + u.tobytes() + v.tobytes()) == int(g["src_crc"]), "synthetic generator drifted from the golden's source"
    return W, H, depth, qp, y, u, v


# ---- golden container (written by tools/refcheck/refcheck.c) ----------------
_DT = {0: np.uint8, 1: np.uint16, 2: np.int16, 3: np.int32, 4: np.uint32, 5: np.int64, 6: np.float64}


def read_golden(group, depth):
    """-> list of (name, [arrays]) records of tests/golden/ref_<group>_<depth>.bin"""
    path = os.path.join(GOLDEN, f"ref_{group}_{depth}.bin")
    recs = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        magic, nl = struct.unpack_from("<II", data, pos); pos += 8
        assert magic == 0x52454631, "corrupt golden file"
        name = data[pos:pos + nl].decode(); pos += nl
        (na,) = struct.unpack_from("<I", data, pos); pos += 4
        arrs = []
        for _ in range(na):
            code, n = struct.unpack_from("<II", data, pos); pos += 8
            dt = np.dtype(_DT[code])
            arrs.append(np.frombuffer(data, dt, n, pos).copy()); pos += n * dt.itemsize
        recs.append((name, arrs))
    return recs


# ---- the reference's own KAT inputs (tests/satd_tests.c:60-106, tests/sad_tests.c:51-71) ----
def satd_kat_buffers(test, log_w):
    size = 1 << (2 * log_w)
    i = np.arange(size)
    if test == 0:      # black / white
        return np.zeros(size, np.uint8), np.full(size, 255, np.uint8)
    if test == 1:      # checkers: buf2 = (buf1 + 1) % 2
        b1 = (255 * ((((i >> log_w) % 2) + (i % 2)) % 2)).astype(np.uint8)
        b2 = ((b1.astype(np.int32) + 1) % 2).astype(np.uint8)
        return b1, b2
    col, row = i % (1 << log_w), i // (1 << log_w)
    r = np.floor(np.sqrt((row * row + col * col).astype(np.float64))).astype(np.int64)
    b1 = (255 // (r + 1)).astype(np.uint8)
    return b1, (255 - 255 // (r + 1)).astype(np.uint8)


SATD_KAT = {0: [510, 1020, 4080, 16320, 65280],       # tests/satd_tests.c:122
            1: [1278, 2556, 10224, 40896, 163584],     # tests/satd_tests.c:140
            2: [2728, 7158, 10775, 23399, 72780]}      # tests/satd_tests.c:159

SAD_REF_8x8 = np.array([1, 2, 2, 2, 2, 2, 2, 3] + [4, 5, 5, 5, 5, 5, 5, 6] * 6 + [7, 8, 8, 8, 8, 8, 8, 9],
                       np.uint8).reshape(8, 8) + 48     # tests/sad_tests.c:51-60,91-98
SAD_PIC_8x8 = np.full((8, 8), 1 + 48, np.uint8)         # tests/sad_tests.c:62-71,84-89
# (ref_x, ref_y) -> expected, tests/sad_tests.c:144-285
SAD_BORDER_KAT = [
    ((-3, -3), 1 * 16 + (2 + 4) * 16 + 5 * 16 - 64), ((0, -3), (1 + 3) * 4 + 2 * 24 + (4 + 6) * 4 + 5 * 24 - 64),
    ((3, -3), 3 * 16 + (2 + 6) * 16 + 5 * 16 - 64), ((-3, 0), (1 + 7) * 4 + 4 * 24 + (2 + 8) * 4 + 5 * 24 - 64),
    ((0, 0), (1 + 3 + 7 + 9) + (2 + 4 + 6 + 8) * 6 + 5 * 36 - 64), ((3, 0), (3 + 9) * 4 + 6 * 24 + (2 + 8) * 4 + 5 * 24 - 64),
    ((-3, 3), 7 * 16 + (4 + 8) * 16 + 5 * 16 - 64), ((0, 3), (7 + 9) * 4 + 8 * 24 + (4 + 6) * 4 + 5 * 24 - 64),
    ((3, 3), 9 * 16 + (6 + 8) * 16 + 5 * 16 - 64),
    ((-10, -10), 1 * 64 - 64), ((0, -10), (1 + 3) * 8 + 2 * 48 - 64), ((10, -10), 3 * 64 - 64),
    ((-10, 0), (1 + 7) * 8 + 4 * 48 - 64), ((10, 0), (3 + 9) * 8 + 6 * 48 - 64),
    ((-10, 10), 7 * 64 - 64), ((0, 10), (7 + 9) * 8 + 8 * 48 - 64), ((10, 10), 9 * 64 - 64),
]
# tests/sad_tests.c:392-399
SAD_DIMS = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16),
            (8, 4), (4, 8), (48, 16), (16, 48), (24, 16), (16, 24), (12, 4), (4, 12)]


def sad_big_planes():
    """tests/sad_tests.c:100-118: 64x64 g_big_pic / g_big_ref"""
    i = np.arange(64 * 64, dtype=np.int64)
    return (((i * i // 32 + i) % 255).astype(np.uint8).reshape(64, 64),
            ((i * i // 16 + i) % 255).astype(np.uint8).reshape(64, 64))


def dct_test_gradient(width=64):
    """tests/dct_tests.c:68-78,88-91: init_gradient(width, width, width, 255/width, buf)"""
    y, x = np.mgrid[0:width, 0:width]
    slope = 255 // width
    val = (slope * np.sqrt(((width - x) ** 2 + (width - y) ** 2).astype(np.float64)) + 0.5).astype(np.int64)
    return np.clip(val, 0, 255).astype(np.int16)


def intra_golden_blocks(depth):
    """Yield (frame, x, y, n, at, al, orig, preds(67,n*n), costs(67)) from ref_intra_<depth>.bin,
    with the dumped crop pasted into a zero frame (only the crop is ever read as reference)."""
    for name, arrs in read_golden("intra", depth):
        if name != "block":
            continue
        meta, crop, orig, preds, costs = arrs
        FW, FH, x, y, n, at, al, cx0, cy0, cw, ch = [int(v) for v in meta[:11]]
        frame = np.zeros((FH, FW), crop.dtype)
        frame[cy0:cy0 + ch, cx0:cx0 + cw] = crop.reshape(ch, cw)
        yield frame, x, y, n, at, al, orig, preds.reshape(67, n * n), costs


def zorder_avail(x, y, n, pic_w, pic_h, ctu=64):
    """Availability a z-order (quad-tree only) encoder would report for an n x n block at (x,y):
    (avail_top, avail_left) in samples, with the reference's limits (2n, picture edge)."""
    def z(sx, sy):
        v = 0
        for b in range(4):
            v |= ((sx >> b) & 1) << (2 * b) | ((sy >> b) & 1) << (2 * b + 1)
        return v
    lx, ly = x % ctu, y % ctu
    zc = z(lx // 4, ly // 4)

    def coded(px, py):
        if px < 0 or py < 0 or px >= pic_w or py >= pic_h:
            return False
        cx, cy = (px // ctu) * ctu, (py // ctu) * ctu
        if cy < y - ly or (cy == y - ly and cx < x - lx):
            return cx <= x - lx + ctu or cy < y - ly     # CTUs above (incl. above-right) and to the left
        if cx == x - lx and cy == y - ly:
            return z((px - cx) // 4, (py - cy) // 4) < zc
        return False
    at = 0
    while at < 2 * n and coded(x + at, y - 1):
        at += 4
    al = 0
    while al < 2 * n and coded(x - 1, y + al):
        al += 4
    return min(at, 2 * n, pic_w - x), min(al, 2 * n, pic_h - y)


# ---- records of the encoder-tree shim run against recording stand-ins (tools/refcheck/rc_shim.inc) ----
def shim_goldens(depth):
    """-> dict of lists: 'sq' (quant / dequant), 'sqr' (quantize_residual), 'sbp' (bipred plane calls).  The views are the bytes
    the shim extracted from the reference's encoder_state_t / cu_info_t; the expectations are the generic strategies' results."""
    from uvg266_amd.lib import StateView, CuView
    px = px_dtype(depth)
    out = {"sq": [], "sqr": [], "sbp": [], "sjc": []}
    for name, a in read_golden("shim", depth):
        if name == "sq":
            m = [int(v) for v in a[0]]
            out["sq"].append(dict(inverse=m[0], w=m[1], h=m[2], color=m[3], scan_idx=m[4], block_type=m[5], ts=m[6], lfnst=m[7],
                                  sv=StateView.from_buffer_copy(a[1].tobytes()), src=a[2], want=a[3]))
        elif name == "sqr":
            m = [int(v) for v in a[0]]
            w, h, si, so = m[0], m[1], m[5], m[6]
            out["sqr"].append(dict(w=w, h=h, color=m[2], scan_order=m[3], trskip=m[4], in_stride=si, out_stride=so, early_skip=m[7],
                                   lmcs_adj=m[8], tree=m[9], has=m[10], branch=m[11], sv=StateView.from_buffer_copy(a[1].tobytes()),
                                   cv=CuView.from_buffer_copy(a[2].tobytes()), ref=a[3].view(px) if a[3].dtype != px else a[3],
                                   pred=a[4], q=a[5], rec=a[6]))
        elif name == "sjc":
            m = [int(v) for v in a[0]]
            out["sjc"].append(dict(w=m[0], h=m[1], scan_order=m[2], in_stride=m[3], out_stride=m[4], early_skip=m[5], lmcs_adj=m[6], tree=m[7],
                                   ret=m[8], S=m[9], sv=StateView.from_buffer_copy(a[1].tobytes()), cv=CuView.from_buffer_copy(a[2].tobytes()),
                                   uref=a[3], vref=a[4], upred=a[5], vpred=a[6], q=a[7], urec=a[8], vrec=a[9]))
        elif name == "sbp":
            m = [int(v) for v in a[0]]
            out["sbp"].append(dict(stride=m[0], i0=m[1], i1=m[2], w=m[3], h=m[4], l0=a[1], l1=a[2], want=a[3]))
    return out


def coeffcost_goldens(depth):
    """-> list of dicts: w, h, color, flags, style, models (1220 bytes), coeff (h*w,) int16, bits, after (1220 bytes)."""
    out = []
    for name, a in read_golden("coeffcost", depth):
        if name == "cc":
            m = [int(v) for v in a[0]]
            out.append(dict(w=m[0], h=m[1], color=m[2], flags=m[3], style=m[4], models=a[1], coeff=a[2], bits=float(a[3][0]), after=a[4]))
    return out


def jccr_goldens(depth):
    """uvg_quant_cbcr_residual records -> list of dicts (planes reshaped to (rows, stride))."""
    out = []
    for name, a in read_golden("jccr", depth):
        if name != "jccr":
            continue
        m = [int(v) for v in a[0]]
        S, so = m[11], m[12]
        out.append(dict(w=m[0], h=m[1], joint=m[2], sign=m[3], qps=m[4], intra=m[5], cu_type=m[6], rdoq=m[7], rdoq_skip=m[8], cbf_u=m[9],
                        early_skip=m[10], S=S, so=so, ret=m[13], lam=float(a[1][0]), ctx=a[2], uref=a[3].reshape(S, S), vref=a[4].reshape(S, S),
                        upred=a[5].reshape(S, S), vpred=a[6].reshape(S, S), q=a[7], urec=a[8], vrec=a[9]))
    return out


def signhide_goldens(depth):
    """cfg.signhide_enable = 1 records -> list of dicts; kind 0 = uvg_rdoq, 1 = uvg_quant."""
    out = []
    for name, a in read_golden("signhide", depth):
        if name != "sh":
            continue
        m = [int(v) for v in a[0]]
        if m[0] == 0:
            out.append(dict(kind=0, w=m[1], h=m[2], color=m[3], qps=m[4], intra=m[5], cbf_u=m[6], lfnst=m[7], mts=m[8], cu_type=m[9],
                            lam=float(a[1][0]), ctx=a[2], coef=a[3], q=a[4]))
        else:
            out.append(dict(kind=1, w=m[1], h=m[2], color=m[3], qps=m[4], intra=m[5], ts=m[6], lfnst=m[7], coef=a[3], q=a[4]))
    return out


def varied_picture(W, Hh, t, depth):
    """Pictures for parity sweeps.  t < 1000: uvg266_amd.layout.synthetic_yuv420 of seed t.  t >= 1000: that picture plus noise
    (t // 1000: 1 -> +-4, 2 -> +-32, 3 -> white noise over the full range, 4 -> sparse impulses on flat grey: lone large coefficients)."""
    from uvg266_amd import layout
    y, u, v = layout.synthetic_yuv420(W, Hh, t % 1000, depth)
    kind = t // 1000
    if kind == 0:
        return y, u, v
    rng = np.random.default_rng(t)
    top = (1 << depth) - 1
    out = []
    for p in (y, u, v):
        if kind == 3:
            q = rng.integers(0, top + 1, p.shape)
        elif kind == 4:
            q = np.full(p.shape, 1 << (depth - 1), np.int64)
            m = rng.random(p.shape) < 0.01
            q[m] = rng.integers(0, top + 1, int(m.sum()))
        else:
            a = (4 if kind == 1 else 32) << (depth - 8)
            q = p.astype(np.int64) + rng.integers(-a, a + 1, p.shape)
        out.append(np.clip(q, 0, top).astype(p.dtype))
    return tuple(out)


def sweep_cases(n, seed, widths=(64, 72, 128, 136, 192, 200, 256, 264, 320), heights=(64, 72, 128, 136, 192)):
    """n (W, H, depth, qp, t) combinations: sizes with 8-sample CTUs at the edges, both depths, QP 0..51, every picture kind."""
    import random
    rng = random.Random(seed)
    return [(rng.choice(list(widths)), rng.choice(list(heights)), rng.choice([8, 10]),
             rng.choice([0, 3, 10, 17, 22, 27, 32, 37, 45, 51]), rng.randrange(0, 64) + 1000 * rng.choice([0, 1, 2, 3, 4])) for _ in range(n)]


# ---- the picture's NAL units behind the parameter sets (checker for uvghip_write_picture_nals / uvghip_picture_checksum) ----
def picture_checksum(plane, depth):
    """uvg_image_checksum of one plane (src/strategies/generic/nal-generic.c:68-92): sum of (byte ^ mask) over the samples' bytes."""
    h, w = plane.shape
    yy, xx = np.mgrid[0:h, 0:w]
    mask = ((xx & 0xff) ^ (yy & 0xff) ^ (xx >> 8) ^ (yy >> 8)) & 0xff
    p = plane.astype(np.int64)
    s = ((p & 0xff) ^ mask).sum()
    if depth > 8:
        s += (((p >> 8) & 0xff) ^ mask).sum()
    return int(s) & 0xffffffff


def picture_nals(row_sizes, rows, checksums, poc=0, sao=True):
    """Slice NAL (header with the entry points + the rows) and the decoded-picture-hash SEI as the encoder writes them for an IDR
    picture of an all-intra stream (src/encoder_state-bitstream.c:993-1139, 1248-1477; src/nal.c:43-74; emulation prevention of
    src/bitstream.c:215-226), restated bit by bit.  rows: list of bytes objects.  -> bytes"""
    def ue(v):
        v += 1
        n = v.bit_length()
        return "0" * (n - 1) + format(v, "b")

    def payload(bits):
        assert len(bits) % 8 == 0
        out, zeros = bytearray(), 0
        for i in range(0, len(bits), 8):
            b = int(bits[i:i + 8], 2)
            if zeros == 2 and b < 4:
                out.append(3)
                zeros = 0
            zeros = zeros + 1 if b == 0 else 0
            out.append(b)
        return bytes(out)

    bits = "1" + "1" + "0" + "0" + "0" + ue(0) + format(poc & 15, "04b") + "0" + "1" + ("11" if sao else "")
    if len(row_sizes) > 1:
        ol = int(max(row_sizes)).bit_length()
        bits += ue(ol - 1) + "".join(format(int(s) - 1, "0%db" % ol) for s in row_sizes[:-1])
    bits += "1"
    bits += "0" * (-len(bits) % 8)
    out = (b"\x00\x00\x01\x00\x41" if poc == 0 else b"\x00\x00\x00\x01\x00\x39") + payload(bits) + b"".join(rows)
    if checksums is not None:
        sei = format(132, "08b") + format(14, "08b") + format(2, "08b") + format(0, "08b") + "".join(format(int(c), "032b") for c in checksums) + "10000000"
        out += b"\x00\x00\x01\x00\xc1" + payload(sei)
    return out


# ---- reconstruction of the encoder's inter decisions (tests/test_oracle_inter_recon.py, tests/test_gpu_inter_recon.py) ----
class OracleBlocks:
    """The block functions of the oracle behind the three operations inter reconstruction needs."""
    def __init__(self, orc, depth):
        self.orc, self.depth = orc, depth

    def picture(self, planes):
        return planes

    def sample(self, plane, pw, ph, x0, y0, w, h, fx, fy, chroma, hi):
        return self.orc.ipol_sample(self.depth, plane, pw, ph, x0, y0, w, h, fx, fy, chroma=chroma, hi=hi)

    def average(self, a, b, w, h):
        return self.orc.bipred_average(self.depth, np.ascontiguousarray(a), np.ascontiguousarray(b), w, h)

    def residual(self, levels, n, qp, color):
        d = self.depth
        qs = self.orc.fn(d, "get_scaled_qp", int)(color, qp, 6 * (d - 8), None)
        co = self.orc.dequant(d, np.ascontiguousarray(levels.reshape(-1)), n, n, d, qs, 0)
        out = np.zeros(n * n, np.int16)
        self.orc.fn(d, "idct_nxn", None)(d, n, ptr(co), ptr(out))
        return out.reshape(n, n).astype(np.int32)


def inter_reconstruct(g, B):
    """Every inter CU of a ref_inter_* golden: motion compensation as uvg_inter_pred_pu does it (src/inter.c:400-530, 532-602, 685-748:
    integer copies, fractional samplers, high-precision intermediates + uvg_bipred_average for bi-prediction, border replication)
    + dequantised, inverse-transformed levels, through the block functions B -> asserts the encoder's reconstruction before the
    in-loop filters; returns what was exercised."""
    W, Hh, depth, qp0, frames = (int(a) for a in g["dims"])
    top = (1 << depth) - 1
    final = [B.picture((g["final_y"][f], g["final_u"][f], g["final_v"][f])) for f in range(frames)]
    seen = dict(inter=0, bi=0, frac=0, resid=0, outside=0)

    def predict_list(ref, x, y, n, mv, want_hi):
        int_x = (mv[0] + 7) >> 4 if mv[0] >= 0 else (mv[0] + 8) >> 4        # uvg_change_precision_vector2d(INTERNAL_MV_PREC, 0)
        int_y = (mv[1] + 7) >> 4 if mv[1] >= 0 else (mv[1] + 8) >> 4
        frac_l = (mv[0] & 15) != 0 or (mv[1] & 15) != 0
        frac_c = (int_x & 1) != 0 or (int_y & 1) != 0
        if frac_l:
            luma = B.sample(ref[0], W, Hh, x + (mv[0] >> 4), y + (mv[1] >> 4), n, n, mv[0] & 15, mv[1] & 15, False, want_hi)
        else:
            luma = B.sample(ref[0], W, Hh, x + int_x, y + int_y, n, n, 0, 0, False, False)
        c = n // 2
        if frac_l or frac_c:
            cx, cy = x // 2 + (mv[0] >> 5), y // 2 + (mv[1] >> 5)
            u = B.sample(ref[1], W // 2, Hh // 2, cx, cy, c, c, mv[0] & 31, mv[1] & 31, True, want_hi)
            v = B.sample(ref[2], W // 2, Hh // 2, cx, cy, c, c, mv[0] & 31, mv[1] & 31, True, want_hi)
        else:
            cx, cy = (x + int_x) // 2, (y + int_y) // 2
            u = B.sample(ref[1], W // 2, Hh // 2, cx, cy, c, c, 0, 0, True, False)
            v = B.sample(ref[2], W // 2, Hh // 2, cx, cy, c, c, 0, 0, True, False)
        return luma, u, v

    for k in range(len(g["meta"])):
        fr, x0, y0, qp = (int(a) for a in g["meta"][k][:4])
        refs = g["refs"][k]
        pocs = refs[1:1 + refs[0]]
        lists = [refs[19:19 + refs[17]], refs[35:35 + refs[18]]]
        cu, mot, co = g["cu"][k], g["motion"][k], g["coeff"][k]
        for i in range(256):
            lx, ly = (i & 15) * 4, (i >> 4) * 4
            c = cu[i]
            if c[0] != 2:
                continue
            n = 1 << int(c[1])
            if (lx & (n - 1)) or (ly & (n - 1)):
                continue                               # not the CU's first unit
            assert int(c[2]) == int(c[1])              # square CUs only in this configuration
            x, y = x0 + lx, y0 + ly
            m = [int(a) for a in mot[i]]
            mvs, ridx, mdir = [(m[0], m[1]), (m[2], m[3])], [m[4], m[5]], m[6]
            used = [l for l in (0, 1) if mdir & (1 << l)]
            preds = []
            for l in used:
                poc = int(pocs[int(lists[l][ridx[l]])])
                preds.append(predict_list(final[poc], x, y, n, mvs[l], len(used) == 2))
                seen["frac"] += (mvs[l][0] & 15) != 0
                seen["outside"] += x + (mvs[l][0] >> 4) < 0 or y + (mvs[l][1] >> 4) < 0 or x + (mvs[l][0] >> 4) + n > W
            if len(used) == 2:
                pred = [B.average(a, b, n >> (p > 0), n >> (p > 0)) for p, (a, b) in enumerate(zip(*preds))]
                seen["bi"] += 1
            else:
                pred = list(preds[0])
            seen["inter"] += 1
            # residual: one transform unit per CU up to 32x32, four 32x32 units in a 64x64 CU (each with its own flags)
            tn = min(n, 32)
            rec = [np.asarray(p).reshape(n >> (j > 0), n >> (j > 0)).astype(np.int32) for j, p in enumerate(pred)]
            for ty in range(0, n, tn):
                for tx in range(0, n, tn):
                    t = cu[((ly + ty) >> 2) * 16 + ((lx + tx) >> 2)]
                    cbf = int(t[5])
                    if cbf & 1:
                        lv = co[:4096].reshape(64, 64)[ly + ty:ly + ty + tn, lx + tx:lx + tx + tn]
                        rec[0][ty:ty + tn, tx:tx + tn] += B.residual(lv, tn, int(t[10]), 0)
                        seen["resid"] += 1
                    for j in (1, 2):
                        if cbf & (1 << j):
                            lv = co[4096 + (j - 1) * 1024:4096 + j * 1024].reshape(32, 32)[(ly + ty) // 2:(ly + ty + tn) // 2, (lx + tx) // 2:(lx + tx + tn) // 2]
                            rec[j][ty // 2:(ty + tn) // 2, tx // 2:(tx + tn) // 2] += B.residual(lv, tn // 2, int(t[10]), j)
            want = (g["rec_y"][fr][y:y + n, x:x + n], g["rec_u"][fr][y // 2:(y + n) // 2, x // 2:(x + n) // 2], g["rec_v"][fr][y // 2:(y + n) // 2, x // 2:(x + n) // 2])
            for j in range(3):
                assert np.array_equal(np.clip(rec[j], 0, top), want[j]), (fr, x, y, n, "YUV"[j], m)
    return seen


def inter_scu_table(g, frame):
    """The uvghip_scu_t / orc_scu table of one picture of a ref_inter_* golden: what the deblocking filter reads of a P / B picture
    (type, sizes, coded flags, edge flags, QP, motion; ref_id = the POC of the reference picture, -1 where a list is unused)."""
    W, Hh = int(g["dims"][0]), int(g["dims"][1])
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    t = np.zeros((hc * 16, wc * 16), SCU_NP)
    for k in range(len(g["meta"])):
        fr, x0, y0 = (int(a) for a in g["meta"][k][:3])
        if fr != frame:
            continue
        refs = g["refs"][k]
        pocs = refs[1:1 + refs[0]]
        lists = [refs[19:19 + refs[17]], refs[35:35 + refs[18]]]
        cu, mot = g["cu"][k].reshape(16, 16, 12), g["motion"][k].reshape(16, 16, 8)
        blk = t[y0 // 4:y0 // 4 + 16, x0 // 4:x0 // 4 + 16]
        for j, f in enumerate(("type", "log2_width", "log2_height", "log2_chroma_width", "log2_chroma_height", "cbf")):
            blk[f] = cu[:, :, j]
        blk["luma_edges"], blk["chroma_edges"], blk["qp"] = cu[:, :, 8], cu[:, :, 9], cu[:, :, 10].astype(np.int8)
        blk["mv_dir"] = mot[:, :, 6]
        mv = np.zeros((16, 16, 2, 2), np.int32)
        mv[:, :, 0, 0], mv[:, :, 0, 1], mv[:, :, 1, 0], mv[:, :, 1, 1] = mot[:, :, 0], mot[:, :, 1], mot[:, :, 2], mot[:, :, 3]
        blk["mv"] = mv
        rid = np.full((16, 16, 2), -1, np.int16)
        for l in (0, 1):
            use = (mot[:, :, 6] & (1 << l)) != 0
            if len(lists[l]):
                poc_of = np.array([int(pocs[int(i)]) for i in lists[l]], np.int16)
                rid[:, :, l][use] = poc_of[np.clip(mot[:, :, 4 + l][use], 0, len(poc_of) - 1)]
        blk["ref_id"] = rid
    return t


_MOVING_BASE = {}


def _moving_base(W, H, depth):
    """The 4x larger picture moving_picture's windows look into: the same for every picture of a sequence (kept for the last geometry)."""
    key = (W, H, depth)
    if key not in _MOVING_BASE:
        _MOVING_BASE.clear()
        _MOVING_BASE[key] = varied_picture(4 * (W + 32), 4 * (H + 32), 2007, depth)
    return _MOVING_BASE[key]


def moving_picture(W, H, t, depth):
    """Picture t of a sequence with fractional motion: a window into a 4x larger noisy picture, shifted by quarter samples per
    picture and box-filtered down -- the left and the right half move differently (partitions, uni- and bi-prediction)."""
    if W > 1920 or H > 1088:
        # above 1080p the window would look into a 16x larger picture (minutes on the GPU box's host): the 960x544 sequence tiled instead
        # (same motion everywhere, tile seams are just more edges)
        tile = moving_picture(960, 544, t, depth)
        return tuple(np.ascontiguousarray(np.tile(p, ((H + 543) // 544, (W + 959) // 960))[:H >> c, :W >> c]) for p, c in zip(tile, (0, 1, 1)))
    base = _moving_base(W, H, depth)
    out = []
    for b, c in zip(base, (0, 1, 1)):
        w, h = W >> c, H >> c

        def window(sx, sy):
            a = b[sy:sy + 4 * h, sx:sx + 4 * w].astype(np.int32)
            return ((a.reshape(h, 4, w, 4).sum(axis=(1, 3)) + 8) >> 4).astype(b.dtype)
        p = window((40 + 5 * t) >> c, (40 + 3 * t) >> c)
        p[:, w // 2:] = window((100 - 7 * t) >> c, (40 + 2 * t) >> c)[:, w // 2:]
        out.append(p)
    return tuple(out)


def clip_picture(W, H, t, depth):
    """Picture t of an arbitrarily long clip: moving_picture's motion run forth and back (the window stays inside its base picture);
    the first 13 pictures are moving_picture's."""
    return moving_picture(W, H, 12 - abs(t % 24 - 12), depth)


def plateau_picture(W, H, t, depth):
    """clip_picture with the samples of every plane posterised to plateaus of 32 (8 bit): flat areas whose intra prediction is exact, so that
    intra and inter candidates of a P / B CU compete at costs near zero -- where cfg.rdo 0 and 1 part ways (search.c:1413-1419)."""
    sh = depth - 3
    return tuple(np.ascontiguousarray(((p.astype(np.int32) >> sh) << sh).astype(p.dtype)) for p in clip_picture(W, H, t, depth))


_RISING_BASE = {}


def rising_picture(W, H, t, depth):
    """Picture t (0..12) of a sequence whose content moves UP faster and faster, 6 t samples per picture (the window runs down a tall base):
    the search follows it through its predictors, and its candidates come to reach more than a CTU row below the block -- where an encoder with
    frames in flight (cfg.owf) may not look (search_inter.c:94-149): the content on which --owf 0 and --owf 1 write different streams."""
    key = (W, H, depth)
    pos = lambda k: 3 * k * (k + 1)
    if key not in _RISING_BASE:
        _RISING_BASE.clear()
        _RISING_BASE[key] = varied_picture(4 * (W + 32), 4 * (H + 32) + 4 * pos(12) + 64, 2011, depth)
    base = _RISING_BASE[key]
    out = []
    for b, c in zip(base, (0, 1, 1)):
        w, h = W >> c, H >> c
        sx, sy = 40 >> c, (40 + 4 * pos(t)) >> c
        a = b[sy:sy + 4 * h, sx:sx + 4 * w].astype(np.int32)
        out.append(((a.reshape(h, 4, w, 4).sum(axis=(1, 3)) + 8) >> 4).astype(b.dtype))
    return tuple(out)


CLIP_GENERATORS = {0: None, 1: "clip_picture", 2: "plateau_picture", 3: "rising_picture"}          # the `clip` key of a ref_inter_* golden


# ---- P / B pictures: the inter search of the oracle (oracle/orc_search.c + orc_search_inter.inc) ---------------------------------
class InterFrame(ctypes.Structure):
    """orc_inter_frame: the picture's reference lists and reference pictures, as the encoder's frame-level bookkeeping hands them over."""
    _fields_ = [("slice_type", ctypes.c_int32), ("poc", ctypes.c_int32), ("n_refs", ctypes.c_int32), ("ref_pocs", ctypes.c_int32 * 16),
                ("l_size", ctypes.c_int32 * 2), ("l", (ctypes.c_int32 * 16) * 2),
                ("tmvp", ctypes.c_int32), ("max_merge", ctypes.c_int32), ("merge_level", ctypes.c_int32), ("bipred", ctypes.c_int32),
                ("fme_level", ctypes.c_int32), ("early_skip", ctypes.c_int32), ("depth_inter_min", ctypes.c_int32), ("depth_inter_max", ctypes.c_int32),
                ("ref_cu_stride", ctypes.c_int32), ("frame_qp", ctypes.c_int32),
                ("ref_y", ctypes.c_void_p * 16), ("ref_u", ctypes.c_void_p * 16), ("ref_v", ctypes.c_void_p * 16), ("ref_cu", ctypes.c_void_p * 16),
                ("owf", ctypes.c_int32), ("owf_margin", ctypes.c_int32)]          # (frames in flight: vectors restricted to what is final in the reference, oracle only)


MODELS_INTER_BYTES = 18 * 5


def ref_cu_table(cu, motion, pocs_of_lists):
    """The per-4x4 table of a reference picture the inter search reads: [type, mv[2][2], mv_dir, the POC the L0 / L1 vector points to].
    cu [h4, w4, >= 1] (type in [..., 0]), motion [h4, w4, 8] (the 'inter' array of ctu_dump.c), pocs_of_lists: ([POC per L0 index], [.. L1])."""
    h4, w4 = cu.shape[:2]
    t = np.zeros((h4, w4, 8), np.int32)
    t[:, :, 0] = cu[:, :, 0]
    t[:, :, 1:5] = motion[:, :, 0:4]
    t[:, :, 5] = motion[:, :, 6]
    t[:, :, 6:8] = -1
    inter = cu[:, :, 0] == 2
    for l in (0, 1):
        lut = np.asarray(list(pocs_of_lists[l]) + [-1] * 256, np.int32)
        use = inter & ((motion[:, :, 6] & (1 << l)) != 0)
        t[:, :, 6 + l] = np.where(use, lut[np.clip(motion[:, :, 4 + l], 0, 255)], -1)
    return np.ascontiguousarray(t)


def oracle_search_inter_picture(orc, depth, prm, fr, y, u, v, keep):
    """orcN_search_inter_picture -> dict like oracle_search_picture + motion [h16, w16, 8], extra [h16, w16, 4], models_inter [ctus, 3, 90].
    `keep`: the arrays fr's pointers refer to (held alive by the caller)."""
    W, H = prm.pic_w, prm.pic_h
    wc, hc = (W + 63) // 64, (H + 63) // 64
    px = px_dtype(depth)
    y, u, v = (np.ascontiguousarray(a, px) for a in (y, u, v))
    ry, ru, rv = np.zeros((H, W), px), np.zeros((H // 2, W // 2), px), np.zeros((H // 2, W // 2), px)
    cu = np.zeros((hc * 16, wc * 16, 20), np.uint8)
    co = np.zeros((wc * hc, 6144), np.int16)
    mo = np.zeros((wc * hc, 3, MODELS_BYTES), np.uint8)
    mot = np.zeros((hc * 16, wc * 16, 8), np.int32)
    ext = np.zeros((hc * 16, wc * 16, 4), np.uint8)
    mi = np.zeros((wc * hc, 3, MODELS_INTER_BYTES), np.uint8)
    rc = orc.fn(depth, "search_inter_picture")(ctypes.byref(prm), ctypes.byref(fr), ptr(y), ptr(u), ptr(v), ptr(ry), ptr(ru), ptr(rv), ptr(cu), ptr(co), ptr(mo),
                                               ptr(mot), ptr(ext), ptr(mi))
    assert rc == 0
    trees = np.ascontiguousarray(cu[:, :, 12:]).view(np.uint32).reshape(hc * 16, wc * 16, 2)
    return dict(rec_y=ry, rec_u=ru, rec_v=rv, cu=cu[:, :, :11].copy(), trees=trees, coeff=co, models=mo, motion=mot, extra=ext, models_inter=mi)


def iter_inter_frames(W, H, P):
    """The P / B pictures of a golden's records in coding order with their frame-level state -> (frame, record dict, SearchParams,
    InterFrame, keep-alive list).  References are the ENCODER's output pictures and side information."""
    wc, hc = (W + 63) // 64, (H + 63) // 64
    by_poc = {}
    for fr in sorted(P):
        d = P[fr]
        refs = d["refs"]
        n_refs, pocs = int(refs[0]), [int(a) for a in refs[1:17]]
        lsz = [int(refs[17]), int(refs[18])]
        lists = [[int(a) for a in refs[19:35]], [int(a) for a in refs[35:51]]]
        poc, slice_type = int(refs[51]), int(d["meta"][6])
        lam = d["lam"]
        cfg = [int(a) for a in d.get("cfg", (1, 6, 2, 1, 4, 1))]
        rd = cfg[6] if len(cfg) > 6 else 0          # cfg.rdo of the run (0: --preset medium, 1: --preset slow)
        owf = cfg[7] if len(cfg) > 7 else 0         # cfg.owf != 0: vectors restricted to what is final in the reference picture (oracle only so far)
        prm = SearchParams(W, H, int(d["meta"][3]), int(d["meta"][3]), 1, 4, 1, 1, 2, rd, float(lam[0]), float(lam[1]), float(lam[2]), float(lam[3]), float(lam[4]))
        F = InterFrame()
        F.slice_type, F.poc, F.n_refs = slice_type, poc, n_refs
        for i in range(16):
            F.ref_pocs[i] = pocs[i]
            F.l[0][i], F.l[1][i] = lists[0][i], lists[1][i]
        F.l_size[0], F.l_size[1] = lsz
        F.tmvp, F.max_merge, F.merge_level, F.bipred, F.fme_level, F.early_skip = cfg[:6]
        F.owf, F.owf_margin = int(owf != 0), 10       # (SAO_DELAY_PX: the goldens' runs have SAO on)
        F.depth_inter_min, F.depth_inter_max = 0, 3
        F.ref_cu_stride, F.frame_qp = wc * 16, int(d["meta"][7])
        keep = []
        for i in range(n_refs):
            rp = by_poc[pocs[i]]
            planes = [np.ascontiguousarray(p) for p in rp["final"]]
            keep += planes + [rp["ref_cu"]]
            F.ref_y[i], F.ref_u[i], F.ref_v[i] = (p.ctypes.data for p in planes)
            F.ref_cu[i] = rp["ref_cu"].ctypes.data
        own_pocs = ([pocs[lists[0][i]] for i in range(lsz[0])], [pocs[lists[1][i]] for i in range(lsz[1])])
        d["ref_cu"] = ref_cu_table(d["cu"], d["motion"], own_pocs)
        by_poc[poc] = d
        yield fr, d, prm, F, keep


def frame_states_from_records(meta, lam, refs):
    """Per-picture rows of a golden (meta [8], lam [6], refs [52] as tools/refcheck/ctu_dump.c writes them) -> the dicts api.LowDelayLoop takes."""
    out = []
    for m, l, r in zip(meta, lam, refs):
        out.append(dict(slice_type=int(m[6]), poc=int(r[51]), qp=int(m[3]), lam=float(l[0]), lam_sqrt=float(l[1]), c_lam=float(l[2]), cw_u=float(l[3]), cw_v=float(l[4]),
                        frame_qp=int(m[7]), n_refs=int(r[0]), ref_pocs=[int(a) for a in r[1:17]], l_size=[int(r[17]), int(r[18])],
                        lists=[[int(a) for a in r[19:35]], [int(a) for a in r[35:51]]]))
    return out


INTER4_NP = np.dtype([("skipped", "u1"), ("merged", "u1"), ("merge_idx", "u1"), ("root_cbf", "u1"), ("mv_cand0", "u1"), ("mv_cand1", "u1"),
                      ("mv_ref0", "u1"), ("mv_ref1", "u1")])      # uvghip_inter4_t
_EMUL_PB = None


def emul_search_inter_picture(depth, prm, F, y, u, v, leafwave=0, depthwave=0, lazy=0):
    """tests/emul/ctu_pb_emul.cpp: the P / B CTU search kernel's source on the host -> the device-layout outputs as a dict.
    leafwave: the order of work of the two-wave build (the 4x4 CUs of an 8x8 area on the leaf wave, all four, beside the area's unsplit CU)."""
    global _EMUL_PB
    if _EMUL_PB is None:
        d = os.path.join(ROOT, "tests", "emul")
        subprocess.check_call(["make", "-s", "-C", d])
        _EMUL_PB = ctypes.CDLL(os.path.join(d, "_build", "libctu_pb_emul.so"))
    _EMUL_PB.ctu_pb_emul_set_leafwave(int(leafwave))
    _EMUL_PB.ctu_pb_emul_set_depthwave(int(depthwave), int(lazy))      # (the three-wave build's order of work; lazy: evaluations come in only when waited for)
    W, H = prm.pic_w, prm.pic_h
    wc, hc = (W + 63) // 64, (H + 63) // 64
    px = px_dtype(depth)
    y, u, v = (np.ascontiguousarray(a, px) for a in (y, u, v))
    ry, ru, rv = np.zeros((H, W), px), np.zeros((H // 2, W // 2), px), np.zeros((H // 2, W // 2), px)
    n4 = hc * 16 * wc * 16
    scu, i4, trees, mot = np.zeros(n4, SCU_NP), np.zeros(n4, INTER4_NP), np.zeros(n4, np.uint32), np.zeros((n4, 8), np.int32)
    co = np.zeros(wc * hc * 6144, np.int16)
    mo, mi = np.zeros(wc * hc * 3 * N_MODELS, np.uint32), np.zeros(wc * hc * 3 * 18, np.uint32)
    cp = ctu_params(prm)
    rc = _EMUL_PB.ctu_pb_emul_search_picture(depth, ctypes.byref(cp), ctypes.byref(F), ptr(y), ptr(u), ptr(v), ptr(ry), ptr(ru), ptr(rv), ptr(scu), ptr(i4),
                                             ptr(trees), ptr(mot), ptr(co), ptr(mo), ptr(mi))
    assert rc == 0
    return inter_result_from_device_layout(W, H, ry, ru, rv, scu, i4, trees, mot, co, mo, mi)


def inter_result_from_device_layout(W, H, ry, ru, rv, scu, i4, trees, mot, co, mo, mi):
    wc, hc = (W + 63) // 64, (H + 63) // 64
    r = search_result_from_device_layout(W, H, ry, ru, rv, scu, co, mo)
    t = trees.reshape(hc * 16, wc * 16)
    r["trees"] = np.stack([t & 0xffff, t >> 16], axis=2).astype(np.uint32)
    r["scu"], r["inter4"], r["motion_dev"] = scu.reshape(hc * 16, wc * 16), i4.reshape(hc * 16, wc * 16), mot.reshape(hc * 16, wc * 16, 8)
    m = mi.reshape(hc * wc, 3, 18)
    r["models_inter_states"] = np.concatenate([(m & 0xffff).astype(np.uint16), (m >> 16).astype(np.uint16)], axis=2)      # [ctus, 3, 36]
    return r


def compare_device_inter_picture(W, H, d, r):
    """The device-layout result of a P / B picture's CTU search (the emulation or the GPU) against the encoder's records d -> differences.
    Vectors and reference indices of lists a unit does not use are not compared (see csrc/ctu_pb.h: one motion table for every depth)."""
    wc, hc = (W + 63) // 64, (H + 63) // 64
    h4, w4 = H // 4, W // 4
    msgs = []
    got, want = r["cu"][:h4, :w4], d["cu"][:h4, :w4]
    if not np.array_equal(got[:, :, :6], want[:, :, :6]):
        j = np.argwhere((got[:, :, :6] != want[:, :, :6]).any(axis=2))[0]
        msgs.append(f"cu differs first at 4x4 {j.tolist()}: got {got[j[0], j[1]].tolist()} want {want[j[0], j[1]].tolist()}")
        return msgs
    intra, inter = want[:, :, 0] == 1, want[:, :, 0] == 2
    if not np.array_equal(got[:, :, 6:8][intra], want[:, :, 6:8][intra]):
        msgs.append("intra modes differ")
    if not np.array_equal(got[:, :, 8:11], want[:, :, 8:11]):
        j = np.argwhere((got[:, :, 8:11] != want[:, :, 8:11]).any(axis=2))[0]
        msgs.append(f"edge flags / qp differ first at {j.tolist()}: got {got[j[0], j[1]].tolist()} want {want[j[0], j[1]].tolist()}")
    if not np.array_equal(r["trees"][:h4, :w4], d["trees"][:h4, :w4]):
        msgs.append("trees differ")
    s, f4, wm = r["scu"][:h4, :w4], r["inter4"][:h4, :w4], d["motion"][:h4, :w4]
    if not np.array_equal(s["mv_dir"][inter], wm[:, :, 6][inter]):
        msgs.append("mv_dir differs")
    for l in (0, 1):
        use = inter & ((wm[:, :, 6] & (1 << l)) != 0)
        for k in (0, 1):
            if not np.array_equal(s["mv"][:, :, l, k][use], wm[:, :, 2 * l + k][use]):
                j = np.argwhere(use & (s["mv"][:, :, l, k] != wm[:, :, 2 * l + k]))[0]
                msgs.append(f"mv[{l}][{k}] differs first at 4x4 {j.tolist()}: got {s['mv'][j[0], j[1]].tolist()} want {wm[j[0], j[1]].tolist()}")
        if not np.array_equal(f4["mv_ref%d" % l][use], wm[:, :, 4 + l][use]):
            msgs.append(f"mv_ref{l} differs")
    flags = wm[:, :, 7]
    for nme, val in (("skipped", flags & 1), ("merged", (flags >> 1) & 1), ("merge_idx", (flags >> 2) & 7), ("mv_cand0", (flags >> 8) & 7), ("mv_cand1", (flags >> 11) & 7)):
        if not np.array_equal(f4[nme][inter], val[inter]):
            j = np.argwhere(inter & (f4[nme] != val))[0]
            msgs.append(f"{nme} differs first at 4x4 {j.tolist()}")
    for c, nme in enumerate(("rec_y", "rec_u", "rec_v")):
        if not np.array_equal(r[nme], d["rec"][c]):
            j = np.argwhere(r[nme] != d["rec"][c])[0]
            msgs.append(f"{nme} differs first at {j.tolist()}")
    for k in range(wc * hc):
        hh, ww = min(64, H - (k // wc) * 64), min(64, W - (k % wc) * 64)
        a, b = r["coeff"][k], d["coeff"][k]
        if not (np.array_equal(a[:4096].reshape(64, 64)[:hh, :ww], b[:4096].reshape(64, 64)[:hh, :ww]) and
                np.array_equal(a[4096:].reshape(2, 32, 32)[:, :hh // 2, :ww // 2], b[4096:].reshape(2, 32, 32)[:, :hh // 2, :ww // 2])):
            msgs.append(f"levels differ in CTU {k}")
            break
    for j, what in enumerate(("start", "after search", "after coder")):
        a, b = r["models"][:, j, :1028], d["models"][:, j, :1028]
        if not np.array_equal(a, b):
            k = int(np.argwhere((a != b).any(axis=1))[0][0])
            av, bv = a[k].view(np.uint16), b[k].view(np.uint16)
            msgs.append(f"models {what}: first CTU {k} entries {(np.argwhere(av != bv).ravel() % 257).tolist()[:8]}")
        a, b = r["models_inter_states"][:, j], np.ascontiguousarray(d["models_inter"][:, j, :72]).view(np.uint16)
        if not np.array_equal(a, b):
            k = int(np.argwhere((a != b).any(axis=1))[0][0])
            msgs.append(f"inter models {what}: first CTU {k} entries {(np.argwhere(a[k] != b[k]).ravel() % 18).tolist()[:8]}")
    return msgs


def run_inter_oracle(orc, W, H, depth, pics, P, ctx_trace=False):
    """The oracle on every picture in coding order, references = the ENCODER's output pictures and side information (so one picture's
    mismatch does not spread) -> yields (frame, record dict, oracle result, trace rows, n_trace).  ctx_trace: d["ctx_trace"] = (ints, doubles)
    of every search_pu_inter call (orcN_search_ctx_trace), d["frame"] = the picture's reference lists, planes and tables."""
    wc, hc = (W + 63) // 64, (H + 63) // 64
    by_poc = {}
    for fr in sorted(P):
        d = P[fr]
        refs = d["refs"]
        n_refs, pocs = int(refs[0]), [int(a) for a in refs[1:17]]
        lsz = [int(refs[17]), int(refs[18])]
        lists = [[int(a) for a in refs[19:35]], [int(a) for a in refs[35:51]]]
        poc, slice_type = int(refs[51]), int(d["meta"][6])
        lam = d["lam"]
        cfg = [int(a) for a in d.get("cfg", (1, 6, 2, 1, 4, 1))]
        rd = cfg[6] if len(cfg) > 6 else 0          # cfg.rdo of the run (0: --preset medium, 1: --preset slow)
        owf = cfg[7] if len(cfg) > 7 else 0         # cfg.owf != 0: vectors restricted to what is final in the reference picture (oracle only so far)
        prm = SearchParams(W, H, int(d["meta"][3]), int(d["meta"][3]), 1, 4, 1, 1, 2, rd, float(lam[0]), float(lam[1]), float(lam[2]), float(lam[3]), float(lam[4]))
        F = InterFrame()
        F.slice_type, F.poc, F.n_refs = slice_type, poc, n_refs
        for i in range(16):
            F.ref_pocs[i] = pocs[i]
            F.l[0][i], F.l[1][i] = lists[0][i], lists[1][i]
        F.l_size[0], F.l_size[1] = lsz
        F.tmvp, F.max_merge, F.merge_level, F.bipred, F.fme_level, F.early_skip = cfg[:6]
        F.owf, F.owf_margin = int(owf != 0), 10       # (SAO_DELAY_PX: the goldens' runs have SAO on)
        F.depth_inter_min, F.depth_inter_max = 0, 3
        F.ref_cu_stride, F.frame_qp = wc * 16, int(d["meta"][7])
        keep = []
        for i in range(n_refs):
            rp = by_poc[pocs[i]]
            planes = [np.ascontiguousarray(p) for p in rp["final"]]
            keep += planes + [rp["ref_cu"]]
            F.ref_y[i], F.ref_u[i], F.ref_v[i] = (p.ctypes.data for p in planes)
            F.ref_cu[i] = rp["ref_cu"].ctypes.data
        buf = np.zeros((wc * hc * 400, 22), np.float64)
        orc.fn(depth, "search_trace", None)(ptr(buf), len(buf))
        if ctx_trace:
            ci, cd = np.zeros((wc * hc * 100, 64 + 290 * 8 + 41 + 7), np.int32), np.zeros((wc * hc * 100, 8), np.float64)
            orc.fn(depth, "search_ctx_trace", None)(ptr(ci), ptr(cd), len(ci))
        y, u, v = pics[fr]
        r = oracle_search_inter_picture(orc, depth, prm, F, y, u, v, keep)
        ntr = orc.fn(depth, "search_trace_count")()
        orc.fn(depth, "search_trace", None)(None, 0)
        if ctx_trace:
            nct = orc.fn(depth, "search_ctx_trace_count")()
            orc.fn(depth, "search_ctx_trace", None)(None, None, 0)
            d["ctx_trace"] = (ci[:nct], cd[:nct])
            d["frame"] = dict(poc=poc, slice_type=slice_type, n_refs=n_refs, pocs=pocs, l_size=lsz, lists=lists,
                              ref_planes=[by_poc[pocs[i]]["final"] for i in range(n_refs)], ref_cu=[by_poc[pocs[i]]["ref_cu"] for i in range(n_refs)])
        # this picture as a reference of later ones
        own_pocs = ([pocs[lists[0][i]] for i in range(lsz[0])], [pocs[lists[1][i]] for i in range(lsz[1])])
        d["ref_cu"] = ref_cu_table(d["cu"], d["motion"], own_pocs)
        by_poc[poc] = d
        d["info"] = (poc, slice_type, pocs[:n_refs], lists[0][:lsz[0]], lists[1][:lsz[1]])
        yield fr, d, r, buf, ntr



def compare_inter_picture(W, H, d, r, buf, ntr):
    """One picture of the oracle's inter search (r, trace rows buf[:ntr]) against the encoder's records d -> list of differences."""
    wc, hc = (W + 63) // 64, (H + 63) // 64
# ---- compare
    msgs = []
    tr = d["cuinter"]
    for i in range(min(ntr, len(tr))):
        a, c = buf[i], tr[i]
        want = np.concatenate([c[0][1:20].astype(np.float64), c[1]])
        got = a[1:22]
        ok = np.array_equal(got[:5], want[:5]) and got[19] == want[19]
        if ok and want[4] == 2 and want[19] < 1e300:        # an inter decision: everything must agree
            sel = [4, 5, 6, 7, 8, 13, 14, 15, 16, 19, 20] + [9 + 2 * l + j for l in (0, 1) if int(want[8]) & (1 << l) for j in (0, 1)]
            ok = all(got[j] == want[j] for j in sel)
        if not ok:
            msgs.append(f"cuinter call {i}: got {got.tolist()} want {want.tolist()}")
            break
    if ntr != len(tr):
        msgs.append(f"cuinter calls: {ntr} vs {len(tr)}")
    h4, w4 = H // 4, W // 4
    if not np.array_equal(r["cu"][:h4, :w4, :6], d["cu"][:h4, :w4, :6]):
        j = np.argwhere((r["cu"][:h4, :w4, :6] != d["cu"][:h4, :w4, :6]).any(axis=2))[0]
        msgs.append(f"cu differs first at 4x4 {j.tolist()}: got {r['cu'][j[0], j[1]].tolist()} want {d['cu'][j[0], j[1]].tolist()}")
    intra = d["cu"][:h4, :w4, 0] == 1
    if not np.array_equal(r["cu"][:h4, :w4, 6:8][intra], d["cu"][:h4, :w4, 6:8][intra]):
        msgs.append("intra modes differ")
    if not np.array_equal(r["cu"][:h4, :w4, 8:11], d["cu"][:h4, :w4, 8:11]):
        msgs.append("edge flags / qp differ")
    if not np.array_equal(r["trees"][:h4, :w4], d["trees"][:h4, :w4]):
        msgs.append("trees differ")
    if not np.array_equal(r["motion"][:h4, :w4], d["motion"][:h4, :w4] & np.array([-1, -1, -1, -1, -1, -1, -1, 0x1f], np.int32)):
        j = np.argwhere((r["motion"][:h4, :w4] != (d["motion"][:h4, :w4] & np.array([-1, -1, -1, -1, -1, -1, -1, 0x1f], np.int32))).any(axis=2))[0]
        msgs.append(f"motion differs first at 4x4 {j.tolist()}: got {r['motion'][j[0], j[1]].tolist()} want {d['motion'][j[0], j[1]].tolist()}")
    for c, nme in enumerate(("rec_y", "rec_u", "rec_v")):
        if not np.array_equal(r[nme], d["rec"][c]):
            msgs.append(nme + " differs")
    for k in range(wc * hc):             # the levels inside the picture (a partial CTU's lcu_coeff_t holds leftovers beyond it)
        hh, ww = min(64, H - (k // wc) * 64), min(64, W - (k % wc) * 64)
        for a, b in ((r["coeff"][k], d["coeff"][k]),):
            if not (np.array_equal(a[:4096].reshape(64, 64)[:hh, :ww], b[:4096].reshape(64, 64)[:hh, :ww]) and
                    np.array_equal(a[4096:].reshape(2, 32, 32)[:, :hh // 2, :ww // 2], b[4096:].reshape(2, 32, 32)[:, :hh // 2, :ww // 2])):
                msgs.append(f"levels differ in CTU {k}")
                break
    for j, what in enumerate(("start", "after search", "after coder")):
        a, b = r["models"][:, j, :1285], d["models"][:, j, :1285]
        if not np.array_equal(a, b):
            msgs.append(f"models {what}: first CTU {int(np.argwhere((a != b).any(axis=1))[0][0])}")
        a, b = r["models_inter"][:, j], d["models_inter"][:, j]
        if not np.array_equal(a, b):
            k = int(np.argwhere((a != b).any(axis=1))[0][0])
            msgs.append(f"inter models {what}: first CTU {k} entries {np.argwhere(a[k] != b[k]).ravel().tolist()[:8]}")
    return msgs


def golden_sources(g):
    """The source pictures of a ref_inter_* / ref_intercrc_* golden's run, one (y, u, v) per coded picture in CODING order (checked
    against the golden's CRCs of what the encoder was fed)."""
    import zlib
    W, H, depth, qp0, frames = (int(a) for a in g["dims"])
    gen = globals()[CLIP_GENERATORS[int(g["clip"])]] if ("clip" in g.files and int(g["clip"])) else moving_picture
    shown = [gen(W, H, t, depth) for t in range(frames)]
    for t in range(frames):
        assert zlib.crc32(b"".join(p.tobytes() for p in shown[t])) == int(g["src_crc"][t]), "the sequence generator drifted from the golden's source"
    # every per-picture array of the golden is in CODING order; the source of coded picture f is display picture display[f] (random access)
    display = [int(a) for a in g["display"]] if "display" in g.files else list(range(frames))
    return [shown[display[f]] for f in range(frames)]


def inter_pictures_from_golden(g):
    """A ref_inter_* golden (tools/refcheck/make_ctu_goldens.py::inter) as the per-picture records run_inter_oracle / compare_inter_picture
    take, plus the source pictures -> (W, H, depth, [source (y, u, v) per frame], {frame: record dict})"""
    W, H, depth, qp0, frames = (int(a) for a in g["dims"])
    wc, hc = (W + 63) // 64, (H + 63) // 64
    pics = golden_sources(g)
    P = {}
    for k in range(len(g["meta"])):
        fr, x, y = (int(a) for a in g["meta"][k][:3])
        d = P.setdefault(fr, dict(cu=np.zeros((hc * 16, wc * 16, 12), np.uint8), trees=np.zeros((hc * 16, wc * 16, 2), np.uint32),
                                  motion=np.zeros((hc * 16, wc * 16, 8), np.int32), coeff=np.zeros((wc * hc, 6144), np.int16),
                                  models=np.zeros((wc * hc, 3, MODELS_BYTES), np.uint8), models_inter=np.zeros((wc * hc, 3, MODELS_INTER_BYTES), np.uint8),
                                  cuinter=[]))
        d["meta"], d["lam"], d["refs"] = g["meta"][k], g["lam"][k], g["refs"][k]
        kk = (y // 64) * wc + x // 64
        d["cu"][y // 4:y // 4 + 16, x // 4:x // 4 + 16] = g["cu"][k].reshape(16, 16, 12)
        d["trees"][y // 4:y // 4 + 16, x // 4:x // 4 + 16] = g["trees"][k].reshape(16, 16, 2)
        d["motion"][y // 4:y // 4 + 16, x // 4:x // 4 + 16] = g["motion"][k].reshape(16, 16, 8)
        d["coeff"][kk] = g["coeff"][k]
        d["models"][kk], d["models_inter"][kk] = g["models"][k], g["models_inter"][k]
    for fr, d in P.items():
        if "cfg" in g.files:
            d["cfg"] = g["cfg"]          # the tools of the run (tmvp, max_merge, merge_level, bipred, fme_level, early_skip); absent: --preset medium's
        d["rec"] = [g["rec_y"][fr], g["rec_u"][fr], g["rec_v"][fr]]
        d["final"] = (g["final_y"][fr], g["final_u"][fr], g["final_v"][fr])
    for a, b in zip(g["cuinter_i"], g["cuinter_d"]):
        P[int(a[0])]["cuinter"].append((a, b))
    return W, H, depth, pics, P


def is_random_access(g):
    return "display" in g.files and not np.array_equal(g["display"], np.arange(len(g["display"])))


def poc_lsb_bits(g):
    """encoder_control->poc_lsb_bits (src/encoder.c:242) of golden g's stream: from the GOP length of a random-access golden (16 where it is not recorded), else 4."""
    if not is_random_access(g):
        return 4
    gop_len = int(g["gop_len"]) if "gop_len" in g.files and int(g["gop_len"]) else 16
    return max(4, int(np.ceil(np.log2(2 * gop_len + 1))))


def write_inter_nals(L, g, poc, slice_type, ref_pocs, bipred, tmvp, qp_delta, rows, sizes, sums, out, n, irap_poc=0):
    """The NAL units of a P / B picture of golden g's stream from the library's host functions: uvghip_write_picture_nals_pb for a low-delay
    stream, uvghip_write_picture_nals_ra for a random-access one (g has `display`: pictures coded out of order, references in the future,
    poc_lsb_bits from the GOP length, src/encoder.c:242)."""
    import ctypes
    hc = len(sizes)
    neg = np.ascontiguousarray(sorted(poc - p for p in ref_pocs if p < poc), np.int32)
    pos = np.ascontiguousarray(sorted(p - poc for p in ref_pocs if p > poc), np.int32)
    if is_random_access(g) and (slice_type == 2 or poc < irap_poc):
        # a later intra period of an open-GOP stream: its I picture is a CRA picture, the pictures before it that are coded after it RASL pictures
        # (src/encoderstate.c:1957-1972; irap_poc: the POC of the most recent I picture in coding order)
        return L.uvghip_write_picture_nals_gop(9 if slice_type == 2 else 3, poc, poc_lsb_bits(g), slice_type, len(neg), ptr(neg) if len(neg) else None, len(pos),
                                               ptr(pos) if len(pos) else None, tmvp, qp_delta, 1, ptr(rows), rows.shape[1], ptr(sizes), hc, ptr(sums), ptr(out), len(out), ctypes.byref(n))
    if is_random_access(g):
        return L.uvghip_write_picture_nals_ra(poc, poc_lsb_bits(g), slice_type, len(neg), ptr(neg), len(pos), ptr(pos) if len(pos) else None, tmvp, qp_delta, 1, ptr(rows), rows.shape[1],
                                              ptr(sizes), hc, ptr(sums), ptr(out), len(out), ctypes.byref(n))
    assert len(pos) == 0
    return L.uvghip_write_picture_nals_pb(poc, 4, slice_type, len(neg), ptr(neg), bipred, tmvp, qp_delta, 1, ptr(rows), rows.shape[1], ptr(sizes), hc,
                                          ptr(sums), ptr(out), len(out), ctypes.byref(n))


def alf_sum_layout(ee, yv, pa, n_coeff):
    """Per-class ALF covariances (ee [C][13][13][4][4], y [C][13][4], pix_acc [C]) -> uvghip_alf_cov_reduce's sum layout [C][1509] int64: the ee
    triangle k <= l at (k * 13 - k (k - 1) / 2 + l - k) * 16 + b0 * 4 + b1, y[k][b] at 1456 + 4 k + b, pix_acc at 1508."""
    C = ee.shape[0]
    out = np.zeros((C, 1509), np.int64)
    for k in range(n_coeff):
        for l in range(k, n_coeff):
            at = (k * 13 - k * (k - 1) // 2 + l - k) * 16
            out[:, at:at + 16] = ee[:, k, l].reshape(C, 16)
    out[:, 1456:1456 + 4 * n_coeff] = yv[:, :n_coeff].reshape(C, 4 * n_coeff)
    out[:, 1508] = pa
    return out
