"""Thin torch-tensor front end over the batched C ABI (include/uvg266_hip.h).

Every function takes CUDA(HIP) tensors that are already resident in HBM,
enqueues the kernel on torch's current stream and returns device tensors.
Planes are 2-D tensors (rows x stride) of dtype uint8 (8-bit) or uint16/int16
(10-bit); `width` is the visible width when the stride is larger.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib


def _depth(t):
    if t.dtype == torch.uint8:
        return 8
    if t.dtype in (torch.uint16, torch.int16):
        return 10
    raise TypeError(f"plane dtype {t.dtype} is not a uvg_pixel type")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t):
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return t.data_ptr()


def make_blocks(cur_xy, ref_xy, device="cuda"):
    """(n,2) cur positions and (n,2) ref positions -> device array of uvghip_blk_t."""
    a = np.concatenate([np.asarray(cur_xy, np.int32).reshape(-1, 2),
                        np.asarray(ref_xy, np.int32).reshape(-1, 2)], axis=1)
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _batch(fn_name, cur, ref, bw, bh, blks, ref_w=None, ref_h=None):
    L = _lib.init(cur.device.index or 0)
    n = blks.shape[0]
    out = torch.empty(n, dtype=torch.int32, device=cur.device)
    ref_w = ref.shape[1] if ref_w is None else ref_w
    ref_h = ref.shape[0] if ref_h is None else ref_h
    rc = getattr(L, fn_name)(_depth(cur), _dev(cur), cur.stride(0), _dev(ref), ref.stride(0),
                             ref_w, ref_h, bw, bh, _dev(blks), n, _dev(out), _stream())
    _lib.check(rc, fn_name)
    return out


def sad_batch(cur, ref, bw, bh, blks, ref_w=None, ref_h=None):
    """uvg_image_calc_sad for n same-size blocks (edge-replicated reference)."""
    return _batch("uvghip_sad_batch", cur, ref, bw, bh, blks, ref_w, ref_h)


def satd_batch(cur, ref, bw, bh, blks, ref_w=None, ref_h=None):
    """uvg_image_calc_satd / uvg_satd_any_size for n same-size blocks."""
    return _batch("uvghip_satd_batch", cur, ref, bw, bh, blks, ref_w, ref_h)


def ssd_batch(a, b, bw, bh, blks):
    L = _lib.init(a.device.index or 0)
    n = blks.shape[0]
    out = torch.empty(n, dtype=torch.int32, device=a.device)
    rc = L.uvghip_ssd_batch(_depth(a), _dev(a), a.stride(0), _dev(b), b.stride(0), bw, bh,
                            _dev(blks), n, _dev(out), _stream())
    _lib.check(rc, "uvghip_ssd_batch")
    return out


def sad_surface(cur, ref, w, h, bw, bh, rng):
    L = _lib.init(cur.device.index or 0)
    nblk = (w // bw) * (h // bh)
    side = 2 * rng + 1
    out = torch.empty((nblk, side, side), dtype=torch.int32, device=cur.device)
    rc = L.uvghip_sad_surface(_depth(cur), _dev(cur), cur.stride(0), _dev(ref), ref.stride(0),
                              w, h, bw, bh, rng, _dev(out), _stream())
    _lib.check(rc, "uvghip_sad_surface")
    return out


def residual_plane(a, b, w=None, h=None):
    L = _lib.init(a.device.index or 0)
    h = a.shape[0] if h is None else h
    w = a.shape[1] if w is None else w
    res = torch.empty((h, w), dtype=torch.int16, device=a.device)
    rc = L.uvghip_residual_plane(_depth(a), _dev(a), a.stride(0), _dev(b), b.stride(0),
                                 _dev(res), res.stride(0), w, h, _stream())
    _lib.check(rc, "uvghip_residual_plane")
    return res


# ---- transforms ------------------------------------------------------------
TR_DCT2, TR_DCT8, TR_DST7 = 0, 1, 2


def mts_select(width, height, color=0, cu_type=1, isp_mode=0, lfnst_idx=0, cr_lfnst_idx=0, tr_idx=0, mts_type=0):
    """-> (type_hor, type_ver, skip_width, skip_height) per uvg_get_tr_type + mts skip rules."""
    import ctypes
    L = _lib.load_library()
    o = [ctypes.c_int() for _ in range(4)]
    L.uvghip_mts_select(width, height, color, cu_type, isp_mode, lfnst_idx, cr_lfnst_idx, tr_idx, mts_type,
                        *[ctypes.byref(v) for v in o])
    return tuple(v.value for v in o)


def transform_batch(blocks, bitdepth, inverse=False, type_hor=TR_DCT2, type_ver=TR_DCT2, skip_w=0, skip_h=0):
    """blocks: (n, h, w) int16 device tensor -> (n, h, w) int16 coefficients / residuals."""
    L = _lib.init(blocks.device.index or 0)
    assert blocks.dtype == torch.int16
    n, h, w = blocks.shape
    out = torch.empty_like(blocks)
    rc = L.uvghip_transform_batch(bitdepth, int(inverse), type_hor, type_ver, w, h, skip_w, skip_h,
                                  _dev(blocks), _dev(out), n, _stream())
    _lib.check(rc, "uvghip_transform_batch")
    return out


# ---- quantisation / TU round trip ---------------------------------------------
def quant_batch(coef, bitdepth, qp_scaled, transform_skip=False, slice_is_intra=True):
    L = _lib.init(coef.device.index or 0)
    n, h, w = coef.shape
    out = torch.empty_like(coef)
    _lib.check(L.uvghip_quant_batch(bitdepth, _dev(coef), _dev(out), w, h, n, qp_scaled, int(transform_skip),
                                    int(slice_is_intra), _stream()), "uvghip_quant_batch")
    return out


def dequant_batch(q_coef, bitdepth, qp_scaled, transform_skip=False):
    L = _lib.init(q_coef.device.index or 0)
    n, h, w = q_coef.shape
    out = torch.empty_like(q_coef)
    _lib.check(L.uvghip_dequant_batch(bitdepth, _dev(q_coef), _dev(out), w, h, n, qp_scaled, int(transform_skip),
                                      _stream()), "uvghip_dequant_batch")
    return out


def coeff_abs_sum_batch(coeffs):
    L = _lib.init(coeffs.device.index or 0)
    n = coeffs.shape[0]
    length = coeffs[0].numel()
    out = torch.empty(n, dtype=torch.int32, device=coeffs.device)
    _lib.check(L.uvghip_coeff_abs_sum_batch(_dev(coeffs), length, n, _dev(out), _stream()), "uvghip_coeff_abs_sum_batch")
    return out


def fast_coeff_cost_batch(coeffs, weights):
    L = _lib.init(coeffs.device.index or 0)
    n, h, w = coeffs.shape
    out = torch.empty(n, dtype=torch.int32, device=coeffs.device)
    _lib.check(L.uvghip_fast_coeff_cost_batch(_dev(coeffs), w, h, n, weights, _dev(out), _stream()),
               "uvghip_fast_coeff_cost_batch")
    return out


def make_tus(xy, device="cuda"):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(xy, np.int32).reshape(-1, 2))).to(device)


def tu_roundtrip_batch(orig, pred, rec, tus, w, h, qp_scaled, slice_is_intra=True,
                       type_hor=TR_DCT2, type_ver=TR_DCT2, skip_w=0, skip_h=0):
    """Fused residual->transform->quant->dequant->inverse->recon for n TUs.  Writes `rec` in place;
    returns (coeff (n,h,w) int16, has_coeffs (n,) uint8)."""
    L = _lib.init(orig.device.index or 0)
    n = tus.shape[0]
    coeff = torch.empty((n, h, w), dtype=torch.int16, device=orig.device)
    has = torch.empty(n, dtype=torch.uint8, device=orig.device)
    _lib.check(L.uvghip_tu_roundtrip_batch(_depth(orig), type_hor, type_ver, skip_w, skip_h, w, h, qp_scaled,
                                           int(slice_is_intra), _dev(orig), orig.stride(0), _dev(pred), pred.stride(0),
                                           _dev(rec), rec.stride(0), _dev(tus), n, _dev(coeff), _dev(has), _stream()),
               "uvghip_tu_roundtrip_batch")
    return coeff, has


def crc32c_batch(plane, blks_xy, size):
    """CRC-32C (IBC hash) of size x size blocks at (x, y) rows of `blks_xy` (device uvghip_tu_t array) -> (n,) int32 bit patterns."""
    L = _lib.init(plane.device.index or 0)
    n = blks_xy.shape[0]
    out = torch.empty(n, dtype=torch.int32, device=plane.device)
    _lib.check(L.uvghip_crc32c_batch(_depth(plane), _dev(plane), plane.stride(0), size, _dev(blks_xy), n, _dev(out), _stream()),
               "uvghip_crc32c_batch")
    return out


def pixel_var_batch(arr):
    """arr (n, len) samples -> (n,) float64 variances."""
    L = _lib.init(arr.device.index or 0)
    n, ln = arr.shape
    out = torch.empty(n, dtype=torch.float64, device=arr.device)
    _lib.check(L.uvghip_pixel_var_batch(_depth(arr), _dev(arr), ln, n, _dev(out), _stream()), "uvghip_pixel_var_batch")
    return out


# ---- intra ---------------------------------------------------------------------
def make_intra_blocks(xyaa, device="cuda"):
    """(n,4) rows of (x, y, avail_top, avail_left) -> device array of uvghip_intra_blk_t."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(xyaa, np.int32).reshape(-1, 4))).to(device)


def make_modes(modes, device="cuda"):
    return torch.from_numpy(np.asarray(modes, np.int8)).to(device)


def intra_pred_batch(rec, blks, w, h, modes, is_chroma=False):
    """-> (n, n_modes, h, w) predictions (dtype of `rec`)."""
    L = _lib.init(rec.device.index or 0)
    n, nm = blks.shape[0], modes.shape[0]
    out = torch.empty((n, nm, h, w), dtype=rec.dtype, device=rec.device)
    _lib.check(L.uvghip_intra_pred_batch(_depth(rec), _dev(rec), rec.stride(0), int(is_chroma), w, h, _dev(blks), n,
                                         _dev(modes), nm, _dev(out), _stream()), "uvghip_intra_pred_batch")
    return out


def mip_pred_batch(rec, blks, w, h, mode_transp):
    """Matrix-based intra prediction; mode_transp: (n,) uint8 = mip_mode | transpose << 7 -> (n, h, w) predictions."""
    L = _lib.init(rec.device.index or 0)
    n = blks.shape[0]
    out = torch.empty((n, h, w), dtype=rec.dtype, device=rec.device)
    _lib.check(L.uvghip_mip_pred_batch(_depth(rec), _dev(rec), rec.stride(0), w, h, _dev(blks), n, _dev(mode_transp), _dev(out),
                                       _stream()), "uvghip_mip_pred_batch")
    return out


def intra_search_batch(rec, orig, blks, size, modes):
    """-> (n, n_modes) int32 costs min(SATD, 2*SAD)."""
    L = _lib.init(rec.device.index or 0)
    n, nm = blks.shape[0], modes.shape[0]
    out = torch.empty((n, nm), dtype=torch.int32, device=rec.device)
    _lib.check(L.uvghip_intra_search_batch(_depth(rec), _dev(rec), rec.stride(0), _dev(orig), orig.stride(0), size,
                                           _dev(blks), n, _dev(modes), nm, _dev(out), _stream()),
               "uvghip_intra_search_batch")
    return out


def intra_search_best_batch(rec, orig, blks, size, modes, want_costs=False):
    """Fused rough search + arg-min -> (best_mode (n,) int8, best_cost (n,) int32[, costs (n, n_modes)])."""
    L = _lib.init(rec.device.index or 0)
    n, nm = blks.shape[0], modes.shape[0]
    best = torch.empty((n,), dtype=torch.int8, device=rec.device)
    cost = torch.empty((n,), dtype=torch.int32, device=rec.device)
    costs = torch.empty((n, nm), dtype=torch.int32, device=rec.device) if want_costs else None
    _lib.check(L.uvghip_intra_search_best_batch(_depth(rec), _dev(rec), rec.stride(0), _dev(orig), orig.stride(0), size,
                                                _dev(blks), n, _dev(modes), nm, _dev(best), _dev(cost),
                                                _dev(costs) if want_costs else None, _stream()),
               "uvghip_intra_search_best_batch")
    return (best, cost, costs) if want_costs else (best, cost)


def intra_pred_plane_batch(rec, blks, size, block_modes, pred_plane):
    """Predict every block with its own mode (block_modes: (n,) int8) into `pred_plane` (in place)."""
    L = _lib.init(rec.device.index or 0)
    _lib.check(L.uvghip_intra_pred_plane_batch(_depth(rec), _dev(rec), rec.stride(0), size, _dev(blks), blks.shape[0],
                                               _dev(block_modes), _dev(pred_plane), pred_plane.stride(0), _stream()),
               "uvghip_intra_pred_plane_batch")
    return pred_plane


def intra_select_best(costs, modes):
    """-> (best_mode (n,) int8, best_cost (n,) int32); ties keep the earlier candidate."""
    L = _lib.init(costs.device.index or 0)
    n, nm = costs.shape
    bm = torch.empty(n, dtype=torch.int8, device=costs.device)
    bc = torch.empty(n, dtype=torch.int32, device=costs.device)
    _lib.check(L.uvghip_intra_select_best(_dev(costs), n, _dev(modes), nm, _dev(bm), _dev(bc), _stream()),
               "uvghip_intra_select_best")
    return bm, bc


# ---- interpolation ---------------------------------------------------------------
def make_mc_blocks(xyff, device="cuda"):
    """(n,4) rows (x, y, fx, fy) -> device array of uvghip_mc_blk_t."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(xyff, np.int32).reshape(-1, 4))).to(device)


def mc_batch(ref, blks, w, h, pic_w=None, pic_h=None, is_chroma=False, hi=False):
    """Motion-compensated blocks -> (n, h, w) pixels, or int16 14-bit intermediates when hi."""
    L = _lib.init(ref.device.index or 0)
    n = blks.shape[0]
    pic_w = ref.shape[1] if pic_w is None else pic_w
    pic_h = ref.shape[0] if pic_h is None else pic_h
    out = torch.empty((n, h, w), dtype=torch.int16 if hi else ref.dtype, device=ref.device)
    _lib.check(L.uvghip_mc_batch(_depth(ref), _dev(ref), ref.stride(0), pic_w, pic_h, int(is_chroma), w, h, _dev(blks), n,
                                 int(hi), _dev(out), _stream()), "uvghip_mc_batch")
    return out


def frac_satd_batch(cur, ref, blks, w, h, cand_mv, pic_w=None, pic_h=None):
    """-> (n, n_cand) SATD costs of the fractional-ME candidates cand_mv ((k,2) int16, 1/16 units)."""
    L = _lib.init(ref.device.index or 0)
    n, k = blks.shape[0], cand_mv.shape[0]
    pic_w = ref.shape[1] if pic_w is None else pic_w
    pic_h = ref.shape[0] if pic_h is None else pic_h
    out = torch.empty((n, k), dtype=torch.int32, device=ref.device)
    _lib.check(L.uvghip_frac_satd_batch(_depth(ref), _dev(cur), cur.stride(0), _dev(ref), ref.stride(0), pic_w, pic_h, w, h,
                                        _dev(blks), n, _dev(cand_mv), k, _dev(out), _stream()), "uvghip_frac_satd_batch")
    return out


def bipred_average_batch(l0, l1, bitdepth):
    """l0/l1: flat tensors, pixel dtype or int16 (14-bit) each -> pixels."""
    L = _lib.init(l0.device.index or 0)
    pxdt = torch.uint8 if bitdepth == 8 else torch.uint16
    mode = (1 if l0.dtype == torch.int16 else 0) | (2 if l1.dtype == torch.int16 else 0)
    out = torch.empty(l0.numel(), dtype=pxdt, device=l0.device)
    _lib.check(L.uvghip_bipred_average_batch(bitdepth, _dev(l0), _dev(l1), mode, l0.numel(), _dev(out), _stream()),
               "uvghip_bipred_average_batch")
    return out


# ---- SAO ---------------------------------------------------------------------------
def make_rects(xywh, device="cuda"):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(xywh, np.int32).reshape(-1, 4))).to(device)


def make_sao_params(rows, device="cuda"):
    """(n,8) rows (type, eo_class, band_position, o0..o4) -> device array of uvghip_sao_param_t."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(rows, np.int32).reshape(-1, 8))).to(device)


def sao_stats_batch(orig, rec, rects, edge=None, band=None):
    """-> (edge (n,4,2,5) int32, band (n,2,32) int32); written into `edge` / `band` when given (contiguous)."""
    L = _lib.init(rec.device.index or 0)
    n = rects.shape[0]
    edge = torch.empty((n, 4, 2, 5), dtype=torch.int32, device=rec.device) if edge is None else edge
    band = torch.empty((n, 2, 32), dtype=torch.int32, device=rec.device) if band is None else band
    _lib.check(L.uvghip_sao_stats_batch(_depth(rec), _dev(orig), orig.stride(0), _dev(rec), rec.stride(0), _dev(rects), n,
                                        _dev(edge), _dev(band), _stream()), "uvghip_sao_stats_batch")
    return edge, band


def sao_edge_offsets_batch(edge, rate_cost=None, params=None, ddist=None):
    """edge (n,4,2,5) statistics -> (n,8) int32 uvghip_sao_param_t rows (type 2, class, 0, offsets[5])."""
    L = _lib.init(edge.device.index or 0)
    n = edge.shape[0]
    if params is None:
        params = torch.empty((n, 8), dtype=torch.int32, device=edge.device)
    _lib.check(L.uvghip_sao_edge_offsets_batch(_dev(edge), _dev(rate_cost) if rate_cost is not None else None, n, _dev(params),
                                               _dev(ddist) if ddist is not None else None, _stream()),
               "uvghip_sao_edge_offsets_batch")
    return params


def sao_decide_buffers(n_pictures, pic_w, pic_h, device):
    """Workspace and outputs of sao_decide_pictures, for callers that decide again and again."""
    L = _lib.init(torch.device(device).index or 0)
    n = n_pictures * ((pic_w + 63) // 64) * ((pic_h + 63) // 64)
    return dict(ws=torch.empty(L.uvghip_sao_decide_workspace_bytes(n_pictures, pic_w, pic_h), dtype=torch.uint8, device=device),
                info=torch.zeros((n, 2, 17), dtype=torch.int32, device=device), models=torch.zeros((n, 6), dtype=torch.int16, device=device),
                params=[torch.empty((n, 8), dtype=torch.int32, device=device) for _ in range(3)])


def sao_decide_pictures(n_pictures, pic_w, pic_h, depth, qp, lam, stats, sao_type=3, out=None):
    """uvghip_sao_decide_pictures.  stats: ((edge_y, band_y), (edge_u, band_u), (edge_v, band_v)) from sao_stats_batch over the CTU
    grids of the pictures ([picture][ctu]).  -> info (n*ctus, 2, 17) int32, models (n*ctus, 6) uint16 (as int16 storage),
    (params_y, params_u, params_v) each (n*ctus, 8) int32."""
    dev = stats[0][0].device
    L = _lib.init(dev.index or 0)
    b = sao_decide_buffers(n_pictures, pic_w, pic_h, dev) if out is None else out
    (ey, by), (eu, bu), (ev, bv) = stats
    _lib.check(L.uvghip_sao_decide_pictures(depth, n_pictures, pic_w, pic_h, qp, float(lam), sao_type, _dev(ey), _dev(by), _dev(eu), _dev(bu),
                                            _dev(ev), _dev(bv), _dev(b["ws"]), _dev(b["info"]), _dev(b["models"]), _dev(b["params"][0]),
                                            _dev(b["params"][1]), _dev(b["params"][2]), _stream()), "uvghip_sao_decide_pictures")
    return b["info"], b["models"], b["params"]


def sao_apply_batch(rec, out, rects, params, pic_w=None, pic_h=None):
    L = _lib.init(rec.device.index or 0)
    pic_w = rec.shape[1] if pic_w is None else pic_w
    pic_h = rec.shape[0] if pic_h is None else pic_h
    _lib.check(L.uvghip_sao_apply_batch(_depth(rec), _dev(rec), rec.stride(0), _dev(out), out.stride(0), pic_w, pic_h,
                                        _dev(rects), _dev(params), rects.shape[0], _stream()), "uvghip_sao_apply_batch")
    return out


# ---- deblocking ---------------------------------------------------------------------
from .layout import SCU_DTYPE  # noqa: E402
assert SCU_DTYPE.itemsize == 32


def make_scu_table(table, device="cuda"):
    """numpy structured array (rows, cols) of SCU_DTYPE -> device byte tensor (rows, cols*32)."""
    t = np.ascontiguousarray(table)
    assert t.dtype == SCU_DTYPE and t.ndim == 2
    return torch.from_numpy(t.view(np.uint8).reshape(t.shape[0], t.shape[1] * 32)).to(device)


def deblock_frame(y, u, v, scu, width, height, beta_offset_div2=0, tc_offset_div2=0, slice_is_b=False, frame_qp=-1,
                  chroma_qp_map=None, sao_snapshot=False):
    """In-place deblocking of a picture.  scu: device table from make_scu_table (row stride = cols).
    sao_snapshot: uvghip_deblock_frame_sao_snapshot instead -- every CTU deblocked by its own edges only, what the reference's
    SAO decision reads."""
    import ctypes
    L = _lib.init(y.device.index or 0)
    qm = None
    if chroma_qp_map is not None:
        qm_arr = np.ascontiguousarray(np.asarray(chroma_qp_map, np.int8)[:64])
        qm = qm_arr.ctypes.data_as(ctypes.c_void_p)
    fn = L.uvghip_deblock_frame_sao_snapshot if sao_snapshot else L.uvghip_deblock_frame
    _lib.check(fn(_depth(y), _dev(y), y.stride(0), None if u is None else _dev(u),
                  None if v is None else _dev(v), 0 if u is None else u.stride(0), width, height,
                  _dev(scu), scu.shape[1] // 32, beta_offset_div2, tc_offset_div2, int(slice_is_b),
                  frame_qp, qm, _stream()), "uvghip_deblock_frame")
    return y


# ---- LFNST -------------------------------------------------------------------------
def make_lfnst_tus(rows, device="cuda"):
    """(n,4) rows of (intra_mode, lfnst_idx, log2_cu_width, log2_cu_height) -> device array of uvghip_lfnst_tu_t."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(rows, np.int8).reshape(-1, 4))).to(device)


def lfnst_batch(coeffs, tus, inverse=False):
    """In place on coeffs (n, h, w) int16."""
    L = _lib.init(coeffs.device.index or 0)
    n, h, w = coeffs.shape
    _lib.check(L.uvghip_lfnst_batch(int(inverse), _dev(coeffs), w, h, _dev(tus), n, _stream()), "uvghip_lfnst_batch")
    return coeffs


# ---- ALF ---------------------------------------------------------------------------
def alf_classify_frame(rec, width, height, shift=None):
    """-> (height/4, width/4) uint8: class_idx | transpose_idx << 5 per 4x4 luma block."""
    L = _lib.init(rec.device.index or 0)
    depth = _depth(rec)
    cls = torch.empty((height // 4, width // 4), dtype=torch.uint8, device=rec.device)
    _lib.check(L.uvghip_alf_classify_frame(depth, _dev(rec), rec.stride(0), width, height, depth + 4 if shift is None else shift,
                                           _dev(cls), cls.stride(0), _stream()), "uvghip_alf_classify_frame")
    return cls


def alf_filter_batch(src, dst, rects, set_idx, coef_sets, clip_sets, cls=None, is_chroma=False, pic_w=None, pic_h=None):
    L = _lib.init(src.device.index or 0)
    pic_w = src.shape[1] if pic_w is None else pic_w
    pic_h = src.shape[0] if pic_h is None else pic_h
    _lib.check(L.uvghip_alf_filter_batch(_depth(src), _dev(src), src.stride(0), _dev(dst), dst.stride(0), pic_w, pic_h,
                                         int(is_chroma), _dev(rects), _dev(set_idx), rects.shape[0], _dev(coef_sets),
                                         _dev(clip_sets), None if cls is None else _dev(cls),
                                         0 if cls is None else cls.stride(0), _stream()), "uvghip_alf_filter_batch")
    return dst


def cc_alf_stats_batch(org_c, rec_c, luma, rects):
    """uvghip_cc_alf_stats_batch: per chroma rectangle (a CTU) the CC-ALF covariance -> (ee [n][7][7] int64, y [n][7] int32, pix_acc [n] int64)."""
    L = _lib.init(org_c.device.index or 0)
    n = rects.shape[0]
    ee = torch.empty((n, 7, 7), dtype=torch.int64, device=org_c.device)
    y = torch.empty((n, 7), dtype=torch.int32, device=org_c.device)
    pix = torch.empty((n,), dtype=torch.int64, device=org_c.device)
    _lib.check(L.uvghip_cc_alf_stats_batch(_depth(org_c), _dev(org_c), org_c.stride(0), _dev(rec_c), rec_c.stride(0), _dev(luma), luma.stride(0), luma.shape[1], luma.shape[0],
                                           _dev(rects), n, _dev(ee), _dev(y), _dev(pix), _stream()), "uvghip_cc_alf_stats_batch")
    return ee, y, pix


def alf_reconstruct_picture(planes, slice_enabled, ctu_flags, filter_set_idx, luma_aps, chroma_aps, alf_full=False, cc_alf_enabled=(0, 0), cc_coeff=None,
                            classification_shift=None):
    """uvghip_alf_reconstruct_picture: the picture ALF leaves, from the picture it gets (three device planes: deblocked + SAO) and the
    encoder's decisions (numpy arrays, the layouts of include/uvg266_hip.h: ctu_flags u8 [7][n], filter_set_idx i16 [n], luma_aps i16
    [k][677], chroma_aps i16 [114], cc_coeff i16 [2][4][8]).  -> three new device planes."""
    L = _lib.init(planes[0].device.index or 0)
    depth = _depth(planes[0])
    h, w = planes[0].shape
    out = [torch.empty_like(p) for p in planes]
    flags = np.ascontiguousarray(ctu_flags, np.uint8)
    sets = np.ascontiguousarray(filter_set_idx, np.int16)
    laps = np.ascontiguousarray(luma_aps, np.int16).reshape(-1, 677)
    caps = None if chroma_aps is None else np.ascontiguousarray(chroma_aps, np.int16)
    ccc = None if cc_coeff is None else np.ascontiguousarray(cc_coeff, np.int16)
    ptr = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(ctypes.c_void_p)
    P = _lib.AlfPicture()
    P.in_y, P.in_u, P.in_v = (_dev(p) for p in planes)
    P.in_stride, P.in_stride_c = planes[0].stride(0), planes[1].stride(0)
    P.out_y, P.out_u, P.out_v = (_dev(o) for o in out)
    P.out_stride, P.out_stride_c = out[0].stride(0), out[1].stride(0)
    P.width, P.height = w, h
    for i in range(3):
        P.slice_enabled[i] = int(slice_enabled[i])
    P.n_luma_aps = laps.shape[0]
    P.ctu_flags, P.filter_set_idx, P.luma_aps, P.chroma_aps, P.cc_coeff = ptr(flags), ptr(sets), ptr(laps), ptr(caps), ptr(ccc)
    P.alf_full = int(bool(alf_full))
    for i in range(2):
        P.cc_alf_enabled[i] = int(cc_alf_enabled[i])
    P.classification_shift = int(classification_shift if classification_shift is not None else depth + 4)
    ws = torch.empty(L.uvghip_alf_reconstruct_workspace_bytes(w, h), dtype=torch.uint8, device=planes[0].device)
    _lib.check(L.uvghip_alf_reconstruct_picture(depth, ctypes.byref(P), _dev(ws), _stream()), "uvghip_alf_reconstruct_picture")
    return out


def alf_stats_batch(org, rec, rects, cls=None, is_chroma=False, pic_w=None, pic_h=None):
    """-> (ee (n,C,13,13,4,4) int64, y (n,C,13,4) int32, pix_acc (n,C) int64), C = 25 (luma) or 1 (chroma)."""
    L = _lib.init(rec.device.index or 0)
    n, C = rects.shape[0], (1 if is_chroma else 25)
    pic_w = rec.shape[1] if pic_w is None else pic_w
    pic_h = rec.shape[0] if pic_h is None else pic_h
    ee = torch.empty((n, C, 13, 13, 4, 4), dtype=torch.int64, device=rec.device)
    yv = torch.empty((n, C, 13, 4), dtype=torch.int32, device=rec.device)
    pa = torch.empty((n, C), dtype=torch.int64, device=rec.device)
    _lib.check(L.uvghip_alf_stats_batch(_depth(rec), _dev(org), org.stride(0), _dev(rec), rec.stride(0), pic_w, pic_h,
                                        int(is_chroma), _dev(rects), n, None if cls is None else _dev(cls),
                                        0 if cls is None else cls.stride(0), _dev(ee), _dev(yv), _dev(pa), _stream()),
               "uvghip_alf_stats_batch")
    return ee, yv, pa


# ---- RDOQ ---------------------------------------------------------------------------
def rdoq_batch(coef, bitdepth, color, block_type, cbf_u, lfnst_idx, mts_idx, qp_scaled, lam, ctx, workspace=None, signhide=False):
    """coef (n, h, w) int16 transformed blocks -> (levels (n, h, w) int16, abs_sum (n,) int32, has_coeffs (n,) uint8).
    ctx: 244 bytes (numpy uint8 or bytes) = uvghip_rdoq_ctx_t, a host-side snapshot."""
    import ctypes
    L = _lib.init(coef.device.index or 0)
    n, h, w = coef.shape
    out = torch.empty_like(coef)
    abs_sum = torch.empty(n, dtype=torch.int32, device=coef.device)
    has = torch.empty(n, dtype=torch.uint8, device=coef.device)
    need = (L.uvghip_rdoq_signhide_workspace_bytes if signhide else L.uvghip_rdoq_workspace_bytes)(w, h, n)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need + 7) // 8, dtype=torch.float64, device=coef.device)
    cbuf = (ctypes.c_uint8 * 244).from_buffer_copy(bytes(np.asarray(ctx, np.uint8).tobytes()))
    _lib.check((L.uvghip_rdoq_signhide_batch if signhide else L.uvghip_rdoq_batch)(bitdepth, _dev(coef), _dev(out), w, h, n, color, block_type, cbf_u, lfnst_idx, mts_idx, qp_scaled,
                                   float(lam), ctypes.cast(cbuf, ctypes.c_void_p), _dev(workspace), workspace.numel() * 8,
                                   _dev(abs_sum), _dev(has), _stream()), "uvghip_rdoq_batch")
    return out, abs_sum, has


def quantize_residual_batch(orig, pred, rec, tus, width, height, bitdepth, color=0, type_hor=0, type_ver=0, skip_w=0, skip_h=0,
                            qp_scaled=22, slice_is_intra=True, cu_type=1, use_trskip=False, rdoq=False, rdoq_skip=False, cbf_u=0,
                            mts_idx=0, lfnst_idx=0, lam=0.0, ctx=None, lfnst_tus=None, signhide=False):
    """uvg_quantize_residual on any branch (staged launches) -> (coeff (n, h, w) int16, has_coeffs (n,) uint8); rec in place."""
    import ctypes
    L = _lib.init(orig.device.index or 0)
    n = tus.shape[0]
    p = _lib.QrParams()
    p.width, p.height, p.color = width, height, color
    p.type_hor, p.type_ver, p.skip_width, p.skip_height = type_hor, type_ver, skip_w, skip_h
    p.qp_scaled, p.slice_is_intra, p.cu_type, p.use_trskip = qp_scaled, int(slice_is_intra), cu_type, int(use_trskip)
    p.rdoq_enable, p.rdoq_skip, p.dep_quant, p.cbf_u, p.mts_idx, p.lfnst_idx = int(rdoq), int(rdoq_skip), 0, cbf_u, mts_idx, lfnst_idx
    p.lambda_ = float(lam)
    p.signhide_enable = int(signhide)
    if ctx is not None:
        ctypes.memmove(p.ctx, np.asarray(ctx, np.uint8).tobytes(), 244)
    need = L.uvghip_quantize_residual_workspace_bytes(ctypes.byref(p), n)
    ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=orig.device)
    coeff = torch.empty((n, height, width), dtype=torch.int16, device=orig.device)
    has = torch.empty(n, dtype=torch.uint8, device=orig.device)
    _lib.check(L.uvghip_quantize_residual_batch(bitdepth, ctypes.byref(p), _dev(orig), orig.stride(0), _dev(pred), pred.stride(0), _dev(rec),
                                                rec.stride(0), _dev(tus), n, None if lfnst_tus is None else _dev(lfnst_tus), _dev(coeff),
                                                _dev(has), _dev(ws), ws.numel() * 8, _stream()), "uvghip_quantize_residual_batch")
    return coeff, has


def coeff_cost_batch(coeff, color, models):
    """coeff (n, h, w) int16 levels -> (bits (n,) float64, flags (n,) uint8): the CABAC bit cost of every block starting from
    `models` (lib.CabacModels or its 1220 bytes), uvg_get_coeff_cost's CABAC branch."""
    import ctypes
    L = _lib.init(coeff.device.index or 0)
    n, h, w = coeff.shape
    if not isinstance(models, _lib.CabacModels):
        models = _lib.CabacModels.from_buffer_copy(bytes(np.asarray(models, np.uint8).tobytes()))
    bits = torch.empty(n, dtype=torch.float64, device=coeff.device)
    flags = torch.empty(n, dtype=torch.uint8, device=coeff.device)
    _lib.check(L.uvghip_coeff_cost_batch(_dev(coeff), w, h, n, color, ctypes.byref(models), _dev(bits), _dev(flags), _stream()),
               "uvghip_coeff_cost_batch")
    return bits, flags


def quant_cbcr_residual_batch(u_orig, v_orig, u_pred, v_pred, u_rec, v_rec, tus, width, height, bitdepth, joint_cb_cr, jccr_sign,
                              qp_scaled=22, slice_is_intra=True, cu_type=1, rdoq=False, rdoq_skip=False, cbf_u=0, lam=0.0, ctx=None,
                              early_skip=False, lfnst_idx=0, lfnst_tus=None):
    """uvg_quant_cbcr_residual (joint Cb-Cr coding) for the TUs at `tus` -> (coeff (n, h, w) int16, ret (n,) uint8 = joint_cb_cr
    where the block has coefficients else 0); u_rec / v_rec written in place."""
    import ctypes
    L = _lib.init(u_orig.device.index or 0)
    n = tus.shape[0]
    p = _lib.QrParams()
    p.width, p.height, p.color = width, height, 1
    p.qp_scaled, p.slice_is_intra, p.cu_type = qp_scaled, int(slice_is_intra), cu_type
    p.rdoq_enable, p.rdoq_skip, p.cbf_u, p.lfnst_idx = int(rdoq), int(rdoq_skip), cbf_u, lfnst_idx
    p.lambda_ = float(lam)
    if ctx is not None:
        ctypes.memmove(p.ctx, np.asarray(ctx, np.uint8).tobytes(), 244)
    need = L.uvghip_quant_cbcr_residual_workspace_bytes(ctypes.byref(p), n)
    ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=u_orig.device)
    coeff = torch.empty((n, height, width), dtype=torch.int16, device=u_orig.device)
    ret = torch.empty(n, dtype=torch.uint8, device=u_orig.device)
    _lib.check(L.uvghip_quant_cbcr_residual_batch(bitdepth, ctypes.byref(p), joint_cb_cr, int(jccr_sign), _dev(u_orig), _dev(v_orig),
                                                  u_orig.stride(0), _dev(u_pred), _dev(v_pred), u_pred.stride(0), _dev(u_rec), _dev(v_rec),
                                                  u_rec.stride(0), _dev(tus), n, None if lfnst_tus is None else _dev(lfnst_tus), _dev(coeff),
                                                  _dev(ret), int(early_skip), _dev(ws), ws.numel() * 8, _stream()),
               "uvghip_quant_cbcr_residual_batch")
    return coeff, ret


def quant_signhide_batch(coef, bitdepth, qp_scaled, transform_skip=False, slice_is_intra=True, lfnst_idx=0):
    """uvg_quant with sign-data hiding on (n, h, w) int16 blocks -> levels."""
    L = _lib.init(coef.device.index or 0)
    n, h, w = coef.shape
    out = torch.empty_like(coef)
    _lib.check(L.uvghip_quant_signhide_batch(bitdepth, _dev(coef), _dev(out), w, h, n, qp_scaled, int(transform_skip), int(slice_is_intra), lfnst_idx,
                                             _stream()), "uvghip_quant_signhide_batch")
    return out


# ---- closed-loop intra search of whole pictures (include/uvg266_hip.h part 4) ----------------------------------------------
def ctu_params(pic_w, pic_h, qp, qp_c=None, lam=None, depth_min=1, depth_max=4, combine_intra_cus=1, rough_levels=2, rd=0):
    """uvghip_ctu_params_t for --preset medium -p 1: lambda = 0.57 * 2^((qp - 12) / 3) (src/rate_control.c qp_to_lambda for an intra
    picture), the default chroma QP table (identity), chroma weights from the luma / chroma QP distance."""
    qp_c = qp if qp_c is None else qp_c
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0) if lam is None else lam
    w = 2.0 ** ((qp - qp_c) / 3.0)
    return _lib.CtuParams(pic_w, pic_h, qp, qp_c, depth_min, depth_max, 1, combine_intra_cus, rough_levels, rd,
                          lam, float(np.sqrt(lam)), lam / w, w, w, lam / w)


class CtuSearch:
    """Device buffers of n pictures of one size and the call that searches them all (uvghip_ctu_search_intra).
    src: list of (y, u, v) device planes.  Outputs stay on the device: rec[i] = (y, u, v), cu[i] (uvghip_scu_t table as bytes,
    [rows of 4x4, 16 * CTUs per row, 32]), coeff[i] ([CTUs, 6144] int16), models[i] ([CTUs, 3, 257] uint32 as int32)."""

    def __init__(self, params, src, _make_plan=True, rows=None):
        """rows = (ctu_row0, ctu_row1): search only that band of CTU rows of every picture (uvghip_ctu_plan_create_rows; the caller
        provides the halo of the row above in rec / cu / models, bands.BandLayout.halo_search)."""
        self.P = params
        self.n = len(src)
        W, H = params.pic_w, params.pic_h
        self.wc, self.hc = (W + 63) // 64, (H + 63) // 64
        dev = src[0][0].device
        self.depth = _depth(src[0][0])
        self.L = _lib.init(dev.index or 0)
        self.src = src
        self.rec = [tuple(torch.zeros_like(p) for p in s) for s in src]
        ctus = self.wc * self.hc
        self.cu = [torch.zeros((self.hc * 16, self.wc * 16, 32), dtype=torch.uint8, device=dev) for _ in src]
        self.coeff = [torch.zeros((ctus, 6144), dtype=torch.int16, device=dev) for _ in src]
        self.models = [torch.zeros((ctus, 3, 257), dtype=torch.int32, device=dev) for _ in src]
        self.pics = (_lib.CtuPicture * self.n)()
        for i, (s, r) in enumerate(zip(src, self.rec)):
            self.pics[i] = _lib.CtuPicture(_dev(s[0]), _dev(s[1]), _dev(s[2]), s[0].stride(0), s[1].stride(0), _dev(r[0]), _dev(r[1]), _dev(r[2]),
                                           r[0].stride(0), r[1].stride(0), _dev(self.cu[i]), self.wc * 16, 0, _dev(self.coeff[i]), _dev(self.models[i]))

        self.plan = None
        if not _make_plan:          # ClosedLoop: uvghip_loop_plan_create builds the search plan inside its own workspace
            return
        import ctypes
        self.ws = torch.empty(self.L.uvghip_ctu_search_workspace_bytes(self.n, W, H), dtype=torch.uint8, device=dev)
        self.plan = ctypes.c_void_p()
        if rows is None:
            _lib.check(self.L.uvghip_ctu_plan_create(self.depth, ctypes.byref(self.P), self.pics, self.n, _dev(self.ws), ctypes.byref(self.plan)),
                       "uvghip_ctu_plan_create")
        else:
            _lib.check(self.L.uvghip_ctu_plan_create_rows(self.depth, ctypes.byref(self.P), self.pics, self.n, int(rows[0]), int(rows[1]), _dev(self.ws),
                                                          ctypes.byref(self.plan)), "uvghip_ctu_plan_create_rows")

    def run(self, stream=None):
        """Enqueue the search of all n pictures on the current (or the given) stream; returns at once (uvghip_ctu_plan_run)."""
        _lib.check(self.L.uvghip_ctu_plan_run(self.plan, _stream() if stream is None else stream), "uvghip_ctu_plan_run")

    def run_oneshot(self, stream=None):
        """uvghip_ctu_search_intra: plan + run + wait, in one call."""
        import ctypes
        rc = self.L.uvghip_ctu_search_intra(self.depth, ctypes.byref(self.P), self.pics, self.n, _dev(self.ws), _stream() if stream is None else stream)
        _lib.check(rc, "uvghip_ctu_search_intra")

    def __del__(self):
        plan, self.plan = getattr(self, "plan", None), None
        if plan:
            try:
                if torch is not None:          # (None at interpreter shutdown)
                    torch.cuda.synchronize()
            finally:
                self.L.uvghip_ctu_plan_destroy(plan)


class FramePool:
    """uvghip_frame_pool_*: all-intra pictures from HOST memory one by one, as uvg_encode_one_frame hands them over
    (src/encoderstate.c:2051-2091; csrc/frame_host.hip, csrc/shim/frame-hip.c): n_slots pictures in flight, the frames that are begun
    before one is asked for share a launch of at most group_max pictures.  begin(slot, params, (y, u, v)) with numpy planes;
    finish(slot) -> ((y, u, v) the picture after deblocking + SAO, [bytes of WPP row 0, row 1, ...])."""

    def __init__(self, params, depth, n_slots, group_max, sao_type=3, device=0, tiles=None):
        """tiles = (columns' widths, rows' heights) in CTUs: frames under --tiles (uvghip_frame_pool_create_tiles); finish() then returns the
        substreams of all tiles in the order of the bitstream."""
        import ctypes
        self.L = _lib.init(device)
        self.depth, self.w, self.h = depth, int(params.pic_w), int(params.pic_h)
        self.pool = ctypes.c_void_p()
        if tiles is None:
            _lib.check(self.L.uvghip_frame_pool_create(depth, ctypes.byref(params), sao_type, n_slots, group_max, ctypes.byref(self.pool)), "uvghip_frame_pool_create")
        else:
            cols, rows = (np.ascontiguousarray(a, np.int32) for a in tiles)
            _lib.check(self.L.uvghip_frame_pool_create_tiles(depth, ctypes.byref(params), sao_type, n_slots, group_max, cols.ctypes.data_as(ctypes.c_void_p), cols.size,
                                                             rows.ctypes.data_as(ctypes.c_void_p), rows.size, ctypes.byref(self.pool)), "uvghip_frame_pool_create_tiles")

    def begin(self, slot, params, yuv):
        import ctypes
        dt = np.uint8 if self.depth == 8 else np.uint16
        y, u, v = (np.ascontiguousarray(p, dtype=dt) for p in yuv)
        _lib.check(self.L.uvghip_frame_pool_begin(self.pool, slot, ctypes.byref(params), y.ctypes.data_as(ctypes.c_void_p), u.ctypes.data_as(ctypes.c_void_p),
                                                  v.ctypes.data_as(ctypes.c_void_p), y.shape[1], u.shape[1]), "uvghip_frame_pool_begin")

    def finish(self, slot):
        import ctypes
        dt = np.uint8 if self.depth == 8 else np.uint16
        y, u, v = np.empty((self.h, self.w), dt), np.empty((self.h // 2, self.w // 2), dt), np.empty((self.h // 2, self.w // 2), dt)
        rows, row_bytes, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(0)
        _lib.check(self.L.uvghip_frame_pool_finish(self.pool, slot, y.ctypes.data_as(ctypes.c_void_p), u.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p),
                                                   self.w, self.w // 2, ctypes.byref(rows), ctypes.byref(row_bytes), ctypes.byref(n)), "uvghip_frame_pool_finish")
        nb = np.ctypeslib.as_array(ctypes.cast(row_bytes, ctypes.POINTER(ctypes.c_int32)), (n.value,)).copy()
        data = ctypes.string_at(rows.value, int(nb.sum()))
        at = np.concatenate([[0], np.cumsum(nb)])
        return (y, u, v), [data[at[r]:at[r + 1]] for r in range(n.value)]

    def __del__(self):
        pool, self.pool = getattr(self, "pool", None), None
        if pool:
            self.L.uvghip_frame_pool_destroy(pool)


class ClosedLoop(CtuSearch):
    """uvghip_loop_plan_*: the search of CtuSearch followed by the in-loop filters on the reference's schedule, one call per
    group of pictures.  Adds out[i] = (y, u, v), the pictures after deblocking + SAO, and after a run sao_info ([n, ctus, 2, 17]
    int32) / sao_models ([n, ctus, 6] as int16 storage of uint16) as device tensors over the plan's own buffers."""

    def __init__(self, params, src, sao_type=3):
        import ctypes
        super().__init__(params, src, _make_plan=False)
        dev = src[0][0].device
        self.out = [tuple(torch.empty_like(p) for p in s) for s in src]
        W, H = params.pic_w, params.pic_h
        self.loop_ws = torch.empty(self.L.uvghip_loop_workspace_bytes(self.depth, self.n, W, H), dtype=torch.uint8, device=dev)
        lp = (_lib.LoopPicture * self.n)()
        for i in range(self.n):
            o = self.out[i]
            lp[i] = _lib.LoopPicture(self.pics[i], _dev(o[0]), _dev(o[1]), _dev(o[2]), o[0].stride(0), o[1].stride(0))
        self.loop_pics = lp
        self.loop = ctypes.c_void_p()
        _lib.check(self.L.uvghip_loop_plan_create(self.depth, ctypes.byref(self.P), lp, self.n, sao_type, _dev(self.loop_ws), ctypes.byref(self.loop)),
                   "uvghip_loop_plan_create")

    def run(self, stream=None):
        """Enqueue search + filters of all n pictures (uvghip_loop_plan_run); returns at once."""
        _lib.check(self.L.uvghip_loop_plan_run(self.loop, _stream() if stream is None else stream), "uvghip_loop_plan_run")

    def run_overlapped(self, stream=None):
        """uvghip_loop_plan_run_overlapped: the same results, the filter stage and the coder BESIDE the search (behind its per-CTU flags) instead
        of after it -- the latency of one group; returns at once."""
        _lib.check(self.L.uvghip_loop_plan_run_overlapped(self.loop, _stream() if stream is None else stream), "uvghip_loop_plan_run_overlapped")

    def run_search(self, stream=None):
        _lib.check(self.L.uvghip_loop_plan_run_search(self.loop, _stream() if stream is None else stream), "uvghip_loop_plan_run_search")

    def run_filters(self, stream=None):
        _lib.check(self.L.uvghip_loop_plan_run_filters(self.loop, _stream() if stream is None else stream), "uvghip_loop_plan_run_filters")

    def results(self):
        """-> (sao_info [n, ctus, 2, 17] int32, sao_models [n, ctus, 6] uint16) copied to the host (synchronises)."""
        import ctypes
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(self.L.uvghip_loop_plan_results(self.loop, ctypes.byref(a), ctypes.byref(b)), "uvghip_loop_plan_results")
        torch.cuda.synchronize()
        ctus = self.wc * self.hc
        base = self.loop_ws.data_ptr()
        info = self.loop_ws[a.value - base:a.value - base + self.n * ctus * 34 * 4].cpu().numpy().view(np.int32).reshape(self.n, ctus, 2, 17)
        models = self.loop_ws[b.value - base:b.value - base + self.n * ctus * 6 * 2].cpu().numpy().view(np.uint16).reshape(self.n, ctus, 6)
        return info, models

    def sao_device(self):
        """Device views of the plan's SAO decisions ([n * ctus * 34] int32) and SAO models ([n * ctus * 6] as int16)."""
        import ctypes
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(self.L.uvghip_loop_plan_results(self.loop, ctypes.byref(a), ctypes.byref(b)), "uvghip_loop_plan_results")
        ctus, base = self.wc * self.hc, self.loop_ws.data_ptr()
        return (self.loop_ws[a.value - base:a.value - base + self.n * ctus * 34 * 4].view(torch.int32),
                self.loop_ws[b.value - base:b.value - base + self.n * ctus * 6 * 2].view(torch.int16))

    def picture_nals(self, picture, poc):
        """uvghip_loop_plan_picture_nals: after run(), the slice NAL + hash SEI of picture `picture` as picture `poc` of the stream -> bytes."""
        import ctypes
        rows = self.hc * (3 * 64 * int(self.P.pic_w)) * (1 if self.depth == 8 else 2) + 64 + 4 * self.hc
        buf = np.zeros(rows, np.uint8)
        n = ctypes.c_size_t(0)
        _lib.check(self.L.uvghip_loop_plan_picture_nals(self.loop, picture, poc, buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(n), _stream()),
                   "uvghip_loop_plan_picture_nals")
        return buf[:n.value].tobytes()

    def group_nals(self, first_poc=0):
        """uvghip_loop_plan_group_nals: after run(), the NAL units of every picture of the group as pictures first_poc, first_poc + 1, ...
        -> list of bytes (what picture_nals(i, first_poc + i) returns for each i), with one download for the whole group."""
        import ctypes
        per = self.hc * (3 * 64 * int(self.P.pic_w)) * (1 if self.depth == 8 else 2) + 64 + 4 * self.hc
        if getattr(self, "_nal_buf", None) is None or self._nal_buf.size < per * self.n:
            self._nal_buf = np.empty(per * self.n, np.uint8)
        lens = (ctypes.c_size_t * self.n)()
        _lib.check(self.L.uvghip_loop_plan_group_nals(self.loop, first_poc, self._nal_buf.ctypes.data_as(ctypes.c_void_p), self._nal_buf.size, lens, _stream()),
                   "uvghip_loop_plan_group_nals")
        out, at = [], 0
        for i in range(self.n):
            out.append(self._nal_buf[at:at + lens[i]].tobytes())
            at += lens[i]
        return out

    def slice_data(self):
        """The rows' substreams the plan coded as the last thing of run(): (rows [n, n_rows, row_cap] uint8, row_bytes [n, n_rows] int32)
        as device views of the plan's buffers."""
        import ctypes
        a, b, cap, nr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self.L.uvghip_loop_plan_slice_data(self.loop, ctypes.byref(a), ctypes.byref(b), ctypes.byref(cap), ctypes.byref(nr)), "uvghip_loop_plan_slice_data")
        base = self.loop_ws.data_ptr()
        rows = self.loop_ws[a.value - base:a.value - base + self.n * nr.value * cap.value].view(self.n, nr.value, cap.value)
        nbytes = self.loop_ws[b.value - base:b.value - base + self.n * nr.value * 4].view(torch.int32).view(self.n, nr.value)
        return rows, nbytes

    def encode_rows(self, row_cap=None, sao=True, stream=None):
        """uvghip_encode_slice_rows on the plan's pictures (after run()): -> (out [n, rows, row_cap] uint8, row_bytes [n, rows] int32),
        device tensors; the slice data of picture i is out[i, r, :row_bytes[i, r]] for r = 0, 1, ..."""
        W, H = self.P.pic_w, self.P.pic_h
        row_cap = 3 * 64 * W * (1 if self.depth == 8 else 2) if row_cap is None else row_cap      # (the library's own bound: twice the bytes at 10 bit)
        dev = self.loop_ws.device
        if not hasattr(self, "_rows") or self._rows[0].shape[2] != row_cap:
            self._rows = (torch.empty((self.n, self.hc, row_cap), dtype=torch.uint8, device=dev), torch.zeros((self.n, self.hc), dtype=torch.int32, device=dev),
                          torch.empty(self.L.uvghip_slice_rows_workspace_bytes(self.n), dtype=torch.uint8, device=dev))
        out, nbytes, ws = self._rows
        info, models = self.sao_device() if sao else (None, None)
        import ctypes
        _lib.check(self.L.uvghip_encode_slice_rows(self.depth, ctypes.byref(self.P), self.pics, self.n, None if info is None else _dev(info),
                                                   None if models is None else _dev(models), _dev(ws), _dev(out), row_cap, _dev(nbytes),
                                                   _stream() if stream is None else stream), "uvghip_encode_slice_rows")
        return out, nbytes

    def encode_rows_alf(self, alf, row_cap=None, stream=None):
        """uvghip_encode_slice_rows_alf on the plan's pictures (after run()): the slice data with the CTU-level ALF syntax.  alf: per picture a dict
        with alf_type, enabled (3), n_luma_aps, n_alternatives_chroma, cc_enabled (2), cc_filter_count (2), ctu_flags (u8 [7][ctus]) and
        filter_set_idx (i16 [ctus]) as numpy arrays.  -> (out, row_bytes) like encode_rows."""
        W, H = self.P.pic_w, self.P.pic_h
        row_cap = 3 * 64 * W * (1 if self.depth == 8 else 2) if row_cap is None else row_cap      # (the library's own bound: twice the bytes at 10 bit)
        dev = self.loop_ws.device
        out = torch.empty((self.n, self.hc, row_cap), dtype=torch.uint8, device=dev)
        nbytes = torch.zeros((self.n, self.hc), dtype=torch.int32, device=dev)
        ws = torch.empty(self.L.uvghip_slice_rows_alf_workspace_bytes(self.n), dtype=torch.uint8, device=dev)
        info, models = self.sao_device()
        A = (_lib.SliceAlf * self.n)()
        keep = []
        for i, a in enumerate(alf):
            A[i].alf_type, A[i].n_luma_aps, A[i].n_alternatives_chroma = int(a["alf_type"]), int(a["n_luma_aps"]), int(a["n_alternatives_chroma"])
            for c in range(3):
                A[i].enabled[c] = int(a["enabled"][c])
            for c in range(2):
                A[i].cc_enabled[c] = int(a["cc_enabled"][c]); A[i].cc_filter_count[c] = int(a["cc_filter_count"][c])
            fl = torch.from_numpy(np.ascontiguousarray(a["ctu_flags"], np.uint8)).to(dev)
            si = torch.from_numpy(np.ascontiguousarray(a["filter_set_idx"], np.int16)).to(dev)
            keep += [fl, si]
            A[i].ctu_flags, A[i].filter_set_idx = _dev(fl), _dev(si)
        _lib.check(self.L.uvghip_encode_slice_rows_alf(self.depth, ctypes.byref(self.P), self.pics, A, self.n, _dev(info), _dev(models), _dev(ws), _dev(out), row_cap,
                                                       _dev(nbytes), _stream() if stream is None else stream), "uvghip_encode_slice_rows_alf")
        torch.cuda.synchronize()          # (the decisions' device copies go out of scope with this call)
        return out, nbytes

    def alf_stage(self, decide, source=None, classification_shift=None):
        """uvghip_loop_plan_alf_stage: the ALF stage of the picture loop (BASELINE configs[3]: --alf full), after run(): for every picture of the plan
          1. the frame's ALF statistics from the loop's SAO output on the device, handed over on demand (AlfStatistics: classification,
             luma covariance per class and CTU, chroma covariance, CC-ALF covariance -- what alf_derive_stats_for_filtering leaves, alf.c:4227),
          2. `decide(i, stats)` on the host -> the picture's decisions: the derivation the reference keeps in alf_encoder / alf_encoder_ctb /
             derive_cc_alf_filter (alf.c:3994, 4369, 2212); the tests replay a real run's.  A dict with alf_type (1 no-cc, 2 full), enabled
             (3), n_luma_aps, luma_aps (i16 [k][677]), chroma_aps (i16 [114], [112] = the number of alternatives), cc_enabled (2),
             cc_filter_count (2), cc_coeff (i16 [2][4][8]), ctu_flags (u8 [7][ctus]), filter_set_idx (i16 [ctus]) -- the layouts of
             uvghip_alf_picture_t / uvghip_slice_alf_t,
          3. uvghip_alf_reconstruct_picture on the SAO output -> alf_out[i], the picture the encoder returns and hashes,
          4. uvghip_encode_slice_rows_alf: the slice data with the CTU-level ALF syntax.
        -> (alf_out: per picture three device planes, rows [n, n_rows, row_cap] uint8, row_bytes [n, n_rows] int32).  source: the plan's
        source pictures (the statistics compare with them); classification_shift: cfg.input_bitdepth + 4 (alf.c:5185; default depth + 4)."""
        shift = self.depth + 4 if classification_shift is None else classification_shift
        dev = self.loop_ws.device
        W, H = int(self.P.pic_w), int(self.P.pic_h)
        self.alf_out = [[torch.empty_like(p) for p in self.out[i]] for i in range(self.n)]
        planes = (_lib.AlfPlanes * self.n)()
        for i, o in enumerate(self.alf_out):
            planes[i] = _lib.AlfPlanes(_dev(o[0]), _dev(o[1]), _dev(o[2]), o[0].stride(0), o[1].stride(0))
        keep, failure = [], []

        def trampoline(user, i, picture, decision):          # uvghip_alf_decide_fn: the library calls back once per picture
            try:
                stats = AlfStatistics(self.out[i], None if source is None else source[i], W, H, shift)
                d = decide(i, stats)
                opt = lambda key, dt: np.zeros(0, dt) if d.get(key) is None else np.ascontiguousarray(d[key], dt)      # (a disabled component's arrays may be absent / None)
                arr = dict(luma_aps=np.ascontiguousarray(np.asarray(opt("luma_aps", np.int16)).reshape(-1, 677)[:int(d.get("n_luma_aps", 0))]) if int(d.get("n_luma_aps", 0)) else np.zeros(0, np.int16),
                           chroma_aps=opt("chroma_aps", np.int16), cc_coeff=opt("cc_coeff", np.int16), ctu_flags=opt("ctu_flags", np.uint8),
                           filter_set_idx=opt("filter_set_idx", np.int16))
                keep[:] = [arr]                               # (valid until the next call, as the ABI asks)
                o = decision.contents
                o.alf_type, o.n_luma_aps = int(d["alf_type"]), int(d.get("n_luma_aps", 0))
                for c in range(3):
                    o.enabled[c] = int(d["enabled"][c])
                for c in range(2):
                    o.cc_enabled[c], o.cc_filter_count[c] = int(d.get("cc_enabled", (0, 0))[c]), int(d.get("cc_filter_count", (0, 0))[c])
                for k, a in arr.items():
                    setattr(o, k, a.ctypes.data if a.size else None)
                return 0
            except Exception as e:                            # noqa: BLE001 -- an exception must not unwind through the C frames
                failure.append(e)
                return 1
        cb = _lib.ALF_DECIDE_FN(trampoline)
        row_cap = 3 * 64 * W * (1 if self.depth == 8 else 2)
        rows = torch.empty((self.n, self.hc, row_cap), dtype=torch.uint8, device=dev)
        nbytes = torch.zeros((self.n, self.hc), dtype=torch.int32, device=dev)
        ws = torch.empty(self.L.uvghip_loop_plan_alf_workspace_bytes(self.loop), dtype=torch.uint8, device=dev)
        rc = self.L.uvghip_loop_plan_alf_stage(self.loop, cb, None, shift, ctypes.byref(planes), _dev(ws), _dev(rows), row_cap, _dev(nbytes), _stream())
        if failure:
            raise failure[0]
        _lib.check(rc, "uvghip_loop_plan_alf_stage")
        torch.cuda.synchronize()          # (the stage's workspace goes out of scope with this call)
        return self.alf_out, rows, nbytes

    def __del__(self):
        loop, self.loop = getattr(self, "loop", None), None
        if loop:
            try:
                if torch is not None:          # (None at interpreter shutdown)
                    torch.cuda.synchronize()
            finally:
                self.L.uvghip_loop_plan_destroy(loop)
        if CtuSearch is not None:
            CtuSearch.__del__(self)


def tile_grid(pic_w, pic_h, cols, rows):
    """uvghip_tile_grid (host function): the uniform grid of --tiles <cols>x<rows> -> (rects [cols * rows, 4] int32 (x, y, w, h in samples,
    raster order of the tiles), first_ctu [cols * rows] int32: the tile-scan address of every tile's first CTU).  cols / rows may be
    sequences instead: the columns' widths / the rows' heights in CTUs (uvghip_tile_grid_split: --tiles-width-split / --tiles-height-split)."""
    L = _lib.load_library()
    if np.ndim(cols) or np.ndim(rows):
        wc, hc = (pic_w + 63) // 64, (pic_h + 63) // 64
        cw = np.ascontiguousarray(cols if np.ndim(cols) else [(i + 1) * wc // cols - i * wc // cols for i in range(cols)], np.int32)
        rh = np.ascontiguousarray(rows if np.ndim(rows) else [(i + 1) * hc // rows - i * hc // rows for i in range(rows)], np.int32)
        rects = np.zeros((len(cw) * len(rh), 4), np.int32)
        first = np.zeros(len(cw) * len(rh), np.int32)
        _lib.check(L.uvghip_tile_grid_split(pic_w, pic_h, cw.ctypes.data, len(cw), rh.ctypes.data, len(rh), rects.ctypes.data, first.ctypes.data), "uvghip_tile_grid_split")
        return rects, first
    rects = np.zeros((cols * rows, 4), np.int32)
    first = np.zeros(cols * rows, np.int32)
    _lib.check(L.uvghip_tile_grid(pic_w, pic_h, cols, rows, rects.ctypes.data, first.ctypes.data), "uvghip_tile_grid")
    return rects, first


class TiledLoop:
    """uvghip_tiles_plan_*: all-intra pictures under --tiles <cols>x<rows> --wpp -- every tile an independent rectangle of the closed loop
    (search, in-loop filters, slice data), the tiles of a picture beside each other on the device.  src: list of (y, u, v) whole-picture
    device planes.  Outputs stay on the device: rec[i] / out[i] (whole pictures: before / after the in-loop filters), cu[i] (picture
    raster), coeff[i] / models[i] (the CTUs in TILE-SCAN order: the order of the bitstream)."""

    def __init__(self, params, src, tiles, sao_type=3, owned=None):
        """owned: per tile (raster order) whether THIS device searches, filters and codes it (uvghip_tiles_plan_create_owned: the tiles of a
        picture over the devices of a node, uvg266_amd.tiles); None = all."""
        import ctypes
        self.P, self.n = params, len(src)
        W, H = int(params.pic_w), int(params.pic_h)
        self.wc, self.hc = (W + 63) // 64, (H + 63) // 64
        # tiles = (cols, rows): the uniform grid of --tiles; either may be a sequence -- the columns' widths / the rows' heights in CTUs
        # (--tiles-width-split / --tiles-height-split)
        uni = lambda n, parts: [(i + 1) * n // parts - i * n // parts for i in range(parts)]
        self.col_ctus = np.ascontiguousarray(tiles[0] if np.ndim(tiles[0]) else uni(self.wc, int(tiles[0])), np.int32)
        self.row_ctus = np.ascontiguousarray(tiles[1] if np.ndim(tiles[1]) else uni(self.hc, int(tiles[1])), np.int32)
        self.cols, self.rows = len(self.col_ctus), len(self.row_ctus)
        dev = src[0][0].device
        self.depth = _depth(src[0][0])
        self.L = _lib.init(dev.index or 0)
        self.src = src
        self.rec = [tuple(torch.zeros(p.shape, dtype=p.dtype, device=dev) for p in s) for s in src]
        self.out = [tuple(torch.zeros(p.shape, dtype=p.dtype, device=dev) for p in s) for s in src]
        ctus = self.wc * self.hc
        self.cu = [torch.zeros((self.hc * 16, self.wc * 16, 32), dtype=torch.uint8, device=dev) for _ in src]
        self.coeff = [torch.zeros((ctus, 6144), dtype=torch.int16, device=dev) for _ in src]
        self.models = [torch.zeros((ctus, 3, 257), dtype=torch.int32, device=dev) for _ in src]
        lp = (_lib.LoopPicture * self.n)()

        def _plane(t):          # the source planes may be views into larger allocations: rows contiguous, any stride between them
            assert t.is_cuda and t.dim() == 2 and t.stride(1) == 1, "device-resident plane with contiguous rows required"
            return ctypes.c_void_p(t.data_ptr())
        for i, (s, r, o) in enumerate(zip(src, self.rec, self.out)):
            sp = _lib.CtuPicture(_plane(s[0]), _plane(s[1]), _plane(s[2]), s[0].stride(0), s[1].stride(0), _dev(r[0]), _dev(r[1]), _dev(r[2]),
                                 r[0].stride(0), r[1].stride(0), _dev(self.cu[i]), self.wc * 16, 0, _dev(self.coeff[i]), _dev(self.models[i]))
            lp[i] = _lib.LoopPicture(sp, _dev(o[0]), _dev(o[1]), _dev(o[2]), o[0].stride(0), o[1].stride(0))
        self.pics = lp
        self.owned = None if owned is None else np.ascontiguousarray(np.asarray(owned) != 0, np.uint8)
        if self.owned is not None and self.owned.size != self.cols * self.rows:
            raise ValueError("owned: one entry per tile")
        own = None if self.owned is None else self.owned.ctypes.data
        nbytes = self.L.uvghip_tiles_workspace_bytes_split(self.depth, self.n, W, H, self.col_ctus.ctypes.data, self.cols, self.row_ctus.ctypes.data, self.rows, own)
        if not nbytes:
            raise ValueError("uvghip_tiles_workspace_bytes: the tile grid does not fit the picture (or no tile is owned)")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.plan = ctypes.c_void_p()
        _lib.check(self.L.uvghip_tiles_plan_create_split(self.depth, ctypes.byref(self.P), lp, self.n, self.col_ctus.ctypes.data, self.cols, self.row_ctus.ctypes.data,
                                                         self.rows, own, sao_type, _dev(self.ws), ctypes.byref(self.plan)), "uvghip_tiles_plan_create_split")
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self.L.uvghip_tiles_plan_layout(self.plan, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "uvghip_tiles_plan_layout")
        self.n_tiles, self.n_classes, self.n_substreams = a.value, b.value, c.value

    def run(self, stream=None):
        """Enqueue search + filters + slice data of every tile of every picture (uvghip_tiles_plan_run); returns at once."""
        _lib.check(self.L.uvghip_tiles_plan_run(self.plan, _stream() if stream is None else stream), "uvghip_tiles_plan_run")

    def nals(self, first_poc=0, first=0, count=None):
        """uvghip_tiles_plan_nals: after run(), slice NAL + hash SEI of pictures [first, first + count) as pictures first_poc, ... of the
        stream -> list of bytes."""
        import ctypes
        count = self.n - first if count is None else count
        per = self.hc * (3 * 64 * int(self.P.pic_w)) * (1 if self.depth == 8 else 2) + 64 + 4 * self.n_substreams      # (the row slots of the tiles of one CTU row add up to a picture row's)
        if getattr(self, "_nal_buf", None) is None or self._nal_buf.size < per * count:
            self._nal_buf = np.empty(per * count, np.uint8)
        lens = (ctypes.c_size_t * count)()
        _lib.check(self.L.uvghip_tiles_plan_nals(self.plan, first, count, first_poc, self._nal_buf.ctypes.data_as(ctypes.c_void_p), self._nal_buf.size, lens, _stream()),
                   "uvghip_tiles_plan_nals")
        out, at = [], 0
        for i in range(count):
            out.append(self._nal_buf[at:at + lens[i]].tobytes())
            at += lens[i]
        return out

    def substreams(self, first=0, count=None):
        """uvghip_tiles_plan_substreams: what this device contributes to the NAL units of pictures [first, first + count) after run():
        (lens [count, n_substreams] int32 -- 0 for the tiles of other devices --, the owned substreams' bytes in the order of the bitstream
        (uint8), sums [count, 3] uint32: the owned tiles' terms of the output pictures' checksums).  Over the devices lengths and sums add up."""
        import ctypes
        count = self.n - first if count is None else count
        lens = np.zeros((count, self.n_substreams), np.int32)
        sums = np.zeros((count, 3), np.uint32)
        cap = count * (self.hc * (3 * 64 * int(self.P.pic_w)) * (1 if self.depth == 8 else 2))
        if getattr(self, "_sub_buf", None) is None or self._sub_buf.size < cap:
            self._sub_buf = np.empty(cap, np.uint8)
        used = ctypes.c_size_t(0)
        _lib.check(self.L.uvghip_tiles_plan_substreams(self.plan, first, count, lens.ctypes.data, self._sub_buf.ctypes.data, self._sub_buf.size, ctypes.byref(used),
                                                       sums.ctypes.data, _stream()), "uvghip_tiles_plan_substreams")
        return lens, self._sub_buf[:used.value].copy(), sums

    def tile(self, picture, tile):
        """uvghip_tiles_plan_tile -> (loop plan handle, picture index there, (x, y, w, h), first CTU in tile scan)."""
        import ctypes
        pl, idx, first = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        rect = np.zeros(4, np.int32)
        _lib.check(self.L.uvghip_tiles_plan_tile(self.plan, picture, tile, ctypes.byref(pl), ctypes.byref(idx), rect.ctypes.data, ctypes.byref(first)), "uvghip_tiles_plan_tile")
        return pl, idx.value, tuple(int(v) for v in rect), first.value

    def __del__(self):
        plan, self.plan = getattr(self, "plan", None), None
        if plan:
            try:
                if torch is not None:          # (None at interpreter shutdown)
                    torch.cuda.synchronize()
            finally:
                self.L.uvghip_tiles_plan_destroy(plan)


def picture_checksum(y, u, v, stream=None):
    """uvghip_picture_checksum: the three plane sums of the decoded picture hash SEI (uvg_image_checksum) -> tensor [3] on the device
    (the uint32 values in an int32 tensor: .cpu().numpy().view(np.uint32))."""
    depth = 8 if y.dtype == torch.uint8 else 10
    sums = torch.empty(4, dtype=torch.int32, device=y.device)
    L = _lib.init(y.device.index or 0)
    _lib.check(L.uvghip_picture_checksum(depth, _dev(y), y.stride(0), _dev(u), _dev(v), u.stride(0), y.shape[1], y.shape[0], _dev(sums),
                                         _stream() if stream is None else stream), "uvghip_picture_checksum")
    return sums[:3]


# ---- P / B pictures: candidate lists ----------------------------------------------------------------------------------------
def merge_cand_batch(ctx, lcu, col, hmvp, stream=None):
    """uvghip_merge_cand_batch: ctx [n, 64], lcu [n, 290, 8] (modified in place like the reference's lcu_t), col [n, L] (or [L]: one
    collocated picture for all calls), hmvp [n, 41], int32 device tensors -> (cands [n, 6, 7], counts [n])."""
    n = ctx.shape[0]
    cands = torch.zeros((n, 6, 7), dtype=torch.int32, device=ctx.device)
    counts = torch.zeros((n,), dtype=torch.int32, device=ctx.device)
    L = _lib.init(ctx.device.index or 0)
    _lib.check(L.uvghip_merge_cand_batch(_dev(ctx), _dev(lcu), _dev(col), col.shape[1] if col.dim() == 2 else 0, _dev(hmvp), n, _dev(cands), _dev(counts),
                                         _stream() if stream is None else stream), "uvghip_merge_cand_batch")
    return cands, counts


def amvp_cand_batch(ctx, lcu, col, hmvp, stream=None):
    """uvghip_amvp_cand_batch -> mv_cand [n, 2, 2] int32."""
    n = ctx.shape[0]
    out = torch.zeros((n, 2, 2), dtype=torch.int32, device=ctx.device)
    L = _lib.init(ctx.device.index or 0)
    _lib.check(L.uvghip_amvp_cand_batch(_dev(ctx), _dev(lcu), _dev(col), col.shape[1] if col.dim() == 2 else 0, _dev(hmvp), n, _dev(out),
                                        _stream() if stream is None else stream), "uvghip_amvp_cand_batch")
    return out


ME_JOB_NP = np.dtype([("x", "<i4"), ("y", "<i4"), ("ref", "<i4"), ("mv_cand", "<i4", (2, 2)), ("extra_mv", "<i4", (2,)), ("n_start", "<i4"), ("start", "<i4", (6, 2))])
ME_RESULT_NP = np.dtype([("mv", "<i4", (2,)), ("int_mv", "<i4", (2,)), ("cost", "<f8"), ("bits", "<f8"), ("int_cost", "<f8"), ("int_bits", "<f8"), ("mv_cand", "<i4"),
                         ("skipped_hexagon", "<i4")])


def ref_table(planes):
    """A device array of pointers to the reference pictures' luma planes (what uvghip_me_search_batch takes as refs_dev)."""
    return torch.tensor([p.data_ptr() for p in planes], dtype=torch.int64, device=planes[0].device)


def me_search_batch(cur, refs, ref_tab, jobs, size, lambda_sqrt, fme_level=4, stream=None):
    """uvghip_me_search_batch: cur [H, W] device plane, refs: list of [H, W] planes (kept alive by the caller), ref_tab = ref_table(refs),
    jobs: uint8 device tensor over n ME_JOB_NP records -> uint8 device tensor over n ME_RESULT_NP records."""
    n = jobs.numel() // ME_JOB_NP.itemsize
    out = torch.zeros(n * ME_RESULT_NP.itemsize, dtype=torch.uint8, device=cur.device)
    L = _lib.init(cur.device.index or 0)
    _lib.check(L.uvghip_me_search_batch(_depth(cur), _dev(cur), cur.stride(0), _dev(ref_tab), refs[0].stride(0), cur.shape[1], cur.shape[0], float(lambda_sqrt),
                                        fme_level, size, _dev(jobs), n, _dev(out), _stream() if stream is None else stream), "uvghip_me_search_batch")
    return out


MOTION_NP = np.dtype([("x", "<i4"), ("y", "<i4"), ("dir", "<i4"), ("ref", "<i4", (2,)), ("mv", "<i4", (2, 2))])


def inter_pred_satd_batch(cur, refs, ref_tab, cands, size, want_pred=False, stream=None):
    """uvghip_inter_pred_satd_batch: cands = uint8 device tensor over n MOTION_NP records -> (satd [n] int32, pred [n, size, size] or None)."""
    n = cands.numel() // MOTION_NP.itemsize
    satd = torch.zeros(n, dtype=torch.int32, device=cur.device)
    pred = torch.zeros((n, size, size), dtype=cur.dtype, device=cur.device) if want_pred else None
    L = _lib.init(cur.device.index or 0)
    _lib.check(L.uvghip_inter_pred_satd_batch(_depth(cur), _dev(cur), cur.stride(0), _dev(ref_tab), refs[0].stride(0), cur.shape[1], cur.shape[0], size, _dev(cands), n,
                                              _dev(satd), None if pred is None else _dev(pred), _stream() if stream is None else stream), "uvghip_inter_pred_satd_batch")
    return satd, pred


def ctu_search_pb(pictures, depth, stream=None):
    """uvghip_ctu_search_pb: the closed-loop CTU search of n independent P / B pictures.  pictures: list of lib.CtuPbPicture whose pointers
    refer to device tensors the caller keeps alive.  Returns the workspace tensor (in use until the stream has run the launch)."""
    n = len(pictures)
    L = _lib.init(torch.cuda.current_device())
    arr = (_lib.CtuPbPicture * n)(*pictures)
    W, H = pictures[0].params.pic_w, pictures[0].params.pic_h
    ws = torch.zeros(L.uvghip_ctu_search_pb_workspace_bytes(n, W, H), dtype=torch.uint8, device="cuda")
    _lib.check(L.uvghip_ctu_search_pb(depth, ctypes.byref(arr), n, _dev(ws), _stream() if stream is None else stream), "uvghip_ctu_search_pb")
    return ws


def loop_pb_run(pictures, depth, sao_type=3, stream=None):
    """uvghip_loop_pb_run: search + in-loop filters + slice data of n independent P / B pictures (list of lib.LoopPbPicture over device
    tensors the caller keeps alive) -> (workspace, sao_info [n, ctus, 34] int32, sao_models [n, ctus, 6] int16, rows [n, hc, row_cap]
    uint8, row_bytes [n, hc] int32), device views into the workspace."""
    n = len(pictures)
    L = _lib.init(torch.cuda.current_device())
    arr = (_lib.LoopPbPicture * n)(*pictures)
    W, H = pictures[0].search.params.pic_w, pictures[0].search.params.pic_h
    ws = torch.zeros(L.uvghip_loop_pb_workspace_bytes(depth, n, W, H), dtype=torch.uint8, device="cuda")
    _lib.check(L.uvghip_loop_pb_run(depth, ctypes.byref(arr), n, sao_type, _dev(ws), _stream() if stream is None else stream), "uvghip_loop_pb_run")
    a, b, c, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    cap, nr = ctypes.c_int(), ctypes.c_int()
    _lib.check(L.uvghip_loop_pb_results(depth, n, W, H, _dev(ws), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d), ctypes.byref(cap), ctypes.byref(nr)),
               "uvghip_loop_pb_results")
    ctus, base = ((W + 63) // 64) * ((H + 63) // 64), ws.data_ptr()
    view = lambda p, nbytes: ws[p.value - base:p.value - base + nbytes]
    return (ws, view(a, n * ctus * 34 * 4).view(torch.int32).view(n, ctus, 34), view(b, n * ctus * 6 * 2).view(torch.int16).view(n, ctus, 6),
            view(c, n * nr.value * cap.value).view(n, nr.value, cap.value), view(d, n * nr.value * 4).view(torch.int32).view(n, nr.value))


def reference_dag(frames):
    """The dependencies between the coded pictures of a sequence: frames = one dict per picture in CODING order with slice_type (2 I, 1 P,
    0 B), poc, n_refs, ref_pocs (the reference buffer the encoder coded the picture with: uvg_encoder_create_ref_lists) ->
    (deps, level): deps[f] = the coded pictures whose output picture f reads (the most recently coded picture of each POC in its buffer),
    level[f] = the length of the longest chain of dependent pictures before f.  Pictures of one level are independent of each other: the
    temporal layer of a random-access GOP, and other layers of neighbouring GOPs; a low-delay sequence has one picture per level."""
    deps, level, coded_as = [], [], {}
    for f, fs in enumerate(frames):
        deps.append([] if fs["slice_type"] == 2 else sorted({coded_as[fs["ref_pocs"][i]] for i in range(fs["n_refs"])}))
        level.append(1 + max([level[d] for d in deps[f]], default=-1))
        coded_as[fs["poc"]] = f
    return deps, level


class AlfStatistics:
    """What ClosedLoop.alf_stage hands the host's ALF derivation: the frame's statistics, computed on the device when asked for.
    rec: the picture ALF gets (deblocked + SAO), org: the source picture (three device planes each)."""

    def __init__(self, rec, org, width, height, shift):
        self.rec, self.org, self.W, self.H, self.shift = rec, org, width, height, shift
        wc, hc = (width + 63) // 64, (height + 63) // 64
        ry, rc = [], []
        for cy in range(hc):
            for cx in range(wc):
                x, y = cx * 64, cy * 64
                bw, bh = min(64, width - x), min(64, height - y)
                ry.append((x, y, bw, bh)); rc.append((x // 2, y // 2, bw // 2, bh // 2))
        dev = rec[0].device
        self.rects_y = torch.from_numpy(np.asarray(ry, np.int32)).to(dev)
        self.rects_c = torch.from_numpy(np.asarray(rc, np.int32)).to(dev)
        self._cls = None

    def classification(self):
        """uvghip_alf_classify_frame: class | transpose << 5 per 4x4 luma block (alf_derive_classification, alf.c:5157)."""
        if self._cls is None:
            self._cls = alf_classify_frame(self.rec[0], self.W, self.H, self.shift)
        return self._cls

    def luma(self):
        """Per CTU and class the covariance of get_blk_stats (alf.c:2378; -> ee, y, pix_acc as alf_stats_batch returns them)."""
        return alf_stats_batch(self.org[0], self.rec[0], self.rects_y, cls=self.classification(), pic_w=self.W, pic_h=self.H)

    def luma_frame(self, scratch=None):
        """The FRAME's luma covariance per class, as the derivation reads it first (alf.c:792-835 accumulates the CTUs'): compact per-CTU
        records (only the classes present in a CTU, the upper triangle of ee) summed on the device -> sums int64 [25][UVGHIP_ALF_SUM_WORDS = 1509]
        (the ee triangle, y widened, pix_acc).  scratch: a dict that keeps the record buffer between pictures (12 KB per CTU and class)."""
        L = _lib.init(self.rec[0].device.index or 0)
        n, dev = self.rects_y.shape[0], self.rec[0].device
        scratch = {} if scratch is None else scratch
        if scratch.get("n") != n:
            scratch.update(n=n, rec=torch.empty((n, 25, 1484), dtype=torch.int64, device=dev), present=torch.zeros(n, dtype=torch.int32, device=dev))
        sums = torch.zeros((25, 1509), dtype=torch.int64, device=dev)
        cls = self.classification()
        depth = _depth(self.rec[0])
        _lib.check(L.uvghip_alf_stats_compact_batch(depth, _dev(self.org[0]), self.org[0].stride(0), _dev(self.rec[0]), self.rec[0].stride(0), self.W, self.H, 0, _dev(self.rects_y), n,
                                                    _dev(cls), cls.stride(0), _dev(scratch["rec"]), _dev(scratch["present"]), _stream()), "uvghip_alf_stats_compact_batch")
        _lib.check(L.uvghip_alf_cov_reduce(_dev(scratch["rec"]), _dev(scratch["present"]), n, 0, _dev(sums), _stream()), "uvghip_alf_cov_reduce")
        return sums

    def chroma(self, c):
        """Per CTU the chroma covariance of plane c = 1, 2."""
        return alf_stats_batch(self.org[c], self.rec[c], self.rects_c, is_chroma=True, pic_w=self.W // 2, pic_h=self.H // 2)

    def cc(self, c):
        """Per CTU the CC-ALF covariance of plane c = 1, 2 (get_blk_stats_cc_alf, alf.c:2613)."""
        return cc_alf_stats_batch(self.org[c], self.rec[c], self.rec[0], self.rects_c)


class LowDelayLoop:
    """The per-picture loop of the encoder's CTU workers for n_seq independent low-delay sequences of the same shape (BASELINE configs[2]),
    picture after picture on the device: an I picture of every sequence through uvghip_loop_plan_* (ClosedLoop), a P / B picture of every
    sequence through uvghip_loop_pb_run with the device's own earlier output pictures and motion tables as references.  The frame-level
    bookkeeping stays with the caller (the reference's encoder_state / GOP code): `frames` is one dict per picture in coding order with
    slice_type (2 I, 1 P, 0 B), poc, qp, lam, lam_sqrt, c_lam, cw_u, cw_v, frame_qp, n_refs, ref_pocs[16], l_size[2], lists[2][16].
    After run(): out[f][s] = (y, u, v) output pictures, rows[f] / row_bytes[f] = the slice data ([n_seq, rows, row_cap] / [n_seq, rows]).
    The same loop serves a random-access GOP (--gop 16: pictures out of display order, references in the future): `frames` and `src` are in
    CODING order there, and run(in_flight=k) issues every picture as soon as the pictures it references are done -- pictures of the same
    temporal layer, and of neighbouring GOPs, run side by side on k streams (deps[f]: the coded pictures f reads)."""

    def __init__(self, W, H, depth, n_seq, frames, src, sao_type=3, tmvp=1, max_merge=6, merge_level=2, bipred=1, fme_level=4, early_skip=1, by_level=False, rd=0, inflight_margin=0,
                 inflight=False, intra_in_flight=False, intra_grid=None):
        """intra_in_flight (with inflight): the I pictures are part of the flight too -- their search runs as `intra_grid` persistent workgroups on a
        second stream BESIDE the in-flight launch, which filters them CTU by CTU as they are searched (uvghip_loop_pb_run_inflight_ext), so the
        P / B pictures behind an I picture follow it four diagonals behind instead of waiting for the whole picture.  intra_grid: workgroups of
        an I group's launch (None: one per CTU its pictures' wavefronts can have in progress, min(wc, hc) per picture -- measured best: more only wait
        and take CUs from the flight, whose workgroups need whole ones).
        inflight: the encoder's --owf schedule (encoderstate.c:1060-1116): ALL P / B pictures go through ONE uvghip_loop_pb_run_inflight -- a
        picture's CTU (x, y) starts when CTU (x + 2, y + 1) of the pictures it reads is final, the in-loop filters run per CTU inside the
        search kernel -- after the I pictures (which depend on nothing).  Needs inflight_margin = 11 (9 without SAO): the vector restriction
        of an encoder run with --owf != 0, whose bitstream this then is.
        inflight_margin: uvghip_ctu_pb_picture_t.inflight_margin (0: every vector inside a reference picture is legal; 11: the search of an
        encoder with frames in flight and SAO on, cfg.owf != 0).
        rd: cfg.rdo, 0 (--preset medium) or 1 (--preset slow: a P / B CU never skips its intra search on a low inter cost).
        by_level: the P / B pictures that sit at the same depth of the reference DAG (deps) go through ONE uvghip_loop_pb_run together --
        the pictures of one temporal layer of a random-access GOP, and of neighbouring GOPs, share a launch (their wavefronts interleave in the
        search kernel); a low-delay sequence has one picture per level and is unchanged."""
        self.W, self.H, self.depth, self.n_seq, self.frames, self.sao_type = W, H, depth, n_seq, frames, sao_type
        wc, hc = (W + 63) // 64, (H + 63) // 64
        ctus, n4 = wc * hc, hc * 16 * wc * 16
        self.hc = hc
        tdt = torch.uint8 if depth == 8 else torch.uint16
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
        self.out, self.mot, self.steps, self.keep = [], [], [], []
        self.deps, self.level = reference_dag(frames)
        by_poc, coded_as = {}, {}
        self.ref_frame = [[-1] * 16 for _ in frames]          # per coded picture and reference: the coded picture whose output it is
        # inflight: the I pictures depend on nothing -- those of one QP / lambda go through ONE all-intra loop, ahead of everything else
        igroup = {}
        if inflight:
            for f, fs in enumerate(frames):
                if fs["slice_type"] == 2:
                    igroup.setdefault((fs["qp"], fs["lam"]), []).append(f)
            for key, fl in igroup.items():
                loop = ClosedLoop(ctu_params(W, H, key[0], lam=key[1]), [tuple(src[s][f]) for f in fl for s in range(n_seq)], sao_type=sao_type)
                igroup[key] = (fl, loop, [z((hc * 16, wc * 16, 8), torch.int32) for _ in range(n_seq * len(fl))])
        for f, fs in enumerate(frames):
            srcs = [tuple(src[s][f]) for s in range(n_seq)]
            if fs["slice_type"] != 2:
                for i in range(fs["n_refs"]):
                    self.ref_frame[f][i] = coded_as[fs["ref_pocs"][i]]
            if fs["slice_type"] == 2 and inflight:
                fl, loop, gm = igroup[(fs["qp"], fs["lam"])]
                j = fl.index(f)
                outs, mots = loop.out[j * n_seq:(j + 1) * n_seq], gm[j * n_seq:(j + 1) * n_seq]
                self.steps.append(("I", loop, gm, fl))
            elif fs["slice_type"] == 2:
                loop = ClosedLoop(ctu_params(W, H, fs["qp"], lam=fs["lam"]), srcs, sao_type=sao_type)
                outs = loop.out
                mots = [z((hc * 16, wc * 16, 8), torch.int32) for _ in range(n_seq)]
                self.steps.append(("I", loop, mots, [f]))
            else:
                arr = (_lib.LoopPbPicture * n_seq)()
                outs, mots, bufs = [], [], []
                for s in range(n_seq):
                    q = arr[s]
                    p = q.search
                    w = 2.0 ** 0
                    p.params = _lib.CtuParams(W, H, fs["qp"], fs["qp"], 1, 4, 1, 1, 2, rd, fs["lam"], fs["lam_sqrt"], fs["c_lam"], fs["cw_u"], fs["cw_v"], fs["lam"])
                    t = dict(rec=[z((H >> c, W >> c), tdt) for c in (0, 1, 1)], out=tuple(z((H >> c, W >> c), tdt) for c in (0, 1, 1)),
                             scu=z(n4 * 32, torch.uint8), i4=z(n4 * 8, torch.uint8), mot=z((hc * 16, wc * 16, 8), torch.int32), co=z(ctus * 6144, torch.int16),
                             mo=z(ctus * 3 * 257, torch.int32), mi=z(ctus * 3 * 18, torch.int32))
                    c = p.pic
                    c.src_y, c.src_u, c.src_v = (_dev(a) for a in srcs[s])
                    c.rec_y, c.rec_u, c.rec_v = (_dev(a) for a in t["rec"])
                    c.src_stride, c.src_stride_c, c.rec_stride, c.rec_stride_c = srcs[s][0].stride(0), srcs[s][1].stride(0), W, W // 2
                    c.cu, c.cu_stride, c.coeff, c.models = _dev(t["scu"]), wc * 16, _dev(t["co"]), _dev(t["mo"])
                    p.slice_type, p.poc, p.n_refs, p.frame_qp = fs["slice_type"], fs["poc"], fs["n_refs"], fs["frame_qp"]
                    p.tmvp, p.max_merge, p.merge_level, p.bipred, p.fme_level, p.early_skip, p.depth_inter_min, p.depth_inter_max = tmvp, max_merge, merge_level, bipred, fme_level, early_skip, 0, 3
                    for i in range(16):
                        p.ref_pocs[i], p.l[0][i], p.l[1][i] = fs["ref_pocs"][i], fs["lists"][0][i], fs["lists"][1][i]
                    p.l_size[0], p.l_size[1] = fs["l_size"]
                    p.ref_stride, p.ref_stride_c, p.ref_motion_stride = W, W // 2, wc * 16
                    p.inflight_margin = inflight_margin
                    for i in range(fs["n_refs"]):
                        planes, rm = by_poc[(s, fs["ref_pocs"][i])]
                        p.ref_y[i], p.ref_u[i], p.ref_v[i], p.ref_motion[i] = _dev(planes[0]), _dev(planes[1]), _dev(planes[2]), _dev(rm)
                    p.inter4, p.models_inter, p.trees, p.motion_out = _dev(t["i4"]), _dev(t["mi"]), None, _dev(t["mot"])
                    q.out_y, q.out_u, q.out_v = (_dev(a) for a in t["out"])
                    q.out_stride, q.out_stride_c = W, W // 2
                    outs.append(t["out"]); mots.append(t["mot"]); bufs.append(t)
                L = _lib.init(torch.cuda.current_device())
                ws = None if inflight else z(L.uvghip_loop_pb_workspace_bytes(depth, n_seq, W, H), torch.uint8)
                self.steps.append(("PB", arr, ws, bufs))
            for s in range(n_seq):
                by_poc[(s, fs["poc"])] = (outs[s], mots[s])
            coded_as[fs["poc"]] = f
            self.out.append(outs); self.mot.append(mots)
        self.L = _lib.init(torch.cuda.current_device())
        self.rows, self.row_bytes = [None] * len(frames), [None] * len(frames)
        # order[i] = (coded pictures of the step, step): per frame, or per level of the DAG with the P / B pictures of a level merged
        self.order = [([f], st) for f, st in enumerate(self.steps)]
        if by_level:
            self.order = []
            for lv in range(1 + max(self.level)):
                fr = [f for f in range(len(frames)) if self.level[f] == lv]
                self.order += [([f], self.steps[f]) for f in fr if self.steps[f][0] == "I"]
                pb = sorted([f for f in fr if self.steps[f][0] == "PB"], key=lambda f: (frames[f]["qp"], frames[f]["slice_type"], f))      # (runs of equal QP / type share the SAO decision and the coder)
                if len(pb) == 1:
                    self.order.append((pb, self.steps[pb[0]]))
                elif pb:
                    arr = (_lib.LoopPbPicture * (n_seq * len(pb)))()
                    for j, f in enumerate(pb):
                        ctypes.memmove(ctypes.addressof(arr) + j * n_seq * ctypes.sizeof(_lib.LoopPbPicture), ctypes.addressof(self.steps[f][1]), n_seq * ctypes.sizeof(_lib.LoopPbPicture))
                    ws = z(self.L.uvghip_loop_pb_workspace_bytes(depth, n_seq * len(pb), W, H), torch.uint8)
                    self.order.append((pb, ("PB", arr, ws, None)))
            for f in range(len(frames)):          # (the per-frame workspaces are not needed)
                if self.steps[f][0] == "PB" and not any(st is self.steps[f] for _, st in self.order):
                    self.steps[f] = ("PB", self.steps[f][1], None, self.steps[f][3])
        self.intra_in_flight = bool(inflight and intra_in_flight)
        if os.environ.get("UVGHIP_INTRA_GRID"):          # (development)
            intra_grid = int(os.environ["UVGHIP_INTRA_GRID"])
        if self.intra_in_flight:
            # an intra picture as a reference: type 1 inside the picture, no vectors -- known before anything runs
            for fl, loop, gm in igroup.values():
                grid = int(intra_grid) if intra_grid else min(wc, hc) * loop.n
                loop.search_grid = min(grid, loop.n * ctus)
                if loop.n * ctus > grid:
                    _lib.check(self.L.uvghip_loop_plan_set_search_grid(loop.loop, grid), "uvghip_loop_plan_set_search_grid")
                for m in gm:
                    m.zero_()
                    m[:H // 4, :W // 4, 0] = 1
                    m[:, :, 6:8] = -1
            pb = [f for f in range(len(frames)) if self.steps[f][0] == "PB"]
            ent = []                                   # the flight's pictures: (kind, coded picture, sequence) -- the I pictures first
            for fl, loop, gm in igroup.values():
                for j, f in enumerate(fl):
                    ent += [("I", f, s_, loop, j * n_seq + s_) for s_ in range(n_seq)]
            n_ext = len(ent)
            ent += [("PB", f, s_, None, 0) for f in pb for s_ in range(n_seq)]
            at = {(e[1], e[2]): i for i, e in enumerate(ent)}
            n = len(ent)
            arr = (_lib.LoopPbPicture * n)()
            ext = (_lib.InflightExternal * n)()
            ric = np.full((n, 16), -1, np.int32)
            a_, b_ = ctypes.c_void_p(), ctypes.c_void_p()
            for i, (kind, f, s_, loop, idx) in enumerate(ent):
                if kind == "I":
                    q = arr[i]
                    q.search.params = loop.P
                    q.search.pic = loop.pics[idx]
                    q.search.slice_type = 2
                    o = loop.out[idx]
                    q.out_y, q.out_u, q.out_v = (_dev(a) for a in o)
                    q.out_stride, q.out_stride_c = o[0].stride(0), o[1].stride(0)
                    _lib.check(self.L.uvghip_loop_plan_results(loop.loop, ctypes.byref(a_), ctypes.byref(b_)), "uvghip_loop_plan_results")
                    ext[i].searched_flags = self.L.uvghip_loop_plan_searched_flags(loop.loop) + idx * ctus * 4
                    ext[i].sao_info, ext[i].sao_models = a_.value + idx * ctus * 34 * 4, b_.value + idx * ctus * 6 * 2
                else:
                    ctypes.memmove(ctypes.addressof(arr) + i * ctypes.sizeof(_lib.LoopPbPicture), ctypes.addressof(self.steps[f][1]) + s_ * ctypes.sizeof(_lib.LoopPbPicture),
                                   ctypes.sizeof(_lib.LoopPbPicture))
                    for k in range(frames[f]["n_refs"]):
                        g = self.ref_frame[f][k]
                        if (g, s_) in at:
                            ric[i, k] = at[(g, s_)]
            ws = z(self.L.uvghip_loop_pb_inflight_workspace_bytes(depth, n, W, H), torch.uint8)
            grid_sum = sum(loop.search_grid for _, loop, _ in igroup.values())
            self.order = [(list(range(len(frames))), ("FLIGHT_EXT", arr, ws, (np.ascontiguousarray(ric), ext, ent, n_ext, grid_sum, list(igroup.values()))))]
            for f in pb:
                self.steps[f] = ("PB", self.steps[f][1], None, self.steps[f][3])
        elif inflight:
            pb = [f for f in range(len(frames)) if self.steps[f][0] == "PB"]
            self.order = [(fl, ("I", loop, gm, fl)) for fl, loop, gm in igroup.values()]
            if pb:
                at = {f: j for j, f in enumerate(pb)}
                n = n_seq * len(pb)
                arr = (_lib.LoopPbPicture * n)()
                ric = np.full((n, 16), -1, np.int32)
                for j, f in enumerate(pb):
                    ctypes.memmove(ctypes.addressof(arr) + j * n_seq * ctypes.sizeof(_lib.LoopPbPicture), ctypes.addressof(self.steps[f][1]), n_seq * ctypes.sizeof(_lib.LoopPbPicture))
                    for i in range(frames[f]["n_refs"]):
                        g = self.ref_frame[f][i]
                        if g in at:
                            for s_ in range(n_seq):
                                ric[j * n_seq + s_, i] = at[g] * n_seq + s_
                ws = z(self.L.uvghip_loop_pb_inflight_workspace_bytes(depth, n, W, H), torch.uint8)
                self.order.append((pb, ("FLIGHT", arr, ws, np.ascontiguousarray(ric))))
            for f in pb:          # (the per-frame workspaces are not needed)
                self.steps[f] = ("PB", self.steps[f][1], None, self.steps[f][3])

    def run(self, stream=None, in_flight=1):
        """Enqueue every picture of every sequence (the P / B groups wait for the stream where the coder's tables are uploaded).
        in_flight > 1: coded picture f goes to stream f % in_flight of the loop's own streams and waits only for the pictures it references
        (and for the caller's stream at the start; the caller's stream waits for all of them at the end)."""
        st = _stream() if stream is None else stream
        wc = (self.W + 63) // 64
        if in_flight > 1:
            if not hasattr(self, "_streams") or len(self._streams) < in_flight:
                self._streams = [torch.cuda.Stream() for _ in range(in_flight)]
            start = torch.cuda.Event()
            start.record(torch.cuda.ExternalStream(st))
            for t in self._streams[:in_flight]:
                t.wait_event(start)
            done = {}
        if self.intra_in_flight:
            return self._run_intra_in_flight(st)
        for k, (fr, step) in enumerate(self.order):
            f = fr[0]
            if in_flight > 1:
                own = self._streams[k % in_flight]
                for d in sorted({d for g in fr for d in self.deps[g]}):
                    own.wait_event(done[d])
                st = own.cuda_stream
            if step[0] == "I":
                _, loop, mots, fl = step
                loop.run(st)
                with torch.cuda.stream(torch.cuda.ExternalStream(st)):      # (on the SAME stream as the search that writes loop.cu and the P / B search that reads mots)
                    for s in range(len(mots)):       # an intra picture as a reference: its units' type, no vectors
                        mots[s][:, :, 0] = loop.cu[s][:, :, 2].to(torch.int32)
                        mots[s][:, :, 6:8] = -1
                rows, nb = loop.slice_data()
                for j, g in enumerate(fl):
                    self.rows[g], self.row_bytes[g] = rows[j * self.n_seq:(j + 1) * self.n_seq], nb[j * self.n_seq:(j + 1) * self.n_seq]
            else:
                kind, arr, ws, ric = step
                n = self.n_seq * len(fr)
                c, d, cap, nr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
                if kind == "FLIGHT":
                    _lib.check(self.L.uvghip_loop_pb_run_inflight(self.depth, ctypes.byref(arr), n, self.sao_type, ric.ctypes.data, _dev(ws), st), "uvghip_loop_pb_run_inflight")
                    _lib.check(self.L.uvghip_loop_pb_inflight_results(self.depth, n, self.W, self.H, _dev(ws), None, None, ctypes.byref(c), ctypes.byref(d), ctypes.byref(cap),
                                                                      ctypes.byref(nr)), "uvghip_loop_pb_inflight_results")
                else:
                    _lib.check(self.L.uvghip_loop_pb_run(self.depth, ctypes.byref(arr), n, self.sao_type, _dev(ws), st), "uvghip_loop_pb_run")
                    _lib.check(self.L.uvghip_loop_pb_results(self.depth, n, self.W, self.H, _dev(ws), None, None, ctypes.byref(c), ctypes.byref(d), ctypes.byref(cap),
                                                             ctypes.byref(nr)), "uvghip_loop_pb_results")
                base = ws.data_ptr()
                rows = ws[c.value - base:c.value - base + n * nr.value * cap.value].view(n, nr.value, cap.value)
                row_bytes = ws[d.value - base:d.value - base + n * nr.value * 4].view(torch.int32).view(n, nr.value)
                for j, g in enumerate(fr):
                    self.rows[g], self.row_bytes[g] = rows[j * self.n_seq:(j + 1) * self.n_seq], row_bytes[j * self.n_seq:(j + 1) * self.n_seq]
            if in_flight > 1:
                ev = torch.cuda.Event()
                ev.record(own)
                for g in fr:
                    done[g] = ev
        if in_flight > 1:
            caller = torch.cuda.ExternalStream(_stream() if stream is None else stream)
            for t in self._streams[:in_flight]:
                e = torch.cuda.Event()
                e.record(t)
                caller.wait_event(e)

    def _run_intra_in_flight(self, st):
        """The I pictures' searches on stream A, the flight (their filters + every P / B picture) on stream B, both behind the caller's stream;
        the I pictures' slice data behind the flight; the caller's stream waits for both."""
        _, (kind, arr, ws, (ric, ext, ent, n_ext, grid_sum, groups)) = self.order[0]
        if not hasattr(self, "_ab"):
            self._ab = (torch.cuda.Stream(), torch.cuda.Stream())
        A, B = self._ab
        caller = torch.cuda.ExternalStream(st)
        start = torch.cuda.Event()
        start.record(caller)
        A.wait_event(start); B.wait_event(start)
        n = len(ent)
        ctus = ((self.W + 63) // 64) * ((self.H + 63) // 64)
        self.L.uvghip_loop_pb_inflight_final_flags.restype = ctypes.c_void_p
        final = self.L.uvghip_loop_pb_inflight_final_flags(self.depth, n, self.W, self.H, _dev(ws))
        off = final - ws.data_ptr()
        with torch.cuda.stream(A):           # the I pictures' "final" flags: zero before their coder (behind the search, on A) can look at them
            ws[off:off + n_ext * ctus * 4].zero_()
        for fl, loop, gm in groups:          # flags back to zero before the flight's kernel can look at them
            _lib.check(self.L.uvghip_loop_plan_search_reset(loop.loop, A.cuda_stream), "uvghip_loop_plan_search_reset")
        cleared = torch.cuda.Event()
        cleared.record(A)
        B.wait_event(cleared)
        for fl, loop, gm in groups:
            _lib.check(self.L.uvghip_loop_plan_search_launch(loop.loop, A.cuda_stream), "uvghip_loop_plan_search_launch")
        _lib.check(self.L.uvghip_loop_pb_run_inflight_ext(self.depth, ctypes.byref(arr), n, self.sao_type, ric.ctypes.data, ctypes.byref(ext), grid_sum, _dev(ws), B.cuda_stream),
                   "uvghip_loop_pb_run_inflight_ext")
        # the I pictures' slice data: behind their search on A, BESIDE the flight -- a row waits, CTU by CTU, for the flight's filter stage
        # (uvghip_loop_plan_run_coder_behind); the I pictures are the first entries of the call, group after group
        first = 0
        for fl, loop, gm in groups:
            _lib.check(self.L.uvghip_loop_plan_run_coder_behind(loop.loop, ctypes.c_void_p(final + first * ctus * 4), A.cuda_stream), "uvghip_loop_plan_run_coder_behind")
            first += loop.n
            rows, nb = loop.slice_data()
            for j, g in enumerate(fl):
                self.rows[g], self.row_bytes[g] = rows[j * self.n_seq:(j + 1) * self.n_seq], nb[j * self.n_seq:(j + 1) * self.n_seq]
        coded = torch.cuda.Event()
        coded.record(A)
        B.wait_event(coded)
        c, d, cap, nr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self.L.uvghip_loop_pb_inflight_results(self.depth, n, self.W, self.H, _dev(ws), None, None, ctypes.byref(c), ctypes.byref(d), ctypes.byref(cap), ctypes.byref(nr)),
                   "uvghip_loop_pb_inflight_results")
        base = ws.data_ptr()
        rows = ws[c.value - base:c.value - base + n * nr.value * cap.value].view(n, nr.value, cap.value)
        row_bytes = ws[d.value - base:d.value - base + n * nr.value * 4].view(torch.int32).view(n, nr.value)
        pb = sorted({e[1] for e in ent[n_ext:]})
        for j, g in enumerate(pb):
            i0 = n_ext + j * self.n_seq
            self.rows[g], self.row_bytes[g] = rows[i0:i0 + self.n_seq], row_bytes[i0:i0 + self.n_seq]
        done = torch.cuda.Event()
        done.record(B)
        caller.wait_event(done)
