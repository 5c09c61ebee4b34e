// uvghip_loop_pb_*: one call per group of independent P / B pictures for the whole per-picture loop of the encoder's CTU worker
// (src/encoderstate.c:808-976): the closed-loop CTU search (uvghip_ctu_search_pb), then per picture the in-loop filters on the
// reference's schedule -- every CTU deblocked by its own edges only (what uvg_sao_search_lcu reads), SAO statistics and decisions with
// the slice type's models, deblocking of the reconstruction, SAO apply into the output picture (the next pictures' reference) -- and
// the slice data (uvghip_encode_slice_rows_pb).  Host code only: it strings the library's own entry points together on the caller's
// stream inside one workspace.
#include "uvghip_common.h"
#include <vector>
#include <cstring>

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct layout_t { size_t search, snap, rects_y, rects_c, edge[3], band[3], decide, info, models, params[3], coder, row_bytes, rows, total; int row_cap; };

layout_t layout_of(int bitdepth, int n, int w, int h)
{
  const size_t ctus = (size_t)((w + 63) / 64) * ((h + 63) / 64), b = bitdepth == 8 ? 1 : 2;
  layout_t L;
  size_t at = 0;
  auto take = [&](size_t bytes) { const size_t o = at; at = align_up(at + bytes, 256); return o; };
  L.search = take(uvghip_ctu_search_pb_workspace_bytes(n, w, h));
  L.snap = take((size_t)n * ((size_t)w * h * 3 / 2) * b);
  L.rects_y = take(ctus * sizeof(uvghip_rect_t));
  L.rects_c = take(ctus * sizeof(uvghip_rect_t));
  for (int c = 0; c < 3; ++c) { L.edge[c] = take((size_t)n * ctus * 40 * 4); L.band[c] = take((size_t)n * ctus * 64 * 4); }
  L.decide = take(uvghip_sao_decide_workspace_bytes(n, w, h));
  L.info = take((size_t)n * ctus * 34 * 4);
  L.models = take((size_t)n * ctus * 6 * 2);
  for (int c = 0; c < 3; ++c) L.params[c] = take((size_t)n * ctus * sizeof(uvghip_sao_param_t));
  const size_t hc = (size_t)((h + 63) / 64);
  L.row_cap = 3 * 64 * w * (int)b;
  L.coder = take(uvghip_slice_rows_pb_workspace_bytes(n));
  L.row_bytes = take((size_t)n * hc * 4);
  L.rows = take((size_t)n * hc * L.row_cap);
  L.total = at;
  return L;
}

}  // namespace

extern "C" size_t uvghip_loop_pb_workspace_bytes(int bitdepth, int n_pictures, int pic_w, int pic_h)
{
  if ((bitdepth != 8 && bitdepth != 10) || n_pictures <= 0 || pic_w <= 0 || pic_h <= 0) return 0;
  return layout_of(bitdepth, n_pictures, pic_w, pic_h).total;
}

extern "C" int uvghip_loop_pb_results(int bitdepth, int n_pictures, int pic_w, int pic_h, void *workspace, const int32_t **sao_info, const uint16_t **sao_models,
                                      const uint8_t **rows, const int32_t **row_bytes, int *row_cap, int *n_rows)
{
  if ((bitdepth != 8 && bitdepth != 10) || n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !workspace) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const layout_t L = layout_of(bitdepth, n_pictures, pic_w, pic_h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  if (sao_info) *sao_info = reinterpret_cast<const int32_t *>(ws + L.info);
  if (sao_models) *sao_models = reinterpret_cast<const uint16_t *>(ws + L.models);
  if (rows) *rows = ws + L.rows;
  if (row_bytes) *row_bytes = reinterpret_cast<const int32_t *>(ws + L.row_bytes);
  if (row_cap) *row_cap = L.row_cap;
  if (n_rows) *n_rows = (pic_h + 63) / 64;
  return 0;
}

extern "C" int uvghip_loop_pb_run(int bitdepth, const uvghip_loop_pb_picture_t *pictures, int n_pictures, int sao_type, void *workspace, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!pictures || n_pictures <= 0 || !workspace || sao_type < 0 || sao_type > 3) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int w = pictures[0].search.params.pic_w, h = pictures[0].search.params.pic_h;
  if (w <= 0 || h <= 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  std::vector<uvghip_ctu_pb_picture_t> sp(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_loop_pb_picture_t &q = pictures[i];
    if (!q.out_y || !q.out_u || !q.out_v || q.out_stride < w || q.out_stride_c < w / 2) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_pb_run: output planes");
    // the filters run without a chroma QP table (deblock.hip: identity)
    if (q.search.params.qp_c != q.search.params.qp) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_pb_run: qp_c != qp needs a chroma QP table");
    sp[i] = q.search;
  }
  const layout_t L = layout_of(bitdepth, n_pictures, w, h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  hipStream_t st = uvghip_stream(stream);
  if (int rc = uvghip_ctu_search_pb(bitdepth, sp.data(), n_pictures, ws + L.search, stream)) return rc;
  const int wc = (w + 63) / 64, hc = (h + 63) / 64, ctus = wc * hc, cw = w / 2, ch = h / 2;
  const size_t b = bitdepth == 8 ? 1 : 2;
  uvghip_rect_t *rects_y = reinterpret_cast<uvghip_rect_t *>(ws + L.rects_y), *rects_c = reinterpret_cast<uvghip_rect_t *>(ws + L.rects_c);
  {
    std::vector<uvghip_rect_t> ry(ctus), rc(ctus);
    for (int cy = 0; cy < hc; ++cy)
      for (int cx = 0; cx < wc; ++cx) {
        const int x = cx * 64, y = cy * 64, bw = x + 64 > w ? w - x : 64, bh = y + 64 > h ? h - y : 64;
        ry[cy * wc + cx] = uvghip_rect_t{x, y, bw, bh};
        rc[cy * wc + cx] = uvghip_rect_t{x / 2, y / 2, bw / 2, bh / 2};
      }
    // (in stream order: an earlier run on this workspace may still read the tables)
    if (int e = uvghip_upload_ordered(rects_y, ry.data(), ry.size() * sizeof(uvghip_rect_t), st)) return e;
    if (int e = uvghip_upload_ordered(rects_c, rc.data(), rc.size() * sizeof(uvghip_rect_t), st)) return e;
  }
  int32_t *edge[3], *band[3];
  uvghip_sao_param_t *prm[3];
  for (int c = 0; c < 3; ++c) {
    edge[c] = reinterpret_cast<int32_t *>(ws + L.edge[c]); band[c] = reinterpret_cast<int32_t *>(ws + L.band[c]);
    prm[c] = reinterpret_cast<uvghip_sao_param_t *>(ws + L.params[c]);
  }
  int32_t *sao_info = reinterpret_cast<int32_t *>(ws + L.info), *row_bytes = reinterpret_cast<int32_t *>(ws + L.row_bytes);
  uint16_t *sao_models = reinterpret_cast<uint16_t *>(ws + L.models);
  const size_t snap_bytes = (size_t)w * h * 3 / 2 * b;
  // pictures that share QP, lambda and slice type (the same temporal position of several sequences) go through the SAO decision and
  // the coder together: runs of such pictures
  for (int i0 = 0; i0 < n_pictures;) {
    const uvghip_ctu_pb_picture_t &s0 = pictures[i0].search;
    int i1 = i0 + 1;
    while (i1 < n_pictures && pictures[i1].search.params.qp == s0.params.qp && pictures[i1].search.params.lambda == s0.params.lambda &&
           pictures[i1].search.slice_type == s0.slice_type && pictures[i1].search.frame_qp == s0.frame_qp)
      ++i1;
    const int m = i1 - i0, is_b = s0.slice_type == 0, qp = s0.params.qp;
    // The slice's context models start from frame_qp everywhere (uvg_init_contexts with state->frame->QP: the search, the coder); the SAO
    // decision below takes ONE QP for its models and its CTUs.  A picture whose CTUs' QP differs from the frame's (per-CTU QP offsets) is not
    // something this loop implements: refuse it instead of deciding SAO on other models than the coder's.
    if (s0.params.qp != s0.frame_qp) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_pb_run: params.qp differs from frame_qp");
    const size_t o0 = (size_t)i0 * ctus;
    if (sao_type) {
      for (int i = i0; i < i1; ++i) {
        const uvghip_ctu_picture_t &p = pictures[i].search.pic;
        unsigned char *sy = ws + L.snap + (size_t)i * snap_bytes, *su = sy + (size_t)w * h * b, *sv = su + (size_t)cw * ch * b;
        UVGHIP_TRY(hipMemcpy2DAsync(sy, (size_t)w * b, p.rec_y, (size_t)p.rec_stride * b, (size_t)w * b, h, hipMemcpyDeviceToDevice, st));
        UVGHIP_TRY(hipMemcpy2DAsync(su, (size_t)cw * b, p.rec_u, (size_t)p.rec_stride_c * b, (size_t)cw * b, ch, hipMemcpyDeviceToDevice, st));
        UVGHIP_TRY(hipMemcpy2DAsync(sv, (size_t)cw * b, p.rec_v, (size_t)p.rec_stride_c * b, (size_t)cw * b, ch, hipMemcpyDeviceToDevice, st));
        if (int rc = uvghip_deblock_frame_sao_snapshot(bitdepth, sy, w, su, sv, cw, w, h, p.cu, p.cu_stride, 0, 0, is_b, qp, nullptr, stream)) return rc;
        const size_t o = (size_t)i * ctus;
        if (int rc = uvghip_sao_stats_batch(bitdepth, p.src_y, p.src_stride, sy, w, rects_y, ctus, edge[0] + o * 40, band[0] + o * 64, stream)) return rc;
        if (int rc = uvghip_sao_stats_batch(bitdepth, p.src_u, p.src_stride_c, su, cw, rects_c, ctus, edge[1] + o * 40, band[1] + o * 64, stream)) return rc;
        if (int rc = uvghip_sao_stats_batch(bitdepth, p.src_v, p.src_stride_c, sv, cw, rects_c, ctus, edge[2] + o * 40, band[2] + o * 64, stream)) return rc;
      }
      if (int rc = uvghip_sao_decide_pictures_slice(bitdepth, m, w, h, qp, s0.params.lambda, sao_type, s0.slice_type, edge[0] + o0 * 40, band[0] + o0 * 64,
                                                    edge[1] + o0 * 40, band[1] + o0 * 64, edge[2] + o0 * 40, band[2] + o0 * 64,
                                                    ws + L.decide, sao_info + o0 * 34, sao_models + o0 * 6, prm[0] + o0, prm[1] + o0, prm[2] + o0, stream))
        return rc;
    }
    std::vector<uvghip_ctu_picture_t> cp(m);
    std::vector<uvghip_slice_pb_t> sl(m);
    for (int i = i0; i < i1; ++i) {
      const uvghip_loop_pb_picture_t &q = pictures[i];
      const uvghip_ctu_pb_picture_t &s = q.search;
      const uvghip_ctu_picture_t &p = s.pic;
      const size_t o = (size_t)i * ctus;
      if (int rc = uvghip_deblock_frame(bitdepth, p.rec_y, p.rec_stride, p.rec_u, p.rec_v, p.rec_stride_c, w, h, p.cu, p.cu_stride, 0, 0, is_b, qp, nullptr, stream)) return rc;
      if (sao_type) {
        if (int rc = uvghip_sao_apply_batch(bitdepth, p.rec_y, p.rec_stride, q.out_y, q.out_stride, w, h, rects_y, prm[0] + o, ctus, stream)) return rc;
        if (int rc = uvghip_sao_apply_batch(bitdepth, p.rec_u, p.rec_stride_c, q.out_u, q.out_stride_c, cw, ch, rects_c, prm[1] + o, ctus, stream)) return rc;
        if (int rc = uvghip_sao_apply_batch(bitdepth, p.rec_v, p.rec_stride_c, q.out_v, q.out_stride_c, cw, ch, rects_c, prm[2] + o, ctus, stream)) return rc;
      } else {
        UVGHIP_TRY(hipMemcpy2DAsync(q.out_y, (size_t)q.out_stride * b, p.rec_y, (size_t)p.rec_stride * b, (size_t)w * b, h, hipMemcpyDeviceToDevice, st));
        UVGHIP_TRY(hipMemcpy2DAsync(q.out_u, (size_t)q.out_stride_c * b, p.rec_u, (size_t)p.rec_stride_c * b, (size_t)cw * b, ch, hipMemcpyDeviceToDevice, st));
        UVGHIP_TRY(hipMemcpy2DAsync(q.out_v, (size_t)q.out_stride_c * b, p.rec_v, (size_t)p.rec_stride_c * b, (size_t)cw * b, ch, hipMemcpyDeviceToDevice, st));
      }
      cp[i - i0] = p;
      uvghip_slice_pb_t &d = sl[i - i0];
      d.slice_type = s.slice_type; d.poc = s.poc; d.n_refs = s.n_refs;
      for (int k = 0; k < 16; ++k) { d.ref_pocs[k] = s.ref_pocs[k]; d.l[0][k] = s.l[0][k]; d.l[1][k] = s.l[1][k]; }
      d.l_size[0] = s.l_size[0]; d.l_size[1] = s.l_size[1];
      d.tmvp = s.tmvp; d.max_merge = s.max_merge; d.merge_level = s.merge_level; d.frame_qp = s.frame_qp;
      d.col = s.ref_motion[s.l[0][0]]; d.col_stride = s.ref_motion_stride; d.reserved = 0;
      d.inter4 = s.inter4; d.models_inter = s.models_inter;
    }
    // the slice data of the run's pictures: the search's hand-over and the SAO decisions through the arithmetic coder (its tables are
    // uploaded in stream order: a later run's upload comes after the earlier run's coder on the stream)
    if (int rc = uvghip_encode_slice_rows_pb(bitdepth, &s0.params, cp.data(), sl.data(), m, sao_type ? sao_info + o0 * 34 : nullptr, sao_type ? sao_models + o0 * 6 : nullptr,
                                             ws + L.coder, ws + L.rows + (size_t)i0 * hc * L.row_cap, L.row_cap, row_bytes + (size_t)i0 * hc, stream))
      return rc;
    i0 = i1;
  }
  return 0;
}

// ---- pictures in flight behind their references: one call = one persistent launch for the whole DAG (uvghip_ctu_search_pb_inflight: the
// search with the per-CTU filters inside), then ONE coder launch over all pictures -- their QPs, lambdas and slice types may differ ----
namespace {
struct flight_t { size_t search, dbk, info, models, coder, row_bytes, rows, total; int row_cap; };
flight_t flight_of(int bitdepth, int n, int w, int h)
{
  const size_t ctus = (size_t)((w + 63) / 64) * ((h + 63) / 64), b = bitdepth == 8 ? 1 : 2, hc = (size_t)((h + 63) / 64);
  flight_t L;
  size_t at = 0;
  auto take = [&](size_t bytes) { const size_t o = at; at = align_up(at + bytes, 256); return o; };
  L.search = take(uvghip_ctu_search_pb_inflight_workspace_bytes(n, w, h));
  L.dbk = take((size_t)n * ((size_t)w * h * 3 / 2) * b);
  L.info = take((size_t)n * ctus * 34 * 4);
  L.models = take((size_t)n * ctus * 6 * 2);
  L.row_cap = 3 * 64 * w * (int)b;
  L.coder = take(uvghip_slice_rows_pb_workspace_bytes(n));
  L.row_bytes = take((size_t)n * hc * 4);
  L.rows = take((size_t)n * hc * L.row_cap);
  L.total = at;
  return L;
}
}  // namespace

extern "C" size_t uvghip_loop_pb_inflight_workspace_bytes(int bitdepth, int n_pictures, int pic_w, int pic_h)
{
  if ((bitdepth != 8 && bitdepth != 10) || n_pictures <= 0 || pic_w <= 0 || pic_h <= 0) return 0;
  return flight_of(bitdepth, n_pictures, pic_w, pic_h).total;
}

extern "C" int uvghip_loop_pb_inflight_results(int bitdepth, int n_pictures, int pic_w, int pic_h, void *workspace, const int32_t **sao_info, const uint16_t **sao_models,
                                               const uint8_t **rows, const int32_t **row_bytes, int *row_cap, int *n_rows)
{
  if ((bitdepth != 8 && bitdepth != 10) || n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !workspace) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const flight_t L = flight_of(bitdepth, n_pictures, pic_w, pic_h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  if (sao_info) *sao_info = reinterpret_cast<const int32_t *>(ws + L.info);
  if (sao_models) *sao_models = reinterpret_cast<const uint16_t *>(ws + L.models);
  if (rows) *rows = ws + L.rows;
  if (row_bytes) *row_bytes = reinterpret_cast<const int32_t *>(ws + L.row_bytes);
  if (row_cap) *row_cap = L.row_cap;
  if (n_rows) *n_rows = (pic_h + 63) / 64;
  return 0;
}

extern "C" const int32_t *uvghip_loop_pb_inflight_final_flags(int bitdepth, int n_pictures, int pic_w, int pic_h, const void *workspace)
{
  if ((bitdepth != 8 && bitdepth != 10) || n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !workspace) return nullptr;
  const flight_t L = flight_of(bitdepth, n_pictures, pic_w, pic_h);
  return uvghip_ctu_search_pb_inflight_final_flags(n_pictures, pic_w, pic_h, static_cast<const unsigned char *>(workspace) + L.search);
}

extern "C" int uvghip_loop_pb_run_inflight(int bitdepth, const uvghip_loop_pb_picture_t *pictures, int n_pictures, int sao_type, const int32_t *ref_in_call, void *workspace,
                                           void *stream)
{
  return uvghip_loop_pb_run_inflight_ext(bitdepth, pictures, n_pictures, sao_type, ref_in_call, nullptr, 0, workspace, stream);
}

// ... with pictures whose SEARCH runs in another launch beside this one (ext[i].searched_flags != NULL: an I picture of the all-intra plan,
// uvghip_loop_plan_search_launch on another stream): this call filters them CTU by CTU as that launch finishes their CTUs, so that the
// P / B pictures behind them are in flight behind an I picture as behind any other.  Their SAO decisions go to ext[i].sao_info / sao_models
// (the all-intra plan's arrays: its coder, uvghip_loop_plan_run_coder, reads them); they get no slice data here.
extern "C" int uvghip_loop_pb_run_inflight_ext(int bitdepth, const uvghip_loop_pb_picture_t *pictures, int n_pictures, int sao_type, const int32_t *ref_in_call,
                                               const uvghip_inflight_external_t *ext, int other_workgroups, void *workspace, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!pictures || n_pictures <= 0 || !workspace || !ref_in_call || sao_type < 0 || sao_type > 3) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int w = pictures[0].search.params.pic_w, h = pictures[0].search.params.pic_h;
  if (w <= 0 || h <= 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const flight_t L = flight_of(bitdepth, n_pictures, w, h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  const int wc = (w + 63) / 64, hc = (h + 63) / 64, ctus = wc * hc;
  const size_t b = bitdepth == 8 ? 1 : 2, plane = (size_t)w * h * b, planes = plane * 3 / 2;
  int32_t *sao_info = reinterpret_cast<int32_t *>(ws + L.info), *row_bytes = reinterpret_cast<int32_t *>(ws + L.row_bytes);
  uint16_t *sao_models = reinterpret_cast<uint16_t *>(ws + L.models);
  std::vector<uvghip_ctu_pb_picture_t> sp(n_pictures);
  std::vector<uvghip_pb_filter_t> fl(n_pictures);
  std::vector<uvghip_ctu_picture_t> cp(n_pictures);
  std::vector<uvghip_slice_pb_t> sl(n_pictures);
  std::vector<const int32_t *> flags(n_pictures, nullptr);
  std::vector<int> coded;          // the pictures this call codes (all but the externally searched ones)
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_loop_pb_picture_t &q = pictures[i];
    const uvghip_ctu_pb_picture_t &s = q.search;
    if (!q.out_y || !q.out_u || !q.out_v || q.out_stride < w || q.out_stride_c < w / 2) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_pb_run_inflight: output planes");
    if (ext && ext[i].searched_flags) flags[i] = ext[i].searched_flags; else coded.push_back(i);
    sp[i] = s;
    uvghip_pb_filter_t &f = fl[i];
    unsigned char *d = ws + L.dbk + (size_t)i * planes;
    f.dbk_y = d; f.dbk_u = d + plane; f.dbk_v = d + plane + plane / 4;
    f.dbk_stride = w; f.dbk_stride_c = w / 2;
    f.out_y = q.out_y; f.out_u = q.out_u; f.out_v = q.out_v; f.out_stride = q.out_stride; f.out_stride_c = q.out_stride_c;
    f.sao_info = sao_info + (size_t)i * ctus * 34; f.sao_models = sao_models + (size_t)i * ctus * 6;
    if (flags[i] && ext[i].sao_info && ext[i].sao_models) { f.sao_info = ext[i].sao_info; f.sao_models = ext[i].sao_models; }
    f.sao_type = sao_type; f.reserved = 0;
    cp[i] = s.pic;
    uvghip_slice_pb_t &o = sl[i];
    if (flags[i]) { memset(&o, 0, sizeof o); continue; }          // (not coded here)
    o.slice_type = s.slice_type; o.poc = s.poc; o.n_refs = s.n_refs;
    for (int k = 0; k < 16; ++k) { o.ref_pocs[k] = s.ref_pocs[k]; o.l[0][k] = s.l[0][k]; o.l[1][k] = s.l[1][k]; }
    o.l_size[0] = s.l_size[0]; o.l_size[1] = s.l_size[1];
    o.tmvp = s.tmvp; o.max_merge = s.max_merge; o.merge_level = s.merge_level; o.frame_qp = s.frame_qp;
    o.col = s.ref_motion[s.l[0][0]]; o.col_stride = s.ref_motion_stride; o.reserved = 0;
    o.inter4 = s.inter4; o.models_inter = s.models_inter;
  }
  if (ext) {
    if (int rc = uvghip_ctu_search_pb_inflight_ext(bitdepth, sp.data(), fl.data(), ref_in_call, flags.data(), other_workgroups, n_pictures, ws + L.search, stream)) return rc;
  } else if (int rc = uvghip_ctu_search_pb_inflight(bitdepth, sp.data(), fl.data(), ref_in_call, n_pictures, ws + L.search, stream)) return rc;
  // the slice data of every picture in one launch (a P / B picture's models start from its own frame_qp and slice type: `params` only names the size).
  // The coded pictures are a suffix of the call in practice (I pictures first); runs of them keep their place in the results' arrays.
  for (size_t a = 0; a < coded.size();) {
    size_t b = a + 1;
    while (b < coded.size() && coded[b] == coded[b - 1] + 1) ++b;
    const int i0 = coded[a], m = (int)(b - a);
    if (int rc = uvghip_encode_slice_rows_pb(bitdepth, &pictures[i0].search.params, cp.data() + i0, sl.data() + i0, m, sao_type ? sao_info + (size_t)i0 * ctus * 34 : nullptr,
                                             sao_type ? sao_models + (size_t)i0 * ctus * 6 : nullptr, ws + L.coder, ws + L.rows + (size_t)i0 * hc * L.row_cap, L.row_cap,
                                             row_bytes + (size_t)i0 * hc, stream))
      return rc;
    a = b;
  }
  return 0;
}
