// uvghip_loop_plan_*: one call per group of all-intra pictures for the whole per-picture loop of the encoder's CTU worker
// (src/encoderstate.c:808-976 without the bitstream writer): the closed-loop CTU search, then the in-loop filters on the
// reference's own schedule -- every CTU deblocked by its own edges only (what uvg_sao_search_lcu reads), SAO statistics, the
// SAO decision of every CTU of every picture, deblocking of the reconstruction, SAO apply.  Host code only: a plan strings the
// library's own entry points together on the caller's stream and owns the tables they need.
#include "uvghip_common.h"
#include <new>
#include <vector>
#include <cstring>
#include <cstdlib>

struct uvghip_loop_plan {
  int bitdepth, n, w, h, qp, sao_type, ctus;
  double lambda;
  uvghip_ctu_plan_t *search;
  std::vector<uvghip_loop_picture_t> pics;
  // carved out of the caller's workspace
  unsigned char *snap;                    // per picture: Y, U, V of the snapshot, tightly packed
  uvghip_rect_t *rects_y, *rects_c;
  int32_t *edge[3], *band[3];
  void *decide_ws;
  int32_t *sao_info;
  uint16_t *sao_models;
  uvghip_sao_param_t *params[3];
  size_t snap_bytes;                      // of one picture
  void *coder_ws;                         // the slice coder's picture table
  uint8_t *rows;                          // the slice data: row r of picture p at rows + (p * hc + r) * row_cap
  int32_t *row_bytes;
  int row_cap, hc;
  uint32_t *sums;                         // per picture: the three plane checksums of the hash SEI (filled on demand)
  uvghip_ctu_params_t ctu_params;
  int fused;                              // the filters are ONE launch behind the search (uvghip_filter_pictures); `snap` holds the deblocked pictures
  void *filt_ws;
  int32_t *coder_ticket;                  // uvghip_loop_plan_run_overlapped: the persistent coder's row counter
  // uvghip_loop_plan_group_nals: the rows of the whole group gathered on the device and brought over in one copy (grown on demand)
  // uvghip_loop_plan_run_overlapped: the filter stage and the coder on streams of the plan's own (created on first use)
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_side[2] = {nullptr, nullptr};
  uint8_t *pack_dev = nullptr, *pack_host = nullptr;
  size_t pack_cap = 0;
  unsigned long long *pack_base = nullptr;    // device: [n + 1] byte offsets of the pictures in the packed buffer, then [n] row pitches
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct layout_t { size_t search, snap, rects_y, rects_c, edge[3], band[3], decide, info, models, params[3], coder, row_bytes, rows, sums, filt, tick, total; int row_cap; };

layout_t layout_of(int bitdepth, int n, int w, int h)
{
  const size_t ctus = (size_t)((w + 63) / 64) * ((h + 63) / 64), b = bitdepth == 8 ? 1 : 2;
  layout_t L;
  size_t at = 0;
  auto take = [&](size_t bytes) { const size_t o = at; at = align_up(at + bytes, 256); return o; };
  L.search = take(uvghip_ctu_search_workspace_bytes(n, w, h));
  L.snap = take((size_t)n * ((size_t)w * h * 3 / 2) * b);
  L.rects_y = take(ctus * sizeof(uvghip_rect_t));
  L.rects_c = take(ctus * sizeof(uvghip_rect_t));
  for (int c = 0; c < 3; ++c) { L.edge[c] = take((size_t)n * ctus * 40 * 4); L.band[c] = take((size_t)n * ctus * 64 * 4); }
  L.decide = take(uvghip_sao_decide_workspace_bytes(n, w, h));
  L.info = take((size_t)n * ctus * 34 * 4);
  L.models = take((size_t)n * ctus * 6 * 2);
  for (int c = 0; c < 3; ++c) L.params[c] = take((size_t)n * ctus * sizeof(uvghip_sao_param_t));
  const size_t hc = (size_t)((h + 63) / 64);
  L.row_cap = 3 * 64 * w * (int)b;        // twice the raw storage of a CTU row of 4:2:0 samples: no row of real content comes near
  L.coder = take(uvghip_slice_rows_workspace_bytes(n));
  L.row_bytes = take((size_t)n * hc * 4);
  L.rows = take((size_t)n * hc * L.row_cap);
  L.sums = take((size_t)n * 3 * sizeof(uint32_t));
  L.filt = take(uvghip_filter_pictures_workspace_bytes(n, w, h));
  L.tick = take(256);
  L.total = at;
  return L;
}

}  // namespace

extern "C" size_t uvghip_loop_workspace_bytes(int bitdepth, int n_pictures, int pic_w, int pic_h)
{
  if ((bitdepth != 8 && bitdepth != 10) || n_pictures <= 0 || pic_w <= 0 || pic_h <= 0) return 0;
  return layout_of(bitdepth, n_pictures, pic_w, pic_h).total;
}

extern "C" int uvghip_loop_plan_create(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_loop_picture_t *pictures, int n_pictures,
                                       int sao_type, void *workspace, uvghip_loop_plan_t **plan_out)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || !pictures || n_pictures <= 0 || !workspace || !plan_out || sao_type < 1 || sao_type > 3)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int w = params->pic_w, h = params->pic_h;
  // the filters run without a chroma QP table (deblock.hip: identity), so a search that quantises chroma at another QP than luma
  // would be filtered with the wrong tc: refused rather than silently different (the reference's default table is the identity)
  if (params->qp_c != params->qp) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_plan_create: qp_c != qp needs a chroma QP table, which the loop plan does not take");
  std::vector<uvghip_ctu_picture_t> sp(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    if (!pictures[i].out_y || !pictures[i].out_u || !pictures[i].out_v) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_plan_create: output planes");
    if (pictures[i].out_stride < w || pictures[i].out_stride_c < w / 2) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_plan_create: output strides");
    sp[i] = pictures[i].search;
  }
  const layout_t L = layout_of(bitdepth, n_pictures, w, h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  uvghip_loop_plan *pl = new (std::nothrow) uvghip_loop_plan;
  if (!pl) return uvghip_set_error(hipErrorOutOfMemory, __func__);
  pl->search = nullptr;
  if (int rc = uvghip_ctu_plan_create(bitdepth, params, sp.data(), n_pictures, ws + L.search, &pl->search)) { delete pl; return rc; }
  const int wc = (w + 63) / 64, hc = (h + 63) / 64;
  pl->bitdepth = bitdepth; pl->n = n_pictures; pl->w = w; pl->h = h; pl->qp = params->qp; pl->lambda = params->lambda; pl->sao_type = sao_type;
  pl->ctus = wc * hc;
  pl->pics.assign(pictures, pictures + n_pictures);
  pl->snap = ws + L.snap;
  pl->snap_bytes = (size_t)w * h * 3 / 2 * (bitdepth == 8 ? 1 : 2);
  pl->rects_y = reinterpret_cast<uvghip_rect_t *>(ws + L.rects_y);
  pl->rects_c = reinterpret_cast<uvghip_rect_t *>(ws + L.rects_c);
  for (int c = 0; c < 3; ++c) {
    pl->edge[c] = reinterpret_cast<int32_t *>(ws + L.edge[c]); pl->band[c] = reinterpret_cast<int32_t *>(ws + L.band[c]);
    pl->params[c] = reinterpret_cast<uvghip_sao_param_t *>(ws + L.params[c]);
  }
  pl->decide_ws = ws + L.decide;
  pl->sao_info = reinterpret_cast<int32_t *>(ws + L.info);
  pl->sao_models = reinterpret_cast<uint16_t *>(ws + L.models);
  pl->coder_ws = ws + L.coder;
  pl->rows = ws + L.rows;
  pl->row_bytes = reinterpret_cast<int32_t *>(ws + L.row_bytes);
  pl->row_cap = L.row_cap; pl->hc = hc;
  pl->sums = reinterpret_cast<uint32_t *>(ws + L.sums);
  pl->ctu_params = *params;
  if (int rc = uvghip_slice_rows_prepare(params, sp.data(), n_pictures, pl->coder_ws)) { uvghip_ctu_plan_destroy(pl->search); delete pl; return rc; }
  // The in-loop filters as ONE launch behind the search, a workgroup per CTU (uvghip_filter_pictures, ctu_filter.h) -- the default;
  // UVGHIP_LOOP_UNFUSED=1 keeps the chain of whole-picture kernels (deblock snapshot, SAO statistics, decision, deblocking in place,
  // SAO apply: the same pictures and decisions, ~40 launches per picture).
  {
    const char *e = getenv("UVGHIP_LOOP_UNFUSED");
    pl->fused = !(e && e[0] == '1');
    pl->filt_ws = ws + L.filt;
    pl->coder_ticket = reinterpret_cast<int32_t *>(ws + L.tick);
    if (pl->fused) {
      std::vector<uvghip_pb_filter_t> fl(n_pictures);
      const size_t b = bitdepth == 8 ? 1 : 2, plane = (size_t)w * h * b;
      for (int i = 0; i < n_pictures; ++i) {
        uvghip_pb_filter_t &f = fl[i];
        unsigned char *d = pl->snap + (size_t)i * pl->snap_bytes;
        f.dbk_y = d; f.dbk_u = d + plane; f.dbk_v = d + plane + plane / 4; f.dbk_stride = w; f.dbk_stride_c = w / 2;
        f.out_y = pictures[i].out_y; f.out_u = pictures[i].out_u; f.out_v = pictures[i].out_v; f.out_stride = pictures[i].out_stride; f.out_stride_c = pictures[i].out_stride_c;
        f.sao_info = pl->sao_info + (size_t)i * pl->ctus * 34; f.sao_models = pl->sao_models + (size_t)i * pl->ctus * 6;
        f.sao_type = sao_type; f.reserved = 0;
      }
      if (int rc = uvghip_filter_pictures_prepare(bitdepth, params, sp.data(), fl.data(), n_pictures, 2, pl->filt_ws)) { uvghip_ctu_plan_destroy(pl->search); delete pl; return rc; }
    }
  }
  // the CTU grids clipped to the picture: the rectangles sao_search_luma / _chroma hand to the decision (sao.c:605-668)
  std::vector<uvghip_rect_t> ry(pl->ctus), rc(pl->ctus);
  for (int cy = 0; cy < hc; ++cy)
    for (int cx = 0; cx < wc; ++cx) {
      const int x = cx * 64, y = cy * 64, bw = x + 64 > w ? w - x : 64, bh = y + 64 > h ? h - y : 64;
      ry[cy * wc + cx] = uvghip_rect_t{x, y, bw, bh};
      rc[cy * wc + cx] = uvghip_rect_t{x / 2, y / 2, bw / 2, bh / 2};
    }
  hipError_t e = hipMemcpy(pl->rects_y, ry.data(), ry.size() * sizeof(uvghip_rect_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(pl->rects_c, rc.data(), rc.size() * sizeof(uvghip_rect_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) { uvghip_ctu_plan_destroy(pl->search); delete pl; return uvghip_set_error(e, "uvghip_loop_plan_create: rectangle tables"); }
  *plan_out = pl;
  return 0;
}

extern "C" int uvghip_loop_plan_run_filters(uvghip_loop_plan_t *pl, void *stream);
extern "C" int uvghip_loop_plan_run(uvghip_loop_plan_t *pl, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (int rc = uvghip_ctu_plan_run(pl->search, stream)) return rc;
  return uvghip_loop_plan_run_filters(pl, stream);
}

extern "C" int uvghip_ctu_plan_reset(uvghip_ctu_plan_t *pl, void *stream);
extern "C" int uvghip_ctu_plan_launch(uvghip_ctu_plan_t *pl, void *stream);
extern "C" const int32_t *uvghip_ctu_plan_done_flags(const uvghip_ctu_plan_t *pl);
extern "C" int uvghip_loop_plan_run_overlapped(uvghip_loop_plan_t *pl, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (!pl->fused) return uvghip_loop_plan_run(pl, stream);
  // Beside a search that fills the device the stage's workgroups and the coder's waves displace search workgroups (a CU's 160 KB of LDS are four
  // search workgroups exactly: one coder wave of 10 KB costs the CU a whole one) and the group gets SLOWER -- measured: 60 pictures of 1080p, up
  // to 1020 CTUs in progress on 1024 slots, 543 -> 621 ms; 16 pictures 484 -> 415 ms, one picture 461 -> 396 ms.  So: only while the
  // pictures' wavefronts leave half the device free.
  {
    const int wcx = (pl->w + 63) / 64;
    if ((long long)pl->n * (wcx < pl->hc ? wcx : pl->hc) > 512 && !getenv("UVGHIP_OVERLAP_ALWAYS")) return uvghip_loop_plan_run(pl, stream);
  }
  // what runs beside the search is capped: a waiting filter workgroup or coder wave holds LDS a search workgroup cannot use (a coder wave 10 KB
  // of a CU's 160 KB beside four search workgroups of 40 KB: one wave costs the CU a search workgroup)
  static const int filter_cap = [] { const char *e = getenv("UVGHIP_OVERLAP_FILTER_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
  static const int coder_cap = [] { const char *e = getenv("UVGHIP_OVERLAP_CODER_WAVES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
  hipStream_t st = uvghip_stream(stream);
  if (!pl->side[0]) {
    for (int i = 0; i < 2; ++i) {
      UVGHIP_TRY(hipStreamCreateWithFlags(&pl->side[i], hipStreamNonBlocking));
      UVGHIP_TRY(hipEventCreateWithFlags(&pl->ev_side[i], hipEventDisableTiming));
    }
    UVGHIP_TRY(hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming));
  }
  // all flags to zero in `stream`, the side streams behind that; then the search, so that it is in the queue before anything waits for it
  if (int rc = uvghip_ctu_plan_reset(pl->search, stream)) return rc;
  if (int rc = uvghip_filter_pictures_reset(pl->n, pl->w, pl->h, pl->filt_ws, stream)) return rc;
  UVGHIP_TRY(hipMemsetAsync(pl->coder_ticket, 0, sizeof(int32_t), st));
  UVGHIP_TRY(hipEventRecord(pl->ev_fork, st));
  if (int rc = uvghip_ctu_plan_launch(pl->search, stream)) return rc;
  for (int i = 0; i < 2; ++i) UVGHIP_TRY(hipStreamWaitEvent(pl->side[i], pl->ev_fork, 0));
  // the filter stage: as many persistent workgroups as the pictures' wavefronts can have CTUs in progress, at most an eighth of the device
  const int wc = (pl->w + 63) / 64, per = wc < pl->hc ? wc : pl->hc;
  long long g = (long long)per * pl->n;
  if (g > filter_cap) g = filter_cap;
  if (int rc = uvghip_filter_pictures_run_behind(pl->bitdepth, pl->n, pl->w, pl->h, pl->filt_ws, uvghip_ctu_plan_done_flags(pl->search), (int)g, pl->side[0])) return rc;
  if (int rc = uvghip_encode_slice_rows_behind_capped(pl->bitdepth, &pl->ctu_params, nullptr, pl->n, pl->sao_info, pl->sao_models,
                                                      uvghip_filter_pictures_final_flags(pl->n, pl->w, pl->h, pl->filt_ws), pl->coder_ticket, coder_cap, pl->coder_ws, pl->rows,
                                                      pl->row_cap, pl->row_bytes, pl->side[1]))
    return rc;
  for (int i = 0; i < 2; ++i) {
    UVGHIP_TRY(hipEventRecord(pl->ev_side[i], pl->side[i]));
    UVGHIP_TRY(hipStreamWaitEvent(st, pl->ev_side[i], 0));
  }
  return 0;
}

extern "C" int uvghip_loop_plan_run_search(uvghip_loop_plan_t *pl, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return uvghip_ctu_plan_run(pl->search, stream);
}

extern "C" int uvghip_loop_plan_run_filters(uvghip_loop_plan_t *pl, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  const size_t b = pl->bitdepth == 8 ? 1 : 2;
  const int w = pl->w, h = pl->h, cw = w / 2, ch = h / 2;
  if (pl->fused) {        // one launch for the filters of every picture, one for the slice data
    if (int rc = uvghip_filter_pictures_run(pl->bitdepth, pl->n, w, h, pl->filt_ws, stream)) return rc;
    return uvghip_encode_slice_rows(pl->bitdepth, &pl->ctu_params, nullptr, pl->n, pl->sao_info, pl->sao_models, pl->coder_ws, pl->rows, pl->row_cap, pl->row_bytes, stream);
  }
  for (int i = 0; i < pl->n; ++i) {
    const uvghip_ctu_picture_t &p = pl->pics[i].search;
    unsigned char *sy = pl->snap + (size_t)i * pl->snap_bytes, *su = sy + (size_t)w * h * b, *sv = su + (size_t)cw * ch * b;
    UVGHIP_TRY(hipMemcpy2DAsync(sy, (size_t)w * b, p.rec_y, (size_t)p.rec_stride * b, (size_t)w * b, h, hipMemcpyDeviceToDevice, st));
    UVGHIP_TRY(hipMemcpy2DAsync(su, (size_t)cw * b, p.rec_u, (size_t)p.rec_stride_c * b, (size_t)cw * b, ch, hipMemcpyDeviceToDevice, st));
    UVGHIP_TRY(hipMemcpy2DAsync(sv, (size_t)cw * b, p.rec_v, (size_t)p.rec_stride_c * b, (size_t)cw * b, ch, hipMemcpyDeviceToDevice, st));
    if (int rc = uvghip_deblock_frame_sao_snapshot(pl->bitdepth, sy, w, su, sv, cw, w, h, p.cu, p.cu_stride, 0, 0, 0, pl->qp, nullptr, stream)) return rc;
    const size_t o = (size_t)i * pl->ctus;
    if (int rc = uvghip_sao_stats_batch(pl->bitdepth, p.src_y, p.src_stride, sy, w, pl->rects_y, pl->ctus, pl->edge[0] + o * 40, pl->band[0] + o * 64, stream)) return rc;
    if (int rc = uvghip_sao_stats_batch(pl->bitdepth, p.src_u, p.src_stride_c, su, cw, pl->rects_c, pl->ctus, pl->edge[1] + o * 40, pl->band[1] + o * 64, stream)) return rc;
    if (int rc = uvghip_sao_stats_batch(pl->bitdepth, p.src_v, p.src_stride_c, sv, cw, pl->rects_c, pl->ctus, pl->edge[2] + o * 40, pl->band[2] + o * 64, stream)) return rc;
  }
  if (int rc = uvghip_sao_decide_pictures(pl->bitdepth, pl->n, w, h, pl->qp, pl->lambda, pl->sao_type, pl->edge[0], pl->band[0], pl->edge[1], pl->band[1],
                                          pl->edge[2], pl->band[2], pl->decide_ws, pl->sao_info, pl->sao_models, pl->params[0], pl->params[1],
                                          pl->params[2], stream))
    return rc;
  for (int i = 0; i < pl->n; ++i) {
    const uvghip_loop_picture_t &q = pl->pics[i];
    const uvghip_ctu_picture_t &p = q.search;
    if (int rc = uvghip_deblock_frame(pl->bitdepth, p.rec_y, p.rec_stride, p.rec_u, p.rec_v, p.rec_stride_c, w, h, p.cu, p.cu_stride, 0, 0, 0, pl->qp, nullptr, stream)) return rc;
    const size_t o = (size_t)i * pl->ctus;
    if (int rc = uvghip_sao_apply_batch(pl->bitdepth, p.rec_y, p.rec_stride, q.out_y, q.out_stride, w, h, pl->rects_y, pl->params[0] + o, pl->ctus, stream)) return rc;
    if (int rc = uvghip_sao_apply_batch(pl->bitdepth, p.rec_u, p.rec_stride_c, q.out_u, q.out_stride_c, cw, ch, pl->rects_c, pl->params[1] + o, pl->ctus, stream)) return rc;
    if (int rc = uvghip_sao_apply_batch(pl->bitdepth, p.rec_v, p.rec_stride_c, q.out_v, q.out_stride_c, cw, ch, pl->rects_c, pl->params[2] + o, pl->ctus, stream)) return rc;
  }
  // the slice data: every WPP row's substream from the levels, the side information and the SAO decisions
  return uvghip_encode_slice_rows(pl->bitdepth, &pl->ctu_params, nullptr, pl->n, pl->sao_info, pl->sao_models, pl->coder_ws, pl->rows, pl->row_cap,
                                  pl->row_bytes, stream);
}

// ---- an all-intra plan whose pictures are filtered ELSEWHERE: the I pictures of a clip, searched here beside the in-flight P / B launch that
// filters them CTU by CTU as they are searched (uvghip_loop_pb_run_inflight_ext) ----
extern "C" int uvghip_ctu_plan_reset(uvghip_ctu_plan_t *pl, void *stream);
extern "C" int uvghip_ctu_plan_launch(uvghip_ctu_plan_t *pl, void *stream);
extern "C" int uvghip_ctu_plan_set_grid(uvghip_ctu_plan_t *pl, int max_workgroups);
extern "C" const int32_t *uvghip_ctu_plan_done_flags(const uvghip_ctu_plan_t *pl);
extern "C" int uvghip_loop_plan_search_reset(uvghip_loop_plan_t *pl, void *stream) { return pl ? uvghip_ctu_plan_reset(pl->search, stream) : uvghip_set_error(hipErrorInvalidValue, __func__); }
extern "C" int uvghip_loop_plan_search_launch(uvghip_loop_plan_t *pl, void *stream) { return pl ? uvghip_ctu_plan_launch(pl->search, stream) : uvghip_set_error(hipErrorInvalidValue, __func__); }
extern "C" int uvghip_loop_plan_set_search_grid(uvghip_loop_plan_t *pl, int max_workgroups) { return pl ? uvghip_ctu_plan_set_grid(pl->search, max_workgroups) : uvghip_set_error(hipErrorInvalidValue, __func__); }
extern "C" const int32_t *uvghip_loop_plan_searched_flags(const uvghip_loop_plan_t *pl) { return pl ? uvghip_ctu_plan_done_flags(pl->search) : nullptr; }
// the slice data alone (the pictures' SAO decisions are in the plan's arrays: uvghip_loop_plan_results)
extern "C" int uvghip_loop_plan_run_coder(uvghip_loop_plan_t *pl, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return uvghip_encode_slice_rows(pl->bitdepth, &pl->ctu_params, nullptr, pl->n, pl->sao_info, pl->sao_models, pl->coder_ws, pl->rows, pl->row_cap, pl->row_bytes, stream);
}

// ... beside the launch that is still filtering the plan's pictures (I pictures in the flight): the rows wait for that launch's per-CTU flags
extern "C" int uvghip_loop_plan_run_coder_behind(uvghip_loop_plan_t *pl, const int32_t *final_flags, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl || !final_flags) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return uvghip_encode_slice_rows_behind(pl->bitdepth, &pl->ctu_params, nullptr, pl->n, pl->sao_info, pl->sao_models, final_flags, pl->coder_ws, pl->rows, pl->row_cap,
                                         pl->row_bytes, stream);
}

extern "C" int uvghip_loop_plan_slice_data(const uvghip_loop_plan_t *pl, const uint8_t **rows, const int32_t **row_bytes, int *row_cap, int *n_rows)
{
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (rows) *rows = pl->rows;
  if (row_bytes) *row_bytes = pl->row_bytes;
  if (row_cap) *row_cap = pl->row_cap;
  if (n_rows) *n_rows = pl->hc;
  return 0;
}

// The NAL units of one picture of the group after a run: the checksum of its output picture on the device, its rows and their
// lengths brought to the host, uvghip_write_picture_nals.  Waits for the stream (the bytes are host memory).
extern "C" int uvghip_loop_plan_picture_nals(uvghip_loop_plan_t *pl, int picture, int poc, uint8_t *out, size_t cap, size_t *len, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl || picture < 0 || picture >= pl->n || poc < 0 || !len || (!out && cap)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  const uvghip_loop_picture_t &q = pl->pics[picture];
  uint32_t *d_sums = pl->sums + 3 * (size_t)picture;
  if (int rc = uvghip_picture_checksum(pl->bitdepth, q.out_y, q.out_stride, q.out_u, q.out_v, q.out_stride_c, pl->w, pl->h, d_sums, stream)) return rc;
  uint32_t sums[3];
  std::vector<int32_t> nb(pl->hc);
  UVGHIP_TRY(hipMemcpyAsync(sums, d_sums, sizeof sums, hipMemcpyDeviceToHost, st));
  UVGHIP_TRY(hipMemcpyAsync(nb.data(), pl->row_bytes + (size_t)picture * pl->hc, (size_t)pl->hc * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  UVGHIP_TRY(hipStreamSynchronize(st));
  size_t pitch = 1;
  for (int r = 0; r < pl->hc; ++r) {
    if (nb[r] <= 0 || nb[r] > pl->row_cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_plan_picture_nals: a row overflowed its slot (or the plan has not run)");
    if ((size_t)nb[r] > pitch) pitch = (size_t)nb[r];
  }
  std::vector<uint8_t> rows(pitch * pl->hc);
  for (int r = 0; r < pl->hc; ++r)
    UVGHIP_TRY(hipMemcpyAsync(rows.data() + (size_t)r * pitch, pl->rows + ((size_t)picture * pl->hc + r) * pl->row_cap, (size_t)nb[r], hipMemcpyDeviceToHost, st));
  UVGHIP_TRY(hipStreamSynchronize(st));
  return uvghip_write_picture_nals(poc, 1, rows.data(), pitch, nb.data(), pl->hc, sums, out, cap, len);
}

// The NAL units of ALL pictures of the group after a run, as pictures first_poc, first_poc + 1, ... : what n calls of
// uvghip_loop_plan_picture_nals write, one after the other into `out`, with two waits for the stream instead of 2 n and one download of the
// rows instead of n * rows: the checksums of all pictures, then their sums and row lengths in one copy; the rows gathered on the device
// (picture p at a 16-byte-aligned base, its rows at the picture's own pitch) into one buffer that comes over in one copy to pinned memory.
namespace {
__global__ void nal_layout_kernel(const int32_t *__restrict__ row_bytes, int n, int hc, unsigned long long *__restrict__ base)
{
  if (threadIdx.x || blockIdx.x) return;
  unsigned long long at = 0;
  unsigned long long *pitch = base + n + 1;
  for (int p = 0; p < n; ++p) {
    int m = 1;
    for (int r = 0; r < hc; ++r) { const int b = row_bytes[(size_t)p * hc + r]; m = b > m ? b : m; }
    const unsigned long long pt = ((unsigned long long)m + 15) & ~15ull;
    base[p] = at; pitch[p] = pt;
    at += pt * hc;
  }
  base[n] = at;
}
__global__ void __launch_bounds__(256) nal_gather_kernel(const uint8_t *__restrict__ rows, int row_cap, const int32_t *__restrict__ row_bytes, int n, int hc,
                                                         const unsigned long long *__restrict__ base, uint8_t *__restrict__ packed)
{
  const int p = blockIdx.x / hc, r = blockIdx.x - p * hc;
  const int nb = row_bytes[blockIdx.x];
  if (nb <= 0 || nb > row_cap) return;
  const uint4 *src = reinterpret_cast<const uint4 *>(rows + (size_t)blockIdx.x * row_cap);
  uint4 *dst = reinterpret_cast<uint4 *>(packed + base[p] + (size_t)r * base[n + 1 + p]);
  for (int i = threadIdx.x; i < (nb + 15) / 16; i += blockDim.x) dst[i] = src[i];
}
}  // namespace

extern "C" int uvghip_loop_plan_group_nals(uvghip_loop_plan_t *pl, int first_poc, uint8_t *out, size_t cap, size_t *lens, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl || first_poc < 0 || !lens || (!out && cap)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (pl->row_cap & 15) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_plan_group_nals: the plan's row slots are not 16-byte multiples");
  hipStream_t st = uvghip_stream(stream);
  const int n = pl->n, hc = pl->hc;
  for (int i = 0; i < n; ++i) {
    const uvghip_loop_picture_t &q = pl->pics[i];
    if (int rc = uvghip_picture_checksum(pl->bitdepth, q.out_y, q.out_stride, q.out_u, q.out_v, q.out_stride_c, pl->w, pl->h, pl->sums + 3 * (size_t)i, stream)) return rc;
  }
  if (!pl->pack_base) UVGHIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->pack_base), (size_t)(2 * n + 1) * sizeof(unsigned long long)));
  hipLaunchKernelGGL(nal_layout_kernel, dim3(1), dim3(1), 0, st, pl->row_bytes, n, hc, pl->pack_base);
  std::vector<uint32_t> sums((size_t)3 * n);
  std::vector<int32_t> nb((size_t)n * hc);
  UVGHIP_TRY(hipMemcpyAsync(sums.data(), pl->sums, sums.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  UVGHIP_TRY(hipMemcpyAsync(nb.data(), pl->row_bytes, nb.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  UVGHIP_TRY(hipStreamSynchronize(st));
  // the same layout on the host
  std::vector<size_t> base((size_t)n + 1), pitch(n);
  size_t at = 0;
  for (int p = 0; p < n; ++p) {
    int m = 1;
    for (int r = 0; r < hc; ++r) {
      const int b = nb[(size_t)p * hc + r];
      if (b <= 0 || b > pl->row_cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_plan_group_nals: a row overflowed its slot (or the plan has not run)");
      m = b > m ? b : m;
    }
    base[p] = at; pitch[p] = ((size_t)m + 15) & ~(size_t)15;
    at += pitch[p] * hc;
  }
  base[n] = at;
  if (at > pl->pack_cap) {
    if (pl->pack_dev) { UVGHIP_TRY(hipFree(pl->pack_dev)); pl->pack_dev = nullptr; }
    if (pl->pack_host) { UVGHIP_TRY(hipHostFree(pl->pack_host)); pl->pack_host = nullptr; }
    pl->pack_cap = 0;
    const size_t want = at + at / 4 + 4096;
    UVGHIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->pack_dev), want));
    UVGHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pl->pack_host), want, hipHostMallocDefault));
    pl->pack_cap = want;
  }
  hipLaunchKernelGGL(nal_gather_kernel, dim3(n * hc), dim3(256), 0, st, pl->rows, pl->row_cap, pl->row_bytes, n, hc, pl->pack_base, pl->pack_dev);
  UVGHIP_TRY(hipGetLastError());
  UVGHIP_TRY(hipMemcpyAsync(pl->pack_host, pl->pack_dev, at, hipMemcpyDeviceToHost, st));
  UVGHIP_TRY(hipStreamSynchronize(st));
  size_t used = 0;
  for (int p = 0; p < n; ++p) {
    size_t len = 0;
    if (int rc = uvghip_write_picture_nals(first_poc + p, 1, pl->pack_host + base[p], pitch[p], nb.data() + (size_t)p * hc, hc, sums.data() + 3 * (size_t)p,
                                           out ? out + used : nullptr, cap > used ? cap - used : 0, &len)) return rc;
    lens[p] = len;
    used += len;
  }
  return 0;
}

extern "C" int uvghip_loop_plan_results(const uvghip_loop_plan_t *pl, const int32_t **sao_info, const uint16_t **sao_models)
{
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (sao_info) *sao_info = pl->sao_info;
  if (sao_models) *sao_models = pl->sao_models;
  return 0;
}

// The ALF stage of the group (BASELINE configs[3]: --alf full), after uvghip_loop_plan_run: where the encoder runs uvg_alf_enc_process
// (src/alf.c:5193) on a picture whose search and SAO the device did.  Per picture the caller's `decide` is called with the picture's planes
// -- what ALF gets (the plan's SAO output) and the source -- and returns the decisions; the derivation itself (alf_encoder :3994,
// alf_encoder_ctb :4369, derive_cc_alf_filter :2212 on the statistics of alf_derive_stats_for_filtering :4227, which `decide` gathers with
// uvghip_alf_classify_frame / uvghip_alf_stats_compact_batch / uvghip_alf_cov_reduce / uvghip_cc_alf_stats_batch on the given planes and
// stream) stays host code.  Then uvghip_alf_reconstruct_picture into the caller's output planes and, for the whole group,
// uvghip_encode_slice_rows_alf: the slice data with the CTU-level ALF syntax.
extern "C" size_t uvghip_loop_plan_alf_workspace_bytes(const uvghip_loop_plan_t *pl)
{
  if (!pl) return 0;
  return align_up(uvghip_alf_reconstruct_workspace_bytes(pl->w, pl->h), 256) + align_up(uvghip_slice_rows_alf_workspace_bytes(pl->n), 256) +
         (size_t)pl->n * align_up((size_t)pl->ctus * 7 + (size_t)pl->ctus * sizeof(int16_t), 256);
}

extern "C" int uvghip_loop_plan_alf_stage(uvghip_loop_plan_t *pl, uvghip_alf_decide_fn decide, void *user, int classification_shift, const uvghip_alf_planes_t *alf_out,
                                          void *workspace, uint8_t *rows, int row_cap, int32_t *row_bytes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl || !decide || !alf_out || !workspace || !rows || row_cap <= 0 || !row_bytes || classification_shift < 8 || classification_shift > 20)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  unsigned char *ws_rec = ws, *ws_coder = ws_rec + align_up(uvghip_alf_reconstruct_workspace_bytes(pl->w, pl->h), 256);
  unsigned char *ws_flags = ws_coder + align_up(uvghip_slice_rows_alf_workspace_bytes(pl->n), 256);
  const size_t per = align_up((size_t)pl->ctus * 7 + (size_t)pl->ctus * sizeof(int16_t), 256), b = pl->bitdepth == 8 ? 1 : 2;
  std::vector<uvghip_slice_alf_t> sl(pl->n);
  std::vector<uvghip_ctu_picture_t> cp(pl->n);
  for (int i = 0; i < pl->n; ++i) {
    const uvghip_loop_picture_t &q = pl->pics[i];
    const uvghip_alf_planes_t &o = alf_out[i];
    if (!o.y || !o.u || !o.v || o.stride < pl->w || o.stride_c < pl->w / 2) return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_plan_alf_stage: output planes");
    uvghip_alf_decision_t d;
    memset(&d, 0, sizeof d);
    if (int rc = decide(user, i, &q, &d)) return uvghip_set_error(hipErrorInvalidValue, rc > 0 ? "uvghip_loop_plan_alf_stage: decide() failed" : "uvghip_loop_plan_alf_stage: decide() refused");
    const bool any = d.enabled[0] || d.enabled[1] || d.enabled[2] || d.cc_enabled[0] || d.cc_enabled[1];
    if ((d.alf_type != 1 && d.alf_type != 2) || d.n_luma_aps < 0 || d.n_luma_aps > 8 || (any && (!d.ctu_flags || !d.filter_set_idx)) || (d.n_luma_aps && !d.luma_aps) ||
        ((d.enabled[1] || d.enabled[2]) && !d.chroma_aps) || ((d.cc_enabled[0] || d.cc_enabled[1]) && !d.cc_coeff))
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_loop_plan_alf_stage: the decision decide() returned");
    if (any) {
      uvghip_alf_picture_t a;
      memset(&a, 0, sizeof a);
      a.in_y = q.out_y; a.in_u = q.out_u; a.in_v = q.out_v; a.in_stride = q.out_stride; a.in_stride_c = q.out_stride_c;
      a.out_y = o.y; a.out_u = o.u; a.out_v = o.v; a.out_stride = o.stride; a.out_stride_c = o.stride_c;
      a.width = pl->w; a.height = pl->h;
      for (int c = 0; c < 3; ++c) a.slice_enabled[c] = d.enabled[c];
      a.n_luma_aps = d.n_luma_aps; a.ctu_flags = d.ctu_flags; a.filter_set_idx = d.filter_set_idx; a.luma_aps = d.luma_aps; a.chroma_aps = d.chroma_aps;
      a.alf_full = d.alf_type == 2; a.cc_alf_enabled[0] = d.cc_enabled[0]; a.cc_alf_enabled[1] = d.cc_enabled[1]; a.cc_coeff = d.cc_coeff;
      a.classification_shift = classification_shift;
      if (int rc = uvghip_alf_reconstruct_picture(pl->bitdepth, &a, ws_rec, stream)) return rc;
    } else {          // a picture ALF leaves alone
      UVGHIP_TRY(hipMemcpy2DAsync(o.y, (size_t)o.stride * b, q.out_y, (size_t)q.out_stride * b, (size_t)pl->w * b, pl->h, hipMemcpyDeviceToDevice, st));
      UVGHIP_TRY(hipMemcpy2DAsync(o.u, (size_t)o.stride_c * b, q.out_u, (size_t)q.out_stride_c * b, (size_t)(pl->w / 2) * b, pl->h / 2, hipMemcpyDeviceToDevice, st));
      UVGHIP_TRY(hipMemcpy2DAsync(o.v, (size_t)o.stride_c * b, q.out_v, (size_t)q.out_stride_c * b, (size_t)(pl->w / 2) * b, pl->h / 2, hipMemcpyDeviceToDevice, st));
    }
    // the slice's side of the decision: the flags and set indices on the device, where the coder reads them
    uvghip_slice_alf_t &s = sl[i];
    memset(&s, 0, sizeof s);
    s.alf_type = d.alf_type; s.n_luma_aps = d.n_luma_aps; s.n_alternatives_chroma = d.chroma_aps ? d.chroma_aps[112] : 0;
    for (int c = 0; c < 3; ++c) s.enabled[c] = d.enabled[c];
    for (int c = 0; c < 2; ++c) { s.cc_enabled[c] = d.cc_enabled[c]; s.cc_filter_count[c] = d.cc_filter_count[c]; }
    unsigned char *fl = ws_flags + (size_t)i * per;
    if (d.ctu_flags && d.filter_set_idx) {
      UVGHIP_TRY(hipMemcpyAsync(fl, d.ctu_flags, (size_t)pl->ctus * 7, hipMemcpyHostToDevice, st));
      UVGHIP_TRY(hipMemcpyAsync(fl + (size_t)pl->ctus * 7 + ((size_t)pl->ctus & 1), d.filter_set_idx, (size_t)pl->ctus * sizeof(int16_t), hipMemcpyHostToDevice, st));
      UVGHIP_TRY(hipStreamSynchronize(st));          // (decide()'s arrays need not outlive the next call)
      s.ctu_flags = fl; s.filter_set_idx = reinterpret_cast<const int16_t *>(fl + (size_t)pl->ctus * 7 + ((size_t)pl->ctus & 1));
    }
    cp[i] = q.search;
  }
  return uvghip_encode_slice_rows_alf(pl->bitdepth, &pl->ctu_params, cp.data(), sl.data(), pl->n, pl->sao_info, pl->sao_models, ws_coder, rows, row_cap, row_bytes, stream);
}

extern "C" void uvghip_loop_plan_destroy(uvghip_loop_plan_t *pl)
{
  if (!pl) return;
  uvghip_ctu_plan_destroy(pl->search);
  if (pl->pack_dev) (void)hipFree(pl->pack_dev);
  if (pl->pack_host) (void)hipHostFree(pl->pack_host);
  if (pl->pack_base) (void)hipFree(pl->pack_base);
  for (int i = 0; i < 2; ++i) {
    if (pl->side[i]) (void)hipStreamDestroy(pl->side[i]);
    if (pl->ev_side[i]) (void)hipEventDestroy(pl->ev_side[i]);
  }
  if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
  delete pl;
}
