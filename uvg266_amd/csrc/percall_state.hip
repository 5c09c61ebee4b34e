// Per-call ("drop-in") entry points for the strategies whose reference typedef takes encoder_state_t* / lcu_t*:
// quant, dequant, quantize_residual (strategies-quant.h:48-86) and inter_recon_bipred (strategies-picture.h:136-148).
// Those structs are encoder-private (kilobytes of configuration, compile-option dependent), so the typedef-exact
// functions live in a shim compiled INSIDE the encoder tree (integration/uvg266_hip_shim.c) that only extracts fields
// into the plain-value views of include/uvg266_hip.h and calls the functions below.  Everything else -- staging on the
// calling thread's stream, branch selection as uvg_quantize_residual does it, the launches, the download -- is here.
// Host buffers in, host buffers out; re-entrant (per-thread arena, percall.h); no CPU arithmetic on samples.
#include "uvghip_common.h"
#include "percall.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

// uvg_get_scaled_qp (transform.c:150-165)
int scaled_qp(const uvghip_state_view_t *sv, int color)
{
  const int off = (sv->bitdepth - 8) * 6;
  if (color == 0) return sv->qp + off;
  return sv->qp_map[sv->qp < 0 ? 0 : (sv->qp > 63 ? 63 : sv->qp)] + off;
}

[[noreturn]] void unsupported(const char *what)
{
  // the strategy typedefs have no error channel: a configuration this backend does not implement must not
  // produce a silently different result.  The shim's registrar has the reference's registrar signature (opaque, bitdepth) and
  // cannot see the configuration, so it registers unconditionally; the integrator's duty is to ask
  // uvg_hip_state_config_supported(cfg) before uvg_strategyselector_init and to leave these five strategy types to the generic
  // backend (UVG_OVERRIDE_<type>=generic) when it says no.  Reaching this line means that check was skipped.
  fprintf(stderr, "uvg266hip: %s is not implemented by the hip backend (register the generic strategy for this configuration)\n", what);
  abort();
}

void check_view(const uvghip_state_view_t *sv)
{
  if (sv->bitdepth != 8 && sv->bitdepth != 10) unsupported("this bit depth");
  if (sv->scaling_list_enabled) unsupported("scaling lists");
  if (sv->dep_quant) unsupported("dependent quantisation");
}

}  // namespace

extern "C" unsigned uvghip_quant_percall(const uvghip_state_view_t *sv, const int16_t *coef, int16_t *q_coef, int32_t width,
                                         int32_t height, int color, int scan_idx, int block_type, int transform_skip, int lfnst_idx)
{
  (void)scan_idx; (void)block_type;
  check_view(sv);
  const size_t bytes = (size_t)width * height * 2;
  percall_ctx *c = percall_get(2 * bytes + 1024);
  const size_t oi = c->take(bytes), oo = c->take(bytes);
  memcpy(c->hp<int16_t>(oi), coef, bytes);
  c->upload(oi, bytes);
  if (sv->signhide_enable)
    c->must(uvghip_quant_signhide_batch(sv->bitdepth, c->dp<int16_t>(oi), c->dp<int16_t>(oo), width, height, 1, scaled_qp(sv, color),
                                        transform_skip, sv->slice_is_intra, lfnst_idx, c->stream), "quant (sign hiding)");
  else
    c->must((lfnst_idx ? uvghip_quant_lfnst_batch : uvghip_quant_batch)(sv->bitdepth, c->dp<int16_t>(oi), c->dp<int16_t>(oo), width, height, 1,
                                                                         scaled_qp(sv, color), transform_skip, sv->slice_is_intra, c->stream),
            "quant");
  c->download(oo, bytes);
  c->sync();
  memcpy(q_coef, c->hp<int16_t>(oo), bytes);
  return 0;
}

extern "C" unsigned uvghip_dequant_percall(const uvghip_state_view_t *sv, const int16_t *q_coef, int16_t *coef, int32_t width,
                                           int32_t height, int color, int block_type, int transform_skip)
{
  (void)block_type;
  check_view(sv);
  const size_t bytes = (size_t)width * height * 2;
  percall_ctx *c = percall_get(2 * bytes + 1024);
  const size_t oi = c->take(bytes), oo = c->take(bytes);
  memcpy(c->hp<int16_t>(oi), q_coef, bytes);
  c->upload(oi, bytes);
  c->must(uvghip_dequant_batch(sv->bitdepth, c->dp<int16_t>(oi), c->dp<int16_t>(oo), width, height, 1, scaled_qp(sv, color), transform_skip,
                               c->stream), "dequant");
  c->download(oo, bytes);
  c->sync();
  memcpy(coef, c->hp<int16_t>(oo), bytes);
  return 0;
}

extern "C" int uvghip_quantize_residual_percall(const uvghip_state_view_t *sv, const uvghip_cu_view_t *cu, int width, int height, int color,
                                                int scan_order, int use_trskip, int in_stride, int out_stride, const void *ref_in,
                                                const void *pred_in, void *rec_out, int16_t *coeff_out, int early_skip,
                                                int lmcs_chroma_adj, int tree_type)
{
  (void)scan_order; (void)lmcs_chroma_adj;
  check_view(sv);
  if (sv->lmcs_chroma_adj_enabled && color != 0) unsupported("LMCS chroma residual scaling");
  if (sv->rdoq_enable && use_trskip) unsupported("transform-skip RDOQ (uvg_ts_rdoq)");
  const int es = sv->bitdepth == 8 ? 1 : 2;
  // ---- what uvg_quantize_residual derives from the CU and the configuration (quant-generic.c:497-540) ----
  uvghip_qr_params_t p;
  memset(&p, 0, sizeof p);
  p.width = width; p.height = height; p.color = color;
  const int chroma_tree = tree_type == 2;                                          // UVG_CHROMA_T
  const int lfnst_index = (!chroma_tree || color == 0) ? cu->lfnst_idx : cu->cr_lfnst_idx;   // :505
  // uvg_transform2d (transform.c:258-275): the MTS path (with its LFNST zero-out) whenever cfg.mts, an LFNST index or a
  // rectangle is involved; uvg_get_tr_type decides the kernels
  uvghip_mts_select(width, height, color, cu->type, cu->type == 1 ? cu->isp_mode : 0, cu->lfnst_idx, cu->cr_lfnst_idx, cu->tr_idx, sv->mts,
                    &p.type_hor, &p.type_ver, &p.skip_width, &p.skip_height);
  p.qp_scaled = scaled_qp(sv, color);
  p.slice_is_intra = sv->slice_is_intra; p.cu_type = cu->type;
  p.use_trskip = use_trskip;
  p.rdoq_enable = sv->rdoq_enable; p.rdoq_skip = sv->rdoq_skip; p.dep_quant = 0;
  p.signhide_enable = sv->signhide_enable;
  p.cbf_u = (cu->cbf >> 1) & 1;                                                    // cbf_is_set(cbf, COLOR_U)
  p.mts_idx = cu->tr_idx; p.lfnst_idx = lfnst_index;
  p.lambda = color ? sv->c_lambda : sv->lambda;
  p.ctx = sv->cabac;
  // where the LFNST transform itself applies (quant-generic.c:507; uvg_fwd_lfnst, transform.c:978-1009)
  const bool separate = cu->log2_height + cu->log2_width < 6 || tree_type != 0;
  const bool mts_skip = cu->tr_idx == 1 && color == 0;                             // MTS_SKIP
  const bool lfnst_tr = sv->lfnst && cu->type == 1 && lfnst_index && !mts_skip && (color == 0 || separate);
  uvghip_lfnst_tu_t lt;
  if (lfnst_tr) {
    int mode = color == 0 ? cu->intra_mode : cu->intra_mode_chroma;
    if (mode >= 81 && mode <= 83) mode = sv->collocated_luma_mode;                  // CCLM: the co-located luma mode
    if (color == 0 && cu->mip_flag) mode = 0;                                       // MIP: planar
    int lw = 0, lh = 0;
    while ((1 << lw) < width) ++lw;
    while ((1 << lh) < height) ++lh;
    lt.intra_mode = (int8_t)mode; lt.lfnst_idx = (int8_t)lfnst_index;
    lt.log2_cu_width = (int8_t)(color == 0 ? cu->log2_width : lw);
    lt.log2_cu_height = (int8_t)(color == 0 ? cu->log2_height : lh);
  }
  // ---- stage: the two input blocks as packed width x height planes, one TU at (0, 0) ----
  const size_t blk = (size_t)width * height * es, cb = (size_t)width * height * 2;
  const size_t ws = uvghip_quantize_residual_workspace_bytes(&p, 1);
  percall_ctx *c = percall_get(3 * blk + cb + ws + 4096);
  const size_t o_ref = c->stage_block(ref_in, (size_t)in_stride, width, height, (size_t)es);
  const size_t o_pred = c->stage_block(pred_in, (size_t)in_stride, width, height, (size_t)es);
  const size_t o_tu = c->take(sizeof(uvghip_tu_t)), o_lt = c->take(sizeof lt);
  *c->hp<uvghip_tu_t>(o_tu) = uvghip_tu_t{0, 0};
  if (lfnst_tr) *c->hp<uvghip_lfnst_tu_t>(o_lt) = lt;
  c->upload(0, c->used);
  const size_t o_rec = c->take(blk), o_co = c->take(cb), o_has = c->take(16), o_ws = c->take(ws);
  c->must(uvghip_quantize_residual_batch(sv->bitdepth, &p, c->dp<char>(o_ref), width, c->dp<char>(o_pred), width, c->dp<char>(o_rec), width,
                                         c->dp<uvghip_tu_t>(o_tu), 1, lfnst_tr ? c->dp<uvghip_lfnst_tu_t>(o_lt) : nullptr, c->dp<int16_t>(o_co),
                                         c->dp<uint8_t>(o_has), c->dp<char>(o_ws), ws, c->stream), "quantize_residual");
  c->download(o_rec, (o_has + 16) - o_rec);
  c->sync();
  const int has = *c->hp<uint8_t>(o_has);
  memcpy(coeff_out, c->hp<int16_t>(o_co), cb);
  // :556-609: with coefficients (and not an early skip) the reconstruction; otherwise the prediction copied, unless
  // rec_out aliases pred_in
  const char *src = (has && !early_skip) ? c->hp<char>(o_rec) : nullptr;
  if (src) {
    for (int y = 0; y < height; ++y) memcpy((char *)rec_out + (size_t)y * out_stride * es, src + (size_t)y * width * es, (size_t)width * es);
  } else if (rec_out != pred_in) {
    for (int y = 0; y < height; ++y)
      memcpy((char *)rec_out + (size_t)y * out_stride * es, (const char *)pred_in + (size_t)y * in_stride * es, (size_t)width * es);
  }
  return has;
}

// uvg_quant_cbcr_residual (quant-generic.c:241-442) for one pair of chroma TUs.
extern "C" int uvghip_quant_cbcr_residual_percall(const uvghip_state_view_t *sv, const uvghip_cu_view_t *cu, int width, int height,
                                                  int scan_order, int in_stride, int out_stride, const void *u_ref_in, const void *v_ref_in,
                                                  const void *u_pred_in, const void *v_pred_in, void *u_rec_out, void *v_rec_out,
                                                  int16_t *coeff_out, int early_skip, int lmcs_chroma_adj, int tree_type)
{
  (void)scan_order; (void)lmcs_chroma_adj;
  check_view(sv);
  if (sv->lmcs_chroma_adj_enabled) unsupported("LMCS chroma residual scaling");
  const int es = sv->bitdepth == 8 ? 1 : 2;
  const int color = cu->joint_cb_cr == 1 ? 2 : 1;
  uvghip_qr_params_t p;
  memset(&p, 0, sizeof p);
  p.width = width; p.height = height; p.color = color;
  const int lfnst_index = tree_type == 2 ? cu->cr_lfnst_idx : cu->lfnst_idx;               // :306
  uvghip_mts_select(width, height, color, cu->type, 0, cu->lfnst_idx, cu->cr_lfnst_idx, cu->tr_idx, sv->mts, &p.type_hor, &p.type_ver,
                    &p.skip_width, &p.skip_height);
  p.qp_scaled = scaled_qp(sv, color);
  p.slice_is_intra = sv->slice_is_intra; p.cu_type = cu->type;
  p.rdoq_enable = sv->rdoq_enable; p.rdoq_skip = sv->rdoq_skip;
  p.signhide_enable = sv->signhide_enable;
  p.cbf_u = (cu->cbf >> 1) & 1;
  p.lfnst_idx = lfnst_index;
  p.lambda = sv->c_lambda;
  p.ctx = sv->cabac;
  // uvg_fwd_lfnst with COLOR_UV (:307-309): chroma rules -- only a separate tree carries an LFNST for chroma
  const bool separate = cu->log2_height + cu->log2_width < 6 || tree_type != 0;
  const bool lfnst_tr = lfnst_index && cu->type == 1 && separate;
  uvghip_lfnst_tu_t lt;
  if (lfnst_tr) {
    int mode = cu->intra_mode_chroma;
    if (mode >= 81 && mode <= 83) mode = sv->collocated_luma_mode;
    int lw = 0, lh = 0;
    while ((1 << lw) < width) ++lw;
    while ((1 << lh) < height) ++lh;
    lt.intra_mode = (int8_t)mode; lt.lfnst_idx = (int8_t)lfnst_index; lt.log2_cu_width = (int8_t)lw; lt.log2_cu_height = (int8_t)lh;
  }
  const size_t blk = (size_t)width * height * es, cb = (size_t)width * height * 2;
  const size_t ws = uvghip_quant_cbcr_residual_workspace_bytes(&p, 1);
  percall_ctx *c = percall_get(6 * blk + cb + ws + 4096);
  const size_t o_ur = c->stage_block(u_ref_in, (size_t)in_stride, width, height, (size_t)es);
  const size_t o_vr = c->stage_block(v_ref_in, (size_t)in_stride, width, height, (size_t)es);
  const size_t o_up = c->stage_block(u_pred_in, (size_t)in_stride, width, height, (size_t)es);
  const size_t o_vp = c->stage_block(v_pred_in, (size_t)in_stride, width, height, (size_t)es);
  const size_t o_tu = c->take(sizeof(uvghip_tu_t)), o_lt = c->take(sizeof lt);
  *c->hp<uvghip_tu_t>(o_tu) = uvghip_tu_t{0, 0};
  if (lfnst_tr) *c->hp<uvghip_lfnst_tu_t>(o_lt) = lt;
  c->upload(0, c->used);
  const size_t o_uo = c->take(blk), o_vo = c->take(blk), o_co = c->take(cb), o_ret = c->take(16), o_ws = c->take(ws);
  c->must(uvghip_quant_cbcr_residual_batch(sv->bitdepth, &p, cu->joint_cb_cr, sv->jccr_sign, c->dp<char>(o_ur), c->dp<char>(o_vr), width,
                                           c->dp<char>(o_up), c->dp<char>(o_vp), width, c->dp<char>(o_uo), c->dp<char>(o_vo), width,
                                           c->dp<uvghip_tu_t>(o_tu), 1, lfnst_tr ? c->dp<uvghip_lfnst_tu_t>(o_lt) : nullptr,
                                           c->dp<int16_t>(o_co), c->dp<uint8_t>(o_ret), early_skip, c->dp<char>(o_ws), ws, c->stream),
          "quant_cbcr_residual");
  c->download(o_uo, (o_ret + 16) - o_uo);
  c->sync();
  memcpy(coeff_out, c->hp<int16_t>(o_co), cb);
  for (int y = 0; y < height; ++y) {
    memcpy((char *)u_rec_out + (size_t)y * out_stride * es, c->hp<char>(o_uo) + (size_t)y * width * es, (size_t)width * es);
    memcpy((char *)v_rec_out + (size_t)y * out_stride * es, c->hp<char>(o_vo) + (size_t)y * width * es, (size_t)width * es);
  }
  return *c->hp<uint8_t>(o_ret);
}

// bipred_average_generic's three sample-wise forms (picture-generic.c:1132-1193) for one plane of a PU: the shim walks
// lcu->rec.{y,u,v} / the L0 / L1 buffers exactly as :1195-1262 does and calls this once per plane.
// l0 / l1: pu_w * pu_h contiguous samples, pixels or 14-bit int16 intermediates (l0_is_im / l1_is_im).
extern "C" void uvghip_bipred_average_percall(int bitdepth, void *dst, int dst_stride, const void *l0, int l0_is_im, const void *l1, int l1_is_im,
                                              unsigned pu_w, unsigned pu_h)
{
  if (bitdepth != 8 && bitdepth != 10) unsupported("this bit depth");
  const size_t n = (size_t)pu_w * pu_h, es = bitdepth == 8 ? 1 : 2;
  const size_t b0 = n * (l0_is_im ? 2 : es), b1 = n * (l1_is_im ? 2 : es);
  percall_ctx *c = percall_get(b0 + b1 + n * es + 2048);
  const size_t o0 = c->take(b0), o1 = c->take(b1);
  memcpy(c->hp<char>(o0), l0, b0);
  memcpy(c->hp<char>(o1), l1, b1);
  c->upload(0, c->used);
  const size_t oo = c->take(n * es);
  c->must(uvghip_bipred_average_batch(bitdepth, c->dp<char>(o0), c->dp<char>(o1), (l0_is_im ? 1 : 0) | (l1_is_im ? 2 : 0), n, c->dp<char>(oo),
                                      c->stream), "bipred average");
  c->download(oo, n * es);
  c->sync();
  for (unsigned y = 0; y < pu_h; ++y)
    memcpy((char *)dst + (size_t)y * dst_stride * es, c->hp<char>(oo) + (size_t)y * pu_w * es, (size_t)pu_w * es);
}
