/* Encoder-tree half of the hip backend for the strategies whose typedefs take encoder-private structs.
 *
 * This file is compiled INSIDE the uvg266 source tree (copy it to src/strategies/hip/ and add it to the build; see
 * INTEGRATION.md section 2): it is the only code of the backend that sees encoder_state_t / cu_info_t / lcu_t.  It does
 * field extraction and nothing else -- every sample and coefficient is handled by libuvg266hip.so behind the
 * plain-value views of uvg266_hip.h.  It is not part of libuvg266hip.so; the repository only syntax-checks it against
 * the encoder's headers (tools/refcheck/run.sh) and runs it against recording stand-ins of the four entry points
 * (tools/refcheck/rc_shim.inc) to prove that the extraction picks the fields the generic strategies read.
 *
 * Typedefs implemented: quant_func, dequant_func, quant_residual_func, quant_cbcr_func (strategies-quant.h:48-86),
 * inter_recon_bipred_func (strategies-picture.h:136-148), and the four of the alf group (strategies-alf.h:48-109).
 */
#include "strategyselector.h"
#include "encoderstate.h"
#include "encoder.h"
#include "cabac.h"
#include "context.h"
#include "cu.h"
#include "reshape.h"
#include "alf.h"
#include "videoframe.h"
#include "strategies/strategies-alf.h"
#include "uvg266_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void hip_snap(uint8_t *dst, const cabac_ctx_t *src, int n)
{
  for (int i = 0; i < n; ++i) dst[i] = (uint8_t)CTX_STATE(&src[i]);       /* cabac.h:175 */
}

/* What uvg_quant / uvg_dequant / uvg_rdoq / uvg_quantize_residual read through `state`. */
static void hip_state_view(const encoder_state_t *const state, uvghip_state_view_t *v, int with_cabac)
{
  const encoder_control_t *const ctrl = state->encoder_control;
  memset(v, 0, sizeof *v);
  v->bitdepth = ctrl->bitdepth;
  v->qp = state->qp;
  v->slice_is_intra = state->frame->slicetype == UVG_SLICE_I;
  v->rdoq_enable = ctrl->cfg.rdoq_enable;
  v->rdoq_skip = ctrl->cfg.rdoq_skip;
  v->dep_quant = ctrl->cfg.dep_quant;
  v->signhide_enable = ctrl->cfg.signhide_enable;
  v->scaling_list_enabled = ctrl->scaling_list.enable || ctrl->cfg.scaling_list != UVG_SCALING_LIST_OFF;
  v->lfnst = ctrl->cfg.lfnst;
  v->mts = ctrl->cfg.mts;
  v->lmcs_chroma_adj_enabled = state->tile->frame->lmcs_aps ? state->tile->frame->lmcs_aps->m_sliceReshapeInfo.enableChromaAdj : 0;
  v->collocated_luma_mode = state->collocated_luma_mode;
  v->jccr_sign = state->frame->jccr_sign;
  v->lambda = state->lambda;
  v->c_lambda = state->c_lambda;
  memcpy(v->qp_map, ctrl->qp_map[0], sizeof v->qp_map);
  if (with_cabac) {
    const cabac_data_t *const cb = &state->cabac;
    uvghip_rdoq_ctx_t *const s = &v->cabac;
    hip_snap(s->sig_group[0], &cb->ctx.sig_coeff_group_model[0], 2);
    hip_snap(s->sig_group[1], &cb->ctx.sig_coeff_group_model[2], 2);
    hip_snap(s->sig[0], cb->ctx.cu_sig_model_luma[0], 12);
    hip_snap(s->sig[1], cb->ctx.cu_sig_model_chroma[0], 8);
    hip_snap(s->par[0], cb->ctx.cu_parity_flag_model_luma, 21);
    hip_snap(s->par[1], cb->ctx.cu_parity_flag_model_chroma, 11);
    hip_snap(s->gt1[0], cb->ctx.cu_gtx_flag_model_luma[1], 21);
    hip_snap(s->gt1[1], cb->ctx.cu_gtx_flag_model_chroma[1], 11);
    hip_snap(s->gt2[0], cb->ctx.cu_gtx_flag_model_luma[0], 21);
    hip_snap(s->gt2[1], cb->ctx.cu_gtx_flag_model_chroma[0], 11);
    hip_snap(s->last_x[0], cb->ctx.cu_ctx_last_x_luma, 20);
    hip_snap(s->last_x[1], cb->ctx.cu_ctx_last_x_chroma, 3);
    hip_snap(s->last_y[0], cb->ctx.cu_ctx_last_y_luma, 20);
    hip_snap(s->last_y[1], cb->ctx.cu_ctx_last_y_chroma, 3);
    hip_snap(s->cbf_luma, cb->ctx.qt_cbf_model_luma, 4);
    hip_snap(s->cbf_cb, cb->ctx.qt_cbf_model_cb, 2);
    hip_snap(s->cbf_cr, cb->ctx.qt_cbf_model_cr, 3);
    hip_snap(&s->root_cbf, &cb->ctx.cu_qt_root_cbf_model, 1);
  }
}

static void hip_cu_view(const cu_info_t *const cu, uvghip_cu_view_t *v)
{
  memset(v, 0, sizeof *v);
  v->type = cu->type;
  v->tr_idx = cu->tr_idx;
  v->lfnst_idx = cu->lfnst_idx;
  v->cr_lfnst_idx = cu->cr_lfnst_idx;
  v->log2_width = cu->log2_width;
  v->log2_height = cu->log2_height;
  v->cbf = cu->cbf;
  v->joint_cb_cr = cu->joint_cb_cr;
  if (cu->type == CU_INTRA) {
    v->intra_mode = cu->intra.mode;
    v->intra_mode_chroma = cu->intra.mode_chroma;
    v->mip_flag = cu->intra.mip_flag;
    v->isp_mode = cu->intra.isp_mode;
  }
}

static unsigned uvg_quant_hip(const encoder_state_t *const state, coeff_t *coef, coeff_t *q_coef, int32_t width, int32_t height,
                              color_t color, int8_t scan_idx, int8_t block_type, int8_t transform_skip, uint8_t lfnst_idx)
{
  uvghip_state_view_t sv;
  hip_state_view(state, &sv, 0);
  return uvghip_quant_percall(&sv, coef, q_coef, width, height, color, scan_idx, block_type, transform_skip, lfnst_idx);
}

static unsigned uvg_dequant_hip(const encoder_state_t *const state, coeff_t *q_coef, coeff_t *coef, int32_t width, int32_t height,
                                color_t color, int8_t block_type, int8_t transform_skip)
{
  uvghip_state_view_t sv;
  hip_state_view(state, &sv, 0);
  return uvghip_dequant_percall(&sv, q_coef, coef, width, height, color, block_type, transform_skip);
}

static unsigned uvg_quantize_residual_hip(encoder_state_t *const state, const cu_info_t *const cur_cu, const int width, const int height,
                                          const color_t color, const coeff_scan_order_t scan_order, const int use_trskip,
                                          const int in_stride, const int out_stride, const uvg_pixel *const ref_in,
                                          const uvg_pixel *const pred_in, uvg_pixel *rec_out, coeff_t *coeff_out, bool early_skip,
                                          int lmcs_chroma_adj, enum uvg_tree_type tree_type)
{
  uvghip_state_view_t sv;
  uvghip_cu_view_t cv;
  hip_state_view(state, &sv, state->encoder_control->cfg.rdoq_enable);
  hip_cu_view(cur_cu, &cv);
  return (unsigned)uvghip_quantize_residual_percall(&sv, &cv, width, height, color, scan_order, use_trskip, in_stride, out_stride, ref_in,
                                                    pred_in, rec_out, coeff_out, early_skip, lmcs_chroma_adj, tree_type);
}

static int uvg_quant_cbcr_residual_hip(encoder_state_t *const state, const cu_info_t *const cur_cu, const int width, const int height,
                                       const coeff_scan_order_t scan_order, const int in_stride, const int out_stride,
                                       const uvg_pixel *const u_ref_in, const uvg_pixel *const v_ref_in, const uvg_pixel *const u_pred_in,
                                       const uvg_pixel *const v_pred_in, uvg_pixel *u_rec_out, uvg_pixel *v_rec_out, coeff_t *coeff_out,
                                       bool early_skip, int lmcs_chroma_adj, enum uvg_tree_type tree_type)
{
  uvghip_state_view_t sv;
  uvghip_cu_view_t cv;
  hip_state_view(state, &sv, state->encoder_control->cfg.rdoq_enable);
  hip_cu_view(cur_cu, &cv);
  return uvghip_quant_cbcr_residual_percall(&sv, &cv, width, height, scan_order, in_stride, out_stride, u_ref_in, v_ref_in, u_pred_in,
                                            v_pred_in, u_rec_out, v_rec_out, coeff_out, early_skip, lmcs_chroma_adj, tree_type);
}

/* bipred_average_generic's walk over the planes (picture-generic.c:1195-1262); the averaging itself is on the device. */
static void uvg_inter_recon_bipred_hip(lcu_t *const lcu, const yuv_t *const px_L0, const yuv_t *const px_L1, const yuv_im_t *const im_L0,
                                       const yuv_im_t *const im_L1, const unsigned pu_x, const unsigned pu_y, const unsigned pu_w,
                                       const unsigned pu_h, const unsigned im_flags_L0, const unsigned im_flags_L1,
                                       const bool predict_luma, const bool predict_chroma)
{
  if (predict_luma) {
    const unsigned off = SUB_SCU(pu_y) * LCU_WIDTH + SUB_SCU(pu_x);
    const int i0 = im_flags_L0 & 1, i1 = im_flags_L1 & 1;
    uvghip_bipred_average_percall(UVG_BIT_DEPTH, lcu->rec.y + off, LCU_WIDTH, i0 ? (const void *)im_L0->y : (const void *)px_L0->y, i0,
                                  i1 ? (const void *)im_L1->y : (const void *)px_L1->y, i1, pu_w, pu_h);
  }
  if (predict_chroma) {
    const unsigned off = SUB_SCU(pu_y) / 2 * LCU_WIDTH_C + SUB_SCU(pu_x) / 2;
    const int i0 = (im_flags_L0 & 2) != 0, i1 = (im_flags_L1 & 2) != 0;
    uvghip_bipred_average_percall(UVG_BIT_DEPTH, lcu->rec.u + off, LCU_WIDTH_C, i0 ? (const void *)im_L0->u : (const void *)px_L0->u, i0,
                                  i1 ? (const void *)im_L1->u : (const void *)px_L1->u, i1, pu_w / 2, pu_h / 2);
    uvghip_bipred_average_percall(UVG_BIT_DEPTH, lcu->rec.v + off, LCU_WIDTH_C, i0 ? (const void *)im_L0->v : (const void *)px_L0->v, i0,
                                  i1 ? (const void *)im_L1->v : (const void *)px_L1->v, i1, pu_w / 2, pu_h / 2);
  }
}

/* ---- the alf group (strategies-alf.h:48-109) ------------------------------------------------------------------------------------
 * The library's whole-picture kernels derive the virtual boundary from a block's place in the picture and clamp at the picture's edges
 * where the encoder's planes carry replicated padding (alf.c:5150-5170): what these four functions add is the extraction of the
 * planes, the conversion between the encoder's per-sample alf_classifier and the library's byte per 4x4 block, and the accumulation
 * into alf_covariance.  Anything alf.c never passes (another virtual boundary, a destination offset, other clipping values) aborts. */
static void hip_alf_die(const char *what)
{
  fprintf(stderr, "hip backend (alf): %s: %s\n", what, uvghip_last_error());
  abort();
}
static void hip_alf_vb(int is_chroma, int vb_ctu_height, int vb_pos)
{
  const int hgt = is_chroma ? LCU_WIDTH >> 1 : LCU_WIDTH, pos = hgt - (is_chroma ? ALF_VB_POS_ABOVE_CTUROW_CHMA : ALF_VB_POS_ABOVE_CTUROW_LUMA);
  if (vb_ctu_height != hgt || vb_pos != pos) hip_alf_die("a virtual boundary other than alf.c's (4:2:0, 64x64 CTUs)");
}
/* the plane's classes as one byte per 4x4 block (class_idx | transpose_idx << 5), rows of (w + 3) / 4 bytes; only the blocks of
 * [x, x + bw) x [y, y + bh) are filled in */
static uint8_t *hip_alf_class_bytes(alf_classifier **classifier, int w, int h, int x, int y, int bw, int bh)
{
  const int cw = (w + 3) / 4, ch = (h + 3) / 4;
  uint8_t *b = calloc((size_t)cw * ch, 1);
  if (!b) hip_alf_die("out of memory");
  for (int i = y; i < y + bh; i += 4)
    for (int j = x; j < x + bw; j += 4)
      b[(i / 4) * cw + j / 4] = (uint8_t)(classifier[i][j].class_idx | classifier[i][j].transpose_idx << 5);
  return b;
}

static void uvg_alf_derive_classification_blk_hip(encoder_state_t *const state, const int shift, const int n_height, const int n_width, const int blk_pos_x,
                                                  const int blk_pos_y, const int blk_dst_x, const int blk_dst_y, const int vb_ctu_height, int vb_pos)
{
  videoframe_t *const frame = state->tile->frame;
  hip_alf_vb(0, vb_ctu_height, vb_pos);
  uint8_t cls[(CLASSIFICATION_BLK_SIZE / 4) * (CLASSIFICATION_BLK_SIZE / 4)];
  if (n_width > CLASSIFICATION_BLK_SIZE || n_height > CLASSIFICATION_BLK_SIZE) hip_alf_die("classification block larger than 32x32");
  if (uvghip_alf_classify_percall(UVG_BIT_DEPTH, frame->rec->y, frame->rec->stride, frame->rec->width, frame->rec->height, shift, blk_pos_x, blk_pos_y, n_width, n_height, cls))
    hip_alf_die("classification");
  alf_classifier **classifier = frame->alf_info->classifier;
  for (int i = 0; i < n_height; ++i)
    for (int j = 0; j < n_width; ++j) {
      const uint8_t c = cls[(i / 4) * (n_width / 4) + j / 4];
      classifier[blk_dst_y + i][blk_dst_x + j].class_idx = c & 31;
      classifier[blk_dst_y + i][blk_dst_x + j].transpose_idx = c >> 5;
    }
}

static void hip_alf_filter(encoder_state_t *const state, int is_chroma, const uvg_pixel *src_pixels, uvg_pixel *dst_pixels, const int src_stride, const int dst_stride,
                           const short *filter_set, const int16_t *clip_set, const int width, const int height, int x_pos, int y_pos, int blk_dst_x, int blk_dst_y,
                           int vb_pos, const int vb_ctu_height)
{
  videoframe_t *const frame = state->tile->frame;
  const int w = frame->rec->width >> is_chroma, h = frame->rec->height >> is_chroma;
  hip_alf_vb(is_chroma, vb_ctu_height, vb_pos);
  uint8_t *cls = is_chroma ? NULL : hip_alf_class_bytes(frame->alf_info->classifier, w, h, x_pos, y_pos, width, height);
  uvg_pixel *blk = malloc((size_t)width * height * sizeof(uvg_pixel));
  if (!blk) hip_alf_die("out of memory");
  if (uvghip_alf_filter_percall(UVG_BIT_DEPTH, is_chroma, src_pixels, src_stride, w, h, x_pos, y_pos, width, height, filter_set, clip_set, cls, (w + 3) / 4, blk))
    hip_alf_die("filter");
  for (int i = 0; i < height; ++i) memcpy(dst_pixels + (size_t)(blk_dst_y + i) * dst_stride + blk_dst_x, blk + (size_t)i * width, (size_t)width * sizeof(uvg_pixel));
  free(blk); free(cls);
}
static void uvg_alf_filter_7x7_blk_hip(encoder_state_t *const state, const uvg_pixel *src_pixels, uvg_pixel *dst_pixels, const int src_stride, const int dst_stride,
                                       const short *filter_set, const int16_t *fClipSet, clp_rng clp_rng, const int width, const int height, int x_pos, int y_pos,
                                       int blk_dst_x, int blk_dst_y, int vb_pos, const int vb_ctu_height)
{
  (void)clp_rng;          /* [0, 2^depth - 1]: the kernel's own clamp */
  hip_alf_filter(state, 0, src_pixels, dst_pixels, src_stride, dst_stride, filter_set, fClipSet, width, height, x_pos, y_pos, blk_dst_x, blk_dst_y, vb_pos, vb_ctu_height);
}
static void uvg_alf_filter_5x5_blk_hip(encoder_state_t *const state, const uvg_pixel *src_pixels, uvg_pixel *dst_pixels, const int src_stride, const int dst_stride,
                                       const short *filter_set, const int16_t *fClipSet, clp_rng clp_rng, const int width, const int height, int x_pos, int y_pos,
                                       int blk_dst_x, int blk_dst_y, int vb_pos, const int vb_ctu_height)
{
  (void)clp_rng;
  hip_alf_filter(state, 1, src_pixels, dst_pixels, src_stride, dst_stride, filter_set, fClipSet, width, height, x_pos, y_pos, blk_dst_x, blk_dst_y, vb_pos, vb_ctu_height);
}

static void uvg_alf_get_blk_stats_hip(encoder_state_t *const state, channel_type channel, alf_covariance *cov, alf_classifier **g_classifier, uvg_pixel *org,
                                      int32_t org_stride, uvg_pixel *rec, int32_t rec_stride, const int x_pos, const int y_pos, const int x_dst, const int y_dst,
                                      const int width, const int height, int vb_ctu_height, int vb_pos, short alf_clipping_values[MAX_NUM_CHANNEL_TYPE][MAX_ALF_NUM_CLIPPING_VALUES])
{
  videoframe_t *const frame = state->tile->frame;
  const int is_chroma = channel != CHANNEL_TYPE_LUMA;
  const int w = frame->rec->width >> is_chroma, h = frame->rec->height >> is_chroma, ncls = g_classifier ? MAX_NUM_ALF_CLASSES : 1, ncoef = is_chroma ? 7 : 13;
  (void)alf_clipping_values;          /* the defaults of the bit depth (alf.c:5248-5260): what the kernel computes with */
  hip_alf_vb(is_chroma, vb_ctu_height, vb_pos);
  if (x_dst != x_pos || y_dst != y_pos || (!is_chroma) != (g_classifier != NULL)) hip_alf_die("a call alf.c does not make");
  uint8_t *cls = g_classifier ? hip_alf_class_bytes(g_classifier, w, h, x_pos, y_pos, width, height) : NULL;
  int64_t *ee = malloc((size_t)ncls * 13 * 13 * 16 * sizeof(int64_t)), *pix = malloc((size_t)ncls * sizeof(int64_t));
  int32_t *yv = malloc((size_t)ncls * 13 * 4 * sizeof(int32_t));
  if (!ee || !pix || !yv) hip_alf_die("out of memory");
  /* org / rec point at the block (alf.c:4305-4306): the planes start x_pos + y_pos * stride before */
  if (uvghip_alf_stats_percall(UVG_BIT_DEPTH, is_chroma, org - ((size_t)y_pos * org_stride + x_pos), org_stride, rec - ((size_t)y_pos * rec_stride + x_pos), rec_stride, w, h,
                               x_pos, y_pos, width, height, cls, (w + 3) / 4, ee, yv, pix))
    hip_alf_die("statistics");
  for (int c = 0; c < ncls; ++c) {
    for (int k = 0; k < ncoef; ++k) {
      for (int l = k; l < ncoef; ++l)
        for (int b0 = 0; b0 < 4; ++b0)
          for (int b1 = 0; b1 < 4; ++b1) cov[c].ee[k][l][b0][b1] += ee[((((size_t)c * 13 + k) * 13 + l) * 4 + b0) * 4 + b1];
      for (int b = 0; b < 4; ++b) cov[c].y[k][b] += yv[((size_t)c * 13 + k) * 4 + b];
    }
    cov[c].pix_acc += (double)pix[c];
    for (int k = 1; k < ncoef; ++k)          /* the lower triangle mirrors the upper (alf-generic.c:982-998) */
      for (int l = 0; l < k; ++l)
        for (int b0 = 0; b0 < 4; ++b0)
          for (int b1 = 0; b1 < 4; ++b1) cov[c].ee[k][l][b0][b1] = cov[c].ee[l][k][b1][b0];
  }
  free(ee); free(pix); free(yv); free(cls);
}

/* The configurations the backend implements for these four strategies.  Call after the configuration is parsed; when it
 * returns 0 start the encoder with UVG_OVERRIDE_quant=generic UVG_OVERRIDE_dequant=generic
 * UVG_OVERRIDE_quantize_residual=generic (strategyselector.c reads them) -- the entry points abort rather than return a
 * result that differs from the generic strategy's. */
int uvg_hip_state_config_supported(const uvg_config *const cfg)
{
  return !cfg->dep_quant && cfg->scaling_list == UVG_SCALING_LIST_OFF && !cfg->lmcs_enable &&
         !(cfg->rdoq_enable && cfg->trskip_enable);
}

/* Which groups take the hip backend: UVG266_HIP unset -> none; "1" / "all" -> every group; otherwise a comma-separated list of
 * group names ("picture,dct").  Used by the registration blocks of INTEGRATION.md section 1. */
int uvg_hip_group_enabled(const char *group)
{
  const char *e = getenv("UVG266_HIP");
  if (!e || !*e || !strcmp(e, "0")) return 0;
  if (!strcmp(e, "1") || !strcmp(e, "all")) return 1;
  const size_t n = strlen(group);
  for (const char *p = e; *p;) {
    const char *q = strchr(p, ',');
    const size_t len = q ? (size_t)(q - p) : strlen(p);
    if (len == n && !strncmp(p, group, n)) return 1;
    p += len + (q ? 1 : 0);
  }
  return 0;
}

/* Called from uvg_strategy_register_quant / _picture (strategies-quant.c:50-66, strategies-picture.c:62-84) next to the
 * generic / avx2 registrars.  The state-free strategies of the same groups are registered by libuvg266hip.so itself
 * (uvg_strategy_register_quant_hip / _picture_hip).  Returning 0 makes uvg_strategyselector_init fail: with the backend asked
 * for and no gfx950 device there is no silent fall-back to the CPU strategies. */
int uvg_strategy_register_state_hip_quant(void *opaque, uint8_t bitdepth)
{
  bool success = true;
  if (bitdepth != UVG_BIT_DEPTH || uvghip_init(0) != 0) { fprintf(stderr, "hip backend: %s\n", uvghip_last_error()); return 0; }
  success &= uvg_strategyselector_register(opaque, "quant", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_quant_hip);
  success &= uvg_strategyselector_register(opaque, "dequant", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_dequant_hip);
  success &= uvg_strategyselector_register(opaque, "quantize_residual", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_quantize_residual_hip);
  success &= uvg_strategyselector_register(opaque, "quant_cbcr_residual", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_quant_cbcr_residual_hip);
  return success;
}
int uvg_strategy_register_state_hip_picture(void *opaque, uint8_t bitdepth)
{
  if (bitdepth != UVG_BIT_DEPTH || uvghip_init(0) != 0) { fprintf(stderr, "hip backend: %s\n", uvghip_last_error()); return 0; }
  return uvg_strategyselector_register(opaque, "bipred_average", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_inter_recon_bipred_hip);
}
int uvg_strategy_register_state_hip_alf(void *opaque, uint8_t bitdepth)
{
  bool success = true;
  if (bitdepth != UVG_BIT_DEPTH || uvghip_init(0) != 0) { fprintf(stderr, "hip backend: %s\n", uvghip_last_error()); return 0; }
  success &= uvg_strategyselector_register(opaque, "alf_derive_classification_blk", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_alf_derive_classification_blk_hip);
  success &= uvg_strategyselector_register(opaque, "alf_filter_5x5_blk", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_alf_filter_5x5_blk_hip);
  success &= uvg_strategyselector_register(opaque, "alf_filter_7x7_blk", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_alf_filter_7x7_blk_hip);
  success &= uvg_strategyselector_register(opaque, "alf_get_blk_stats", UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, &uvg_alf_get_blk_stats_hip);
  return success;
}
int uvg_strategy_register_state_hip(void *opaque, uint8_t bitdepth)
{
  return uvg_strategy_register_state_hip_quant(opaque, bitdepth) && uvg_strategy_register_state_hip_picture(opaque, bitdepth) &&
         uvg_strategy_register_state_hip_alf(opaque, bitdepth);
}
