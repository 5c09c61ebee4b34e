/*
 * oracle/orc_quant.c -- restatement of the "quant" strategy group on plain
 * parameters (the reference reads them from encoder_state_t):
 *   quant            strategies/generic/quant-generic.c:51-121 (scaling list off, lfnst_idx 0,
 *                    sign hiding off -- :123-233 is not restated; signhide=0 in every preset used)
 *   dequant          :618-669 (no scaling list, no dep-quant)
 *   quantize_residual :460-612 restricted to the plain-quant branch (:532-536): RDOQ / dep-quant /
 *                    TS-RDOQ are serial entropy-coupled code outside the hot-path scope
 *   coeff_abs_sum    :671-678      fast_coeff_cost :688-705
 *   uvg_get_scaled_qp  transform.c:150-165 (chroma map passed in by the caller)
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"

static const int16_t quant_scales[2][6] = {     /* scalinglist.c:91-94 (H.266 levelScale inverse) */
  {26214, 23302, 20560, 18396, 16384, 14564}, {18396, 16384, 14564, 13107, 11651, 10280}};
static const int16_t inv_quant_scales[2][6] = { /* scalinglist.c:95-98 (H.266 levelScale) */
  {40, 45, 51, 57, 64, 72}, {57, 64, 72, 80, 90, 102}};

void ORC_FN(generate_residual)(const orc_px *, const orc_px *, int16_t *, int, int, int, int);
void ORC_FN(tr_forward)(int, int, int, int, int, int, int, const int16_t *, int16_t *);
void ORC_FN(tr_inverse)(int, int, int, int, int, int, int, const int16_t *, int16_t *);

/* transform.c:150-165; chroma_map may be NULL */
ORC_EXPORT int ORC_FN(get_scaled_qp)(int color, int qp, int qp_offset, const int8_t *chroma_map)
{
  if (color == 0) return qp + qp_offset;
  if (chroma_map) return chroma_map[qp] + qp_offset;
  return orc_clip3(-qp_offset, 57, qp) + qp_offset;
}

/* quant-generic.c:51-121.  qp_scaled: output of get_scaled_qp (before the transform-skip floor). */
ORC_EXPORT void ORC_FN(quant)(const int16_t *coef, int16_t *q_coef, int width, int height, int bitdepth,
                              int qp_scaled, int transform_skip, int slice_is_intra)
{
  const int lw = orc_log2i(width), lh = orc_log2i(height);
  if (transform_skip && qp_scaled < 4 + 6 * 2) qp_scaled = 4 + 6 * 2;       /* MIN_QP_PRIME_TS = 2 */
  const int sqrt2 = !transform_skip && ((lw + lh) & 1);
  const int transform_shift = 15 - bitdepth - ((lw + lh) >> 1) - sqrt2;
  const int64_t q_bits = 14 + qp_scaled / 6 + (transform_skip ? 0 : transform_shift);
  const int32_t add = (slice_is_intra ? 171 : 85) << (q_bits - 9);
  const int32_t scale = quant_scales[sqrt2][qp_scaled % 6];
  for (int n = 0; n < width * height; ++n) {
    const int32_t c = coef[n];
    const int64_t a = c < 0 ? -(int64_t)c : c;
    int32_t level = (int32_t)((a * scale + add) >> q_bits);
    if (c < 0) level = -level;
    q_coef[n] = (int16_t)orc_clip16(level);
  }
}

/* quant-generic.c:618-669 */
ORC_EXPORT void ORC_FN(dequant)(const int16_t *q_coef, int16_t *coef, int width, int height, int bitdepth,
                                int qp_scaled, int transform_skip)
{
  const int lw = orc_log2i(width), lh = orc_log2i(height);
  const int transform_shift = 15 - bitdepth - ((lw + lh) >> 1);
  const int sqrt2 = !transform_skip && ((lw + lh) & 1);
  if (transform_skip && qp_scaled < 4 + 6 * 2) qp_scaled = 4 + 6 * 2;
  const int shift = 20 - 14 - (transform_skip ? 0 : transform_shift - sqrt2);
  const int32_t scale = inv_quant_scales[sqrt2][qp_scaled % 6] << (qp_scaled / 6);
  const int32_t add = 1 << (shift - 1);
  for (int n = 0; n < width * height; ++n)
    coef[n] = (int16_t)orc_clip16((q_coef[n] * scale + add) >> shift);
}

ORC_EXPORT uint32_t ORC_FN(coeff_abs_sum)(const int16_t *c, size_t n)
{
  uint32_t s = 0;
  for (size_t i = 0; i < n; ++i) s += (uint32_t)orc_iabs(c[i]);
  return s;
}

ORC_EXPORT uint32_t ORC_FN(fast_coeff_cost)(const int16_t *c, int width, int height, uint64_t weights)
{
  uint32_t s = 0;
  for (int i = 0; i < width * height; ++i) {
    int a = orc_iabs(c[i]);
    if (a > 3) a = 3;
    s += (uint32_t)((weights >> (16 * a)) & 0xffff);
  }
  return (s + 128) >> 8;
}

/*
 * TU round trip, plain-quant branch of uvg_quantize_residual_generic
 * (quant-generic.c:460-612): residual -> forward transform -> quant ->
 * has_coeffs -> dequant -> inverse transform -> clip(pred + residual).
 * Transform kernel selection is explicit (hor/ver/skips as returned by
 * get_tr_type/mts_skips).  rec may alias pred.  Returns has_coeffs.
 */
ORC_EXPORT int ORC_FN(tu_roundtrip)(int bitdepth, int hor, int ver, int skip_w, int skip_h,
                                    int width, int height, int qp_scaled, int slice_is_intra,
                                    const orc_px *ref_in, const orc_px *pred_in, int in_stride,
                                    orc_px *rec_out, int out_stride, int16_t *coeff_out)
{
  int16_t residual[32 * 32], coeff[32 * 32];
  ORC_FN(generate_residual)(ref_in, pred_in, residual, width, height, in_stride, in_stride);
  ORC_FN(tr_forward)(bitdepth, hor, ver, width, height, skip_w, skip_h, residual, coeff);
  ORC_FN(quant)(coeff, coeff_out, width, height, bitdepth, qp_scaled, 0, slice_is_intra);
  int has = 0;
  for (int i = 0; i < width * height; ++i) if (coeff_out[i]) { has = 1; break; }
  if (has) {
    ORC_FN(dequant)(coeff_out, coeff, width, height, bitdepth, qp_scaled, 0);
    ORC_FN(tr_inverse)(bitdepth, hor, ver, width, height, skip_w, skip_h, coeff, residual);
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        const int16_t v = (int16_t)(residual[y * width + x] + pred_in[y * in_stride + x]);
        rec_out[y * out_stride + x] = orc_clip_px(v);
      }
  } else if (rec_out != pred_in) {
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) rec_out[y * out_stride + x] = pred_in[y * in_stride + x];
  }
  return has;
}

/* Frame-level driver (bench cpu_baseline / tests): TU round trip of n_tus w x h TUs, OpenMP over TUs. */
ORC_EXPORT void ORC_FN(tu_roundtrip_frame)(int bitdepth, int width, int height, int qp_scaled, int slice_is_intra,
                                           const orc_px *orig, const orc_px *pred, orc_px *rec, int stride,
                                           const int32_t *tus, int n_tus, int16_t *coeff_out)
{
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < n_tus; ++i) {
    const size_t off = (size_t)tus[2 * i + 1] * stride + tus[2 * i];
    ORC_FN(tu_roundtrip)(bitdepth, 0, 0, 0, 0, width, height, qp_scaled, slice_is_intra, orig + off, pred + off, stride,
                         rec + off, stride, coeff_out + (size_t)i * width * height);
  }
}
