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

void ORC_FN(rdoq_scans)(int width, int height, uint32_t *scan, uint32_t *scan_cg);

/* uvg_quant with cfg.signhide_enable (quant-generic.c:51-232), lfnst_idx as the reference's argument.  Quirks kept: the
 * lfnst form and every delta_u take their scale from the (flat) scaling-list array = quant_scales[0][qp % 6], the plain form
 * its levels from the sqrt(2)-aware default (:94, :133, :113, :145). */
ORC_EXPORT void ORC_FN(quant_sh)(const int16_t *coef, int16_t *q_coef, int width, int height, int bitdepth, int qp_scaled,
                                 int transform_skip, int slice_is_intra, int lfnst_idx)
{
  const int lw = orc_log2i(width), lh = orc_log2i(height), wh = width * height;
  if (transform_skip && qp_scaled < 4 + 6 * 2) qp_scaled = 4 + 6 * 2;
  const int sqrt2 = !transform_skip && ((lw + lh) & 1);
  const int transform_shift = 15 - bitdepth - ((lw + lh) >> 1) - sqrt2;
  const int64_t q_bits = 14 + qp_scaled / 6 + (transform_skip ? 0 : transform_shift);
  const int32_t add = (slice_is_intra ? 171 : 85) << (q_bits - 9);
  const int32_t q_bits8 = (int32_t)q_bits - 8;
  const int32_t scale = quant_scales[sqrt2][qp_scaled % 6], flat = quant_scales[0][qp_scaled % 6];
  static uint32_t scan[1024], scan_cg[64];
#pragma omp threadprivate(scan, scan_cg)
  ORC_FN(rdoq_scans)(width, height, scan, scan_cg);
  uint32_t ac_sum = 0;
  static int32_t delta_u[1024];
#pragma omp threadprivate(delta_u)
  const int maxn = ((width == 4 && height == 4) || (width == 8 && height == 8)) ? 8 : 16;
  if (lfnst_idx == 0) {
    for (int n = 0; n < wh; ++n) {
      const int32_t c = coef[n];
      const int64_t a = c < 0 ? -(int64_t)c : c;
      int32_t level = (int32_t)((a * scale + add) >> q_bits);
      ac_sum += (uint32_t)level;
      if (c < 0) level = -level;
      q_coef[n] = (int16_t)orc_clip16(level);
    }
  } else {
    for (int n = 0; n < wh; ++n) q_coef[n] = 0;
    for (int n = 0; n < maxn; ++n) {
      const uint32_t idx = scan[n];
      const int32_t c = coef[idx];
      const int64_t a = c < 0 ? -(int64_t)c : c;
      int32_t level = (int32_t)((a * flat + add) >> q_bits);
      ac_sum += (uint32_t)level;
      if (c < 0) level = -level;
      q_coef[idx] = (int16_t)orc_clip16(level);
    }
  }
  if (ac_sum < 2) return;
  if (lfnst_idx == 0) {
    for (int n = 0; n < wh; ++n) {
      const int32_t c = coef[n];
      const int64_t a = c < 0 ? -(int64_t)c : c;
      const int32_t level = (int32_t)((a * flat + add) >> q_bits);
      delta_u[n] = (int32_t)((a * flat - ((int64_t)level << q_bits)) >> q_bits8);
    }
  } else {
    for (int n = 0; n < wh; ++n) delta_u[n] = 0;       /* (the reference leaves the rest uninitialised; it never reads it) */
    for (int n = 0; n < maxn; ++n) {
      const uint32_t idx = scan[n];
      const int32_t c = coef[idx];
      const int64_t a = c < 0 ? -(int64_t)c : c;
      const int32_t level = (int32_t)((a * flat + add) >> q_bits);
      delta_u[idx] = (int32_t)((a * flat - ((int64_t)level << q_bits)) >> q_bits8);
    }
  }
  int last_cg = -1;
  for (int subset = (wh - 1) >> 4; subset >= 0; subset--) {
    int first_nz = 16, last_nz = -1, abssum = 0;
    const int subpos = subset << 4;
    for (int n = 15; n >= 0; n--) if (q_coef[scan[n + subpos]]) { last_nz = n; break; }
    for (int n = 0; n < 16; n++) if (q_coef[scan[n + subpos]]) { first_nz = n; break; }
    for (int n = first_nz; n <= last_nz; n++) abssum += q_coef[scan[n + subpos]];
    if (last_nz >= 0 && last_cg == -1) last_cg = 1;
    if (last_nz - first_nz >= 4) {
      const int signbit = q_coef[scan[subpos + first_nz]] > 0 ? 0 : 1;
      if (signbit != (abssum & 1)) {
        int32_t min_cost = 0x7fffffff, cur_cost = 0x7fffffff;
        int min_pos = -1;
        int16_t final_change = 0, cur_change = 0;
        for (int n = (last_cg == 1 ? last_nz : 15); n >= 0; n--) {
          const uint32_t b = scan[n + subpos];
          if (q_coef[b] != 0) {
            if (delta_u[b] > 0) { cur_cost = -delta_u[b]; cur_change = 1; }
            else if (n == first_nz && abs((int)q_coef[b]) == 1) cur_cost = 0x7fffffff;
            else { cur_cost = delta_u[b]; cur_change = -1; }
          } else if (n < first_nz && ((coef[b] >= 0) ? 0 : 1) != signbit) cur_cost = 0x7fffffff;
          else { cur_cost = -delta_u[b]; cur_change = 1; }
          if (cur_cost < min_cost) { min_cost = cur_cost; final_change = cur_change; min_pos = (int)b; }
        }
        if (q_coef[min_pos] == 32767 || q_coef[min_pos] == -32768) final_change = -1;
        if (coef[min_pos] >= 0) q_coef[min_pos] = (int16_t)(q_coef[min_pos] + final_change);
        else q_coef[min_pos] = (int16_t)(q_coef[min_pos] - final_change);
      }
    }
    if (last_cg == 1) last_cg = 0;
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
