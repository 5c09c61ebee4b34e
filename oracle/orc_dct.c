/*
 * oracle/orc_dct.c -- restatement of the "dct" strategy group: square DCT-2
 * (dct_NxN / idct_NxN), the MTS front end (mts_dct / mts_idct with DCT-2,
 * DST-7, DCT-8, non-square blocks and zero-out) and the dead 4x4 DST.
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 *
 * Reference followed (/root/reference/src/strategies/generic/dct-generic.c):
 *   :396-446   partial_butterfly_(inverse_)N: fwd truncates to int16, inv clips
 *   :720-740   dct/idct shifts: fwd log2N-1+(depth-8) then log2N+6; inv 7 then 20-depth
 *   :359-393,752-770  fast_forward/inverse_dst_4x4
 *   :1030-2370 fastForward/InverseXXX_Bn: 1-D kernels; the butterflies are exact
 *              integer factorisations of the matrix products written here
 *              (no intermediate rounding), so a plain product gives identical
 *              bits.  What is NOT uniform is how each kernel treats
 *              skip_line2 ("cutoff"); that is restated case by case below.
 *   :2501-2557 uvg_get_tr_type
 *   :2560-2678 mts_dct_generic / mts_idct_generic
 */
#include "orc_common.h"
#include "orc_tables.h"

enum { TR_DCT2 = 0, TR_DCT8 = 1, TR_DST7 = 2 };            /* uvg266.h:235-237 */
enum { MTS_OFF = 0, MTS_INTRA = 1, MTS_INTER = 2, MTS_BOTH = 3, MTS_IMPLICIT = 4 }; /* uvg266.h:225-229 */

static const int16_t *tr_matrix(int type, int n)
{
  switch (type * 64 + n) {
    case TR_DCT2 * 64 + 2:  return ORC_T_DCT2_2;
    case TR_DCT2 * 64 + 4:  return ORC_T_DCT2_4;
    case TR_DCT2 * 64 + 8:  return ORC_T_DCT2_8;
    case TR_DCT2 * 64 + 16: return ORC_T_DCT2_16;
    case TR_DCT2 * 64 + 32: return ORC_T_DCT2_32;
    case TR_DCT8 * 64 + 4:  return ORC_T_DCT8_4;
    case TR_DCT8 * 64 + 8:  return ORC_T_DCT8_8;
    case TR_DCT8 * 64 + 16: return ORC_T_DCT8_16;
    case TR_DCT8 * 64 + 32: return ORC_T_DCT8_32;
    case TR_DST7 * 64 + 4:  return ORC_T_DST7_4;
    case TR_DST7 * 64 + 8:  return ORC_T_DST7_8;
    case TR_DST7 * 64 + 16: return ORC_T_DST7_16;
    case TR_DST7 * 64 + 32: return ORC_T_DST7_32;
  }
  return NULL;
}

/* Which kernels honour skip_line2:
 *   forward: DST7/DCT8 of size 8,16,32 zero the rows >= cutoff (dct-generic.c:
 *            1651-1655,1768-1772,2014-2018,2139-2143,2264-2268,2335-2339);
 *            DCT2 (all sizes) and the 4-point DST7/DCT8 ignore it.
 *   inverse: only the 8-point DST7/DCT8 stop the sum at cutoff (:2280-2291,
 *            2351-2362); everything else sums all n inputs. */
static int fwd_uses_cutoff(int type, int n) { return type != TR_DCT2 && n >= 8; }
static int inv_uses_cutoff(int type, int n) { return type != TR_DCT2 && n == 8; }

/* 1-D forward over `line` rows of n samples: dst[j*line + i] (transposed). */
static void fwd_1d(int type, int n, const int16_t *src, int16_t *dst, int shift, int line,
                   int skip_line, int skip_line2)
{
  const int16_t *T = tr_matrix(type, n);
  const int add = shift > 0 ? 1 << (shift - 1) : 0;
  const int reduced = line - skip_line;
  const int cutoff = fwd_uses_cutoff(type, n) ? n - skip_line2 : n;
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < line; ++i) {
      int16_t v = 0;
      if (j < cutoff && i < reduced) {
        int32_t acc = 0;
        for (int k = 0; k < n; ++k) acc += (int32_t)T[j * n + k] * src[i * n + k];
        v = (int16_t)((acc + add) >> shift);          /* truncation, not clipping */
      }
      dst[j * line + i] = v;   /* every kernel writes (value or zero) the whole n x line area */
    }
  }
}

/* 1-D inverse: for each of `line` columns i, dst[i*n + j] = clip16(sum_k src[k*line+i] T[k][j]). */
static void inv_1d(int type, int n, const int16_t *src, int16_t *dst, int shift, int line,
                   int skip_line, int skip_line2)
{
  const int16_t *T = tr_matrix(type, n);
  const int add = 1 << (shift - 1);
  const int reduced = line - skip_line;
  const int kmax = inv_uses_cutoff(type, n) ? n - skip_line2 : n;
  for (int i = 0; i < line; ++i) {
    for (int j = 0; j < n; ++j) {
      int16_t v = 0;
      if (i < reduced) {
        int32_t acc = 0;
        for (int k = 0; k < kmax; ++k) acc += (int32_t)src[k * line + i] * T[k * n + j];
        v = (int16_t)orc_clip16((acc + add) >> shift);
      }
      dst[i * n + j] = v;
    }
  }
}

/* dct_NxN (dct-generic.c:720-729): two forward passes */
ORC_EXPORT void ORC_FN(dct_nxn)(int bitdepth, int n, const int16_t *in, int16_t *out)
{
  int16_t tmp[32 * 32];
  const int lg = orc_log2i(n);
  fwd_1d(TR_DCT2, n, in, tmp, lg - 1 + (bitdepth - 8), n, 0, 0);
  fwd_1d(TR_DCT2, n, tmp, out, lg + 6, n, 0, 0);
}

/* idct_NxN (dct-generic.c:731-740) */
ORC_EXPORT void ORC_FN(idct_nxn)(int bitdepth, int n, const int16_t *in, int16_t *out)
{
  int16_t tmp[32 * 32];
  inv_1d(TR_DCT2, n, in, tmp, 7, n, 0, 0);
  inv_1d(TR_DCT2, n, tmp, out, 12 - (bitdepth - 8), n, 0, 0);
}

/* HEVC-style 4x4 DST (dct-generic.c:35-44 matrix, :359-393, :752-770). Dead code upstream. */
static const int16_t dst4_mat[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
ORC_EXPORT void ORC_FN(fast_forward_dst_4x4)(int bitdepth, const int16_t *in, int16_t *out)
{
  int16_t tmp[16];
  const int s[2] = {1 + (bitdepth - 8), 8};
  const int16_t *src = in; int16_t *dst = tmp;
  for (int pass = 0; pass < 2; ++pass) {
    const int rnd = 1 << (s[pass] - 1);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        int32_t acc = 0;
        for (int k = 0; k < 4; ++k) acc += (int32_t)dst4_mat[j * 4 + k] * src[i * 4 + k];
        dst[j * 4 + i] = (int16_t)((acc + rnd) >> s[pass]);
      }
    src = tmp; dst = out;
  }
}
ORC_EXPORT void ORC_FN(fast_inverse_dst_4x4)(int bitdepth, const int16_t *in, int16_t *out)
{
  int16_t tmp[16];
  const int s[2] = {7, 12 - (bitdepth - 8)};
  const int16_t *src = in; int16_t *dst = tmp;
  for (int pass = 0; pass < 2; ++pass) {
    const int rnd = 1 << (s[pass] - 1);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        int32_t acc = 0;
        for (int k = 0; k < 4; ++k) acc += (int32_t)src[k * 4 + i] * dst4_mat[k * 4 + j];
        dst[i * 4 + j] = (int16_t)orc_clip16((acc + rnd) >> s[pass]);
      }
    src = tmp; dst = out;
  }
}

/* uvg_get_tr_type (dct-generic.c:2501-2557) on plain arguments.
 * color: 0 = luma.  cu_intra: tu->type == CU_INTRA, cu_inter: == CU_INTER. */
ORC_EXPORT void ORC_FN(get_tr_type)(int width, int height, int color, int cu_intra, int cu_inter,
                                    int isp_mode, int lfnst_idx, int cr_lfnst_idx, int tr_idx,
                                    int mts_type, int *hor, int *ver)
{
  *hor = TR_DCT2; *ver = TR_DCT2;
  if (color != 0) return;
  const int explicit_mts = mts_type == MTS_BOTH ||
      (cu_intra ? mts_type == MTS_INTRA : (mts_type == MTS_INTER && cu_inter));
  const int implicit_mts = cu_intra && (mts_type == MTS_IMPLICIT || mts_type == MTS_INTER);
  const int is_isp = cu_intra && isp_mode ? isp_mode : 0;
  const int lf = color == 0 ? lfnst_idx : cr_lfnst_idx;
  if (is_isp && lf) return;
  if (implicit_mts || (is_isp && explicit_mts)) {
    if (width >= 4 && width <= 16) *hor = TR_DST7;
    if (height >= 4 && height <= 16) *ver = TR_DST7;
    return;
  }
  if (explicit_mts && tr_idx > 1) {
    static const int subset[4][2] = {{TR_DST7, TR_DST7}, {TR_DCT8, TR_DST7}, {TR_DST7, TR_DCT8}, {TR_DCT8, TR_DCT8}};
    *hor = subset[tr_idx - 2][0];
    *ver = subset[tr_idx - 2][1];
  }
}

/* skip_width / skip_height of mts_(i)dct_generic (dct-generic.c:2582-2600) */
ORC_EXPORT void ORC_FN(mts_skips)(int width, int height, int hor, int ver, int lfnst_active,
                                  int *skip_w, int *skip_h)
{
  *skip_w = (hor != TR_DCT2 && width == 32) ? 16 : (width > 32 ? width - 32 : 0);
  *skip_h = (ver != TR_DCT2 && height == 32) ? 16 : (height > 32 ? height - 32 : 0);
  if (lfnst_active) {
    if ((width == 4 && height > 4) || (width > 4 && height == 4)) { *skip_w = width - 4; *skip_h = height - 4; }
    else if (width >= 8 && height >= 8) { *skip_w = width - 8; *skip_h = height - 8; }
  }
}

/* The two-pass cores of mts_dct_generic / mts_idct_generic (the `else` branch,
 * dct-generic.c:2602-2617 and :2654-2676) for width, height in {1,2,4,8,16,32} (1- and 2-point lines: DCT-2 only). */
ORC_EXPORT void ORC_FN(tr_forward)(int bitdepth, int hor, int ver, int width, int height,
                                   int skip_w, int skip_h, const int16_t *in, int16_t *out)
{
  int16_t tmp[32 * 32];
  const int s1 = orc_log2i(width) - 1 + bitdepth - 8;
  const int s2 = orc_log2i(height) - 1 + 7;
  if (height == 1) { fwd_1d(hor, width, in, out, s1, 1, 0, skip_w); return; }                        /* dct-generic.c:2608-2609 */
  if (width == 1) { fwd_1d(ver, height, in, out, orc_log2i(height) - 1 + 1 + bitdepth + 6 - 15, 1, 0, skip_h); return; }   /* :2610-2612 */
  fwd_1d(hor, width, in, tmp, s1, height, 0, skip_w);
  fwd_1d(ver, height, tmp, out, s2, width, skip_w, skip_h);
}

ORC_EXPORT void ORC_FN(tr_inverse)(int bitdepth, int hor, int ver, int width, int height,
                                   int skip_w, int skip_h, const int16_t *in, int16_t *out)
{
  int16_t tmp[32 * 32];
  const int s1 = 7, s2 = 20 - bitdepth;
  if (height == 1) { inv_1d(hor, width, in, out, s2 + 1, 1, 0, skip_w); return; }                     /* dct-generic.c:2669-2670 */
  if (width == 1) { inv_1d(ver, height, in, out, s2 + 1, 1, 0, skip_h); return; }                      /* :2671-2672 */
  inv_1d(ver, height, in, tmp, s1, width, skip_w, skip_h);
  inv_1d(hor, width, tmp, out, s2, height, 0, skip_w);
}

/* mts_dct_generic / mts_idct_generic (dct-generic.c:2560,2621) */
ORC_EXPORT void ORC_FN(mts_dct)(int bitdepth, int color, int cu_intra, int cu_inter, int isp_mode,
                                int lfnst_idx, int cr_lfnst_idx, int tr_idx, int width, int height,
                                const int16_t *in, int16_t *out, int mts_type, int inverse)
{
  int hor, ver, sw, sh;
  ORC_FN(get_tr_type)(width, height, color, cu_intra, cu_inter, isp_mode, lfnst_idx, cr_lfnst_idx,
                      tr_idx, mts_type, &hor, &ver);
  if (hor == TR_DCT2 && ver == TR_DCT2 && !lfnst_idx && !cr_lfnst_idx && width == height) {
    if (inverse) ORC_FN(idct_nxn)(bitdepth, width, in, out);
    else ORC_FN(dct_nxn)(bitdepth, width, in, out);
    return;
  }
  const int lf_active = (lfnst_idx && color == 0) || (cr_lfnst_idx && color != 0);
  ORC_FN(mts_skips)(width, height, hor, ver, lf_active, &sw, &sh);
  if (inverse) ORC_FN(tr_inverse)(bitdepth, hor, ver, width, height, sw, sh, in, out);
  else ORC_FN(tr_forward)(bitdepth, hor, ver, width, height, sw, sh, in, out);
}
