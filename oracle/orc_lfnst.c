/*
 * orc_lfnst.c -- low-frequency non-separable transform (LFNST), restated from
 *   src/transform.c:880-917   uvg_fwd_lfnst_NxN  (16 or 48 inputs -> zero_out outputs, (sum + 64) >> 7)
 *   src/transform.c:919-944   get_lfnst_intra_mode, get_transpose_flag
 *   src/transform.c:965-1077  uvg_fwd_lfnst      (gather top-left 4x4 / 8x8-minus-one-quadrant, optional
 *                                                 transpose, kernel, diagonal-scan placement)
 *   src/transform.c:1079-1102 uvg_inv_lfnst_NxN  (result truncated to int16: the cast precedes the clip)
 *   src/transform.c:1104-1225 uvg_inv_lfnst
 *   src/intra.c:637-658       uvg_wide_angle_correction (account_for_dc_planar = true)
 * The kernels (H.266 8.7.4.3) come from orc_lfnst_tables.h, generated from the reference's measured
 * responses (tools/refcheck/dump_lfnst.c); the mode -> set rule is 8.7.4.1 (lfnst_tables.h:51-54 as ranges);
 * the scans are the VVC up-right diagonal scan of 4x4 coefficient groups.
 *
 * The caller resolves which intra mode applies (luma mode, or for chroma the chroma mode / co-located
 * luma mode for CCLM / planar for MIP: transform.c:996-1001) and passes the CU log2 dimensions used
 * for the wide-angle correction (transform.c:1004-1009).  TEST INFRASTRUCTURE ONLY.
 * Depth-independent: exported once (8-bit build).
 */
#include "orc_common.h"
#if ORC_BIT_DEPTH == 8
#include "orc_lfnst_tables.h"

/* up-right diagonal scan of a 4x4 group: position k -> (x, y) */
static void diag4(int pos[16][2])
{
  int k = 0;
  for (int s = 0; s < 7; ++s)
    for (int y = s < 4 ? s : 3; y >= 0 && s - y < 4; --y) { pos[k][0] = s - y; pos[k][1] = y; ++k; }
}

/* first `count` (16 or 48) raster offsets of the LFNST region of a TU `width` wide: coefficient groups
 * (0,0), then for 8x8 regions (0,1) [below] and (1,0) [right] in the diagonal order of groups */
static void lfnst_scan(int width, int big, int *scan)
{
  int p[16][2];
  diag4(p);
  static const int cg[3][2] = {{0, 0}, {0, 1}, {1, 0}};          /* (cgx, cgy) in scan order */
  const int ngroups = big ? 3 : 1;
  for (int g = 0; g < ngroups; ++g)
    for (int k = 0; k < 16; ++k) scan[g * 16 + k] = (cg[g][1] * 4 + p[k][1]) * width + cg[g][0] * 4 + p[k][0];
}

static int lfnst_set_of_mode(int m)   /* lfnst_tables.h:51-54 */
{
  if (m <= 1) return 0;
  if (m <= 12) return 1;
  if (m <= 23) return 2;
  if (m <= 44) return 3;
  if (m <= 55) return 2;
  return 1;
}

static int lfnst_mode(int intra_mode, int log2_w, int log2_h, int *transpose)
{
  int pm = intra_mode;                                    /* intra.c:637-658, account_for_dc_planar */
  if (log2_w != log2_h && intra_mode > 1 && intra_mode <= 66) {
    static const int shift[6] = {0, 6, 10, 12, 14, 15};
    const int d = abs(log2_w - log2_h);
    if (log2_w > log2_h && intra_mode < 2 + shift[d]) pm += 65;
    else if (log2_h > log2_w && intra_mode > 66 - shift[d]) pm -= 67;
  }
  int m = pm < 0 ? pm + 14 + 67 : (pm >= 67 ? pm + 14 : pm);   /* transform.c:919-937 */
  *transpose = (m >= 67 && m >= 81) || (m < 67 && m > 34);      /* transform.c:939-943 */
  return m;
}

/* lfnst_idx 1..2.  coeffs: width x height TU buffer (row-major), modified in place. */
ORC_EXPORT void orc_lfnst_fwd(int16_t *coeffs, int width, int height, int intra_mode, int log2_w, int log2_h, int lfnst_idx)
{
  if (lfnst_idx < 1 || lfnst_idx > 2) return;
  int transpose;
  const int m = lfnst_mode(intra_mode, log2_w, log2_h, &transpose);
  const int big = width >= 8 && height >= 8, sb = big ? 8 : 4;
  const int tr_size = big ? 48 : 16;
  const int zero_out = ((width == 4 && height == 4) || (width == 8 && height == 8)) ? 8 : 16;
  int16_t in[48], out[48];
  int k = 0;
  if (transpose) {
    /* in[x * sb' + y]: columns become rows; for 8x8 the last quadrant is skipped (transform.c:1021-1052) */
    if (sb == 4) { for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) in[x * 4 + y] = coeffs[y * width + x]; }
    else {
      for (int y = 0; y < 8; ++y) {
        for (int x = 0; x < 4; ++x) in[x * 8 + y] = coeffs[y * width + x];
        if (y < 4) for (int x = 4; x < 8; ++x) in[32 + (x - 4) * 4 + y] = coeffs[y * width + x];
      }
    }
  } else {
    for (int y = 0; y < sb; ++y) { const int n = y < 4 ? sb : 4; for (int x = 0; x < n; ++x) in[k++] = coeffs[y * width + x]; }
  }
  const int8_t *M = big ? ORC_LFNST8 + ((size_t)lfnst_set_of_mode(m) * 2 + (lfnst_idx - 1)) * 16 * 48
                        : ORC_LFNST4 + ((size_t)lfnst_set_of_mode(m) * 2 + (lfnst_idx - 1)) * 16 * 16;
  for (int j = 0; j < zero_out; ++j) {
    int acc = 0;
    for (int i = 0; i < tr_size; ++i) acc += in[i] * M[i * 16 + j];        /* tables are input-major: [input][output] */
    out[j] = (int16_t)((acc + 64) >> 7);
  }
  for (int j = zero_out; j < tr_size; ++j) out[j] = 0;
  int scan[48];
  lfnst_scan(width, big, scan);
  for (int j = 0; j < tr_size; ++j) coeffs[scan[j]] = out[j];
}

ORC_EXPORT void orc_lfnst_inv(int16_t *coeffs, int width, int height, int intra_mode, int log2_w, int log2_h, int lfnst_idx)
{
  if (lfnst_idx < 1 || lfnst_idx > 2) return;
  int transpose;
  const int m = lfnst_mode(intra_mode, log2_w, log2_h, &transpose);
  const int big = width >= 8 && height >= 8, sb = big ? 8 : 4;
  const int tr_size = big ? 48 : 16;
  const int zero_out = ((width == 4 && height == 4) || (width == 8 && height == 8)) ? 8 : 16;
  int scan[48];
  lfnst_scan(width, big, scan);
  int16_t in[16], out[48];
  for (int j = 0; j < 16; ++j) in[j] = coeffs[scan[j]];
  const int8_t *M = big ? ORC_LFNST8 + ((size_t)lfnst_set_of_mode(m) * 2 + (lfnst_idx - 1)) * 16 * 48
                        : ORC_LFNST4 + ((size_t)lfnst_set_of_mode(m) * 2 + (lfnst_idx - 1)) * 16 * 16;
  for (int j = 0; j < tr_size; ++j) {
    int acc = 0;
    for (int i = 0; i < zero_out; ++i) acc += in[i] * M[j * 16 + i];        /* transposed kernel: spatial j from coefficient i */
    const int v = (acc + 64) >> 7;
    out[j] = (int16_t)v;    /* transform.c:1098 casts to coeff_t before CLIP(-2^15, 2^15-1): the clip never acts, the value wraps */
  }
  int k = 0;
  if (transpose) {
    if (sb == 4) { for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) coeffs[y * width + x] = out[x * 4 + y]; }
    else {
      for (int y = 0; y < 8; ++y) {
        for (int x = 0; x < 4; ++x) coeffs[y * width + x] = out[x * 8 + y];
        if (y < 4) for (int x = 4; x < 8; ++x) coeffs[y * width + x] = out[32 + (x - 4) * 4 + y];
      }
    }
  } else {
    for (int y = 0; y < sb; ++y) { const int n = y < 4 ? sb : 4; for (int x = 0; x < n; ++x) coeffs[y * width + x] = out[k++]; }
  }
}
#endif
