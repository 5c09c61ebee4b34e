/*
 * oracle/orc_alf.c -- restatement of the "alf" strategy group on whole planes:
 *   alf_derive_classification_blk   strategies/generic/alf-generic.c:49-288
 *   alf_filter_7x7_blk / 5x5_blk    strategies/generic/alf-generic.c:290-737
 *   alf_calc_covariance / alf_get_blk_stats  strategies/generic/alf-generic.c:742-999
 * Picture borders: the encoder replicates the border samples into a 4-sample
 * padding before ALF (adjust_pixels, alf.c:937-1113), restated here as
 * coordinate clamping.  Virtual boundary: row vb_pos of every vb_ctu_height
 * rows (luma 60 of 64, chroma 30 of 32; alf.c:5048-5051, alf.h:32-33).
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"

static inline int pxc(const orc_px *p, int stride, int w, int h, int x, int y)
{
  return p[(size_t)orc_clip3(0, h - 1, y) * stride + orc_clip3(0, w - 1, x)];
}

/*
 * Classification of every 4x4 luma block: cls[(y/4)*cls_stride + x/4] = class_idx | transpose_idx << 5.
 * shift = input_bitdepth + 4 (alf.c:5185).
 */
ORC_EXPORT void ORC_FN(alf_classify_frame)(const orc_px *rec, int stride, int w, int h, int shift, int vbh, int vb_pos,
                                           uint8_t *cls, int cls_stride)
{
  static const int th[16] = {0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4};
  static const int transpose_table[8] = {0, 1, 0, 2, 2, 3, 1, 3};
#pragma omp parallel for schedule(static)
  for (int by = 0; by < h; by += 4)
    for (int bx = 0; bx < w; bx += 4) {
      /* quad Laplacians on the 2x-subsampled 8x8 window around the block: rows by-2+2r, columns bx-2+2c */
      int sum[4] = {0, 0, 0, 0};
      const int ymod = by % vbh;
      for (int r = 0; r < 4; ++r) {
        if (ymod == vb_pos - 4 && r == 3) continue;     /* alf-generic.c:183-189 */
        if (ymod == vb_pos && r == 0) continue;         /* :190-196 */
        const int y = by - 2 + 2 * r;
        /* row substitutions at the virtual boundary (:96-103); y is also the block-relative dst row */
        int y_up2 = y + 2, y_dn = y - 1;
        if (y > 0 && (y & (vbh - 1)) == vb_pos - 2) y_up2 = y + 1;
        else if (y > 0 && (y & (vbh - 1)) == vb_pos) y_dn = y;
        for (int c = 0; c < 4; ++c) {
          const int x = bx - 2 + 2 * c;
#define P(dx, yy) pxc(rec, stride, w, h, x + (dx), (yy))
          const int y0 = P(0, y) << 1, y1 = P(1, y + 1) << 1;
          sum[0] += orc_iabs(y0 - P(0, y_dn) - P(0, y + 1)) + orc_iabs(y1 - P(1, y) - P(1, y_up2));            /* ver */
          sum[1] += orc_iabs(y0 - P(1, y) - P(-1, y)) + orc_iabs(y1 - P(2, y + 1) - P(0, y + 1));               /* hor */
          sum[2] += orc_iabs(y0 - P(-1, y_dn) - P(1, y + 1)) + orc_iabs(y1 - P(0, y) - P(2, y_up2));            /* diag0 */
          sum[3] += orc_iabs(y0 - P(-1, y + 1) - P(1, y_dn)) + orc_iabs(y1 - P(0, y_up2) - P(2, y));            /* diag1 */
#undef P
        }
      }
      const int sv = sum[0], sh = sum[1], sd0 = sum[2], sd1 = sum[3];
      const int yv = by & (vbh - 1);
      const int act = orc_clip3(0, 15, ((sv + sh) * ((yv == vb_pos - 4 || yv == vb_pos) ? 96 : 64)) >> shift);
      int class_idx = th[act];
      int hv1, hv0, d1, d0, dir_hv, dir_d, hvd1, hvd0, main_dir, sec_dir;
      if (sv > sh) { hv1 = sv; hv0 = sh; dir_hv = 1; } else { hv1 = sh; hv0 = sv; dir_hv = 3; }
      if (sd0 > sd1) { d1 = sd0; d0 = sd1; dir_d = 0; } else { d1 = sd1; d0 = sd0; dir_d = 2; }
      if ((uint32_t)d1 * (uint32_t)hv0 > (uint32_t)hv1 * (uint32_t)d0) { hvd1 = d1; hvd0 = d0; main_dir = dir_d; sec_dir = dir_hv; }
      else { hvd1 = hv1; hvd0 = hv0; main_dir = dir_hv; sec_dir = dir_d; }
      int strength = 0;
      if (hvd1 > 2 * hvd0) strength = 1;
      if (hvd1 * 2 > 9 * hvd0) strength = 2;
      if (strength) class_idx += (((main_dir & 1) << 1) + strength) * 5;
      const int tr = transpose_table[main_dir * 2 + (sec_dir >> 1)];
      cls[(by >> 2) * cls_stride + (bx >> 2)] = (uint8_t)(class_idx | (tr << 5));
    }
}

static const int8_t k_perm7[4][13] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12}, {9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6, 12},
                                      {0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11, 12}, {9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6, 12}};

static inline int clip_pair(int clip, int ref, int v0, int v1)
{
  return orc_clip3(-clip, clip, v0 - ref) + orc_clip3(-clip, clip, v1 - ref);
}

/*
 * Filter the rectangle (x0,y0,w,h) of `src` into `dst` (alf-generic.c:290-737).  Luma (chroma = 0):
 * 7x7 diamond, coefficients/clips chosen per 4x4 block from cls (25 classes x 13, transposed by
 * transpose_idx).  Chroma: 5x5 diamond, one 7-entry set.  pic_w/pic_h bound the clamped reads.
 */
ORC_EXPORT void ORC_FN(alf_filter_rect)(const orc_px *src, orc_px *dst, int stride, int pic_w, int pic_h, int x0, int y0,
                                        int w, int h, int chroma, const int16_t *coef_set, const int16_t *clip_set,
                                        const uint8_t *cls, int cls_stride, int vbh, int vb_pos)
{
  const int shift = ORC_BIT_DEPTH - 1, offset = 1 << (shift - 1);
  for (int y = y0; y < y0 + h; ++y) {
    const int y_vb = y & (vbh - 1);
    /* row offsets of the six neighbour rows, clamped towards the centre near the virtual boundary (:600-622) */
    int up1 = 1, up2 = 2, up3 = 3;   /* rows y+1.. (p_img_1,3,5) */
    int dn1 = 1, dn2 = 2, dn3 = 3;   /* rows y-1.. (p_img_2,4,6) */
    if (y_vb < vb_pos && y_vb >= vb_pos - (chroma ? 2 : 4)) {
      const int lim = vb_pos - 1 - y_vb;           /* rows available below before the boundary */
      up1 = up1 > lim ? lim : up1; up2 = up2 > lim ? lim : up2; up3 = up3 > lim ? lim : up3;
      dn1 = dn1 > lim ? lim : dn1; dn2 = dn2 > lim ? lim : dn2; dn3 = dn3 > lim ? lim : dn3;
    } else if (y_vb >= vb_pos && y_vb <= vb_pos + (chroma ? 1 : 3)) {
      const int lim = y_vb - vb_pos;               /* rows available above after the boundary */
      up1 = up1 > lim ? lim : up1; up2 = up2 > lim ? lim : up2; up3 = up3 > lim ? lim : up3;
      dn1 = dn1 > lim ? lim : dn1; dn2 = dn2 > lim ? lim : dn2; dn3 = dn3 > lim ? lim : dn3;
    }
    const int near_vb = (y_vb == vb_pos - 1) || (y_vb == vb_pos);
    for (int x = x0; x < x0 + w; ++x) {
      int f[13], c[13];
      if (!chroma) {
        const int cl = cls[(y >> 2) * cls_stride + (x >> 2)];
        const int16_t *cf = coef_set + (cl & 31) * 13, *cc = clip_set + (cl & 31) * 13;
        const int8_t *perm = k_perm7[cl >> 5];
        for (int k = 0; k < 13; ++k) { f[k] = cf[perm[k]]; c[k] = cc[perm[k]]; }
      } else {
        for (int k = 0; k < 7; ++k) { f[k] = coef_set[k]; c[k] = clip_set[k]; }
      }
#define S(dx, dy) pxc(src, stride, pic_w, pic_h, x + (dx), y + (dy))
      const int cur = S(0, 0);
      int sum = 0;
      if (!chroma) {
        sum += f[0] * clip_pair(c[0], cur, S(0, up3), S(0, -dn3));
        sum += f[1] * clip_pair(c[1], cur, S(1, up2), S(-1, -dn2));
        sum += f[2] * clip_pair(c[2], cur, S(0, up2), S(0, -dn2));
        sum += f[3] * clip_pair(c[3], cur, S(-1, up2), S(1, -dn2));
        sum += f[4] * clip_pair(c[4], cur, S(2, up1), S(-2, -dn1));
        sum += f[5] * clip_pair(c[5], cur, S(1, up1), S(-1, -dn1));
        sum += f[6] * clip_pair(c[6], cur, S(0, up1), S(0, -dn1));
        sum += f[7] * clip_pair(c[7], cur, S(-1, up1), S(1, -dn1));
        sum += f[8] * clip_pair(c[8], cur, S(-2, up1), S(2, -dn1));
        sum += f[9] * clip_pair(c[9], cur, S(3, 0), S(-3, 0));
        sum += f[10] * clip_pair(c[10], cur, S(2, 0), S(-2, 0));
        sum += f[11] * clip_pair(c[11], cur, S(1, 0), S(-1, 0));
      } else {
        sum += f[0] * clip_pair(c[0], cur, S(0, up2), S(0, -dn2));
        sum += f[1] * clip_pair(c[1], cur, S(1, up1), S(-1, -dn1));
        sum += f[2] * clip_pair(c[2], cur, S(0, up1), S(0, -dn1));
        sum += f[3] * clip_pair(c[3], cur, S(-1, up1), S(1, -dn1));
        sum += f[4] * clip_pair(c[4], cur, S(2, 0), S(-2, 0));
        sum += f[5] * clip_pair(c[5], cur, S(1, 0), S(-1, 0));
      }
#undef S
      sum = near_vb ? (sum + (1 << (shift + 2))) >> (shift + 3) : (sum + offset) >> shift;
      dst[(size_t)y * stride + x] = orc_clip_px(sum + cur);
    }
  }
}

/* e_local of one sample (alf-generic.c:742-905): e[k][b], k < 13 (luma) or 7 (chroma) */
static void covariance_sample(int e[13][4], const orc_px *rec, int stride, int pic_w, int pic_h, int x, int y, int chroma,
                              int transpose_idx, int vb_distance, const int16_t clip[4])
{
  static const int pat5[13] = {0, 1, 2, 3, 4, 5, 6, 5, 4, 3, 2, 1, 0};
  static const int pat7[25] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0};
  int top = -4, bot = 4;
  if (vb_distance >= -3 && vb_distance < 0) { bot = -vb_distance - 1; top = -bot; }
  else if (vb_distance >= 0 && vb_distance < 3) { top = -vb_distance; bot = -top; }
  const int *pat = chroma ? pat5 : pat7;
  const int half = chroma ? 2 : 3;
  const int cur = pxc(rec, stride, pic_w, pic_h, x, y);
  int k = 0;
#define R(dx, dy) pxc(rec, stride, pic_w, pic_h, x + (dx), y + (dy))
#define ROWP(i) ((i) > top ? (i) : top)           /* MAX(i, clip_top_row) */
#define ROWN(i) (-((i) > -bot ? (i) : -bot))      /* -MAX(i, -clip_bot_row) */
  if (transpose_idx == 0 || transpose_idx == 2) {
    for (int i = -half; i < 0; ++i) {
      if (transpose_idx == 0)
        for (int j = -half - i; j <= half + i; ++j, ++k)
          for (int b = 0; b < 4; ++b) e[pat[k]][b] += clip_pair(clip[b], cur, R(j, ROWP(i)), R(-j, ROWN(i)));
      else
        for (int j = half + i; j >= -half - i; --j, ++k)
          for (int b = 0; b < 4; ++b) e[pat[k]][b] += clip_pair(clip[b], cur, R(j, ROWP(i)), R(-j, ROWN(i)));
    }
    for (int j = -half; j < 0; ++j, ++k)
      for (int b = 0; b < 4; ++b) e[pat[k]][b] += clip_pair(clip[b], cur, R(j, 0), R(-j, 0));
  } else {
    for (int j = -half; j < 0; ++j) {
      if (transpose_idx == 1)
        for (int i = -half - j; i <= half + j; ++i, ++k)
          for (int b = 0; b < 4; ++b) e[pat[k]][b] += clip_pair(clip[b], cur, R(j, ROWP(i)), R(-j, ROWN(i)));
      else
        for (int i = half + j; i >= -half - j; --i, ++k)
          for (int b = 0; b < 4; ++b) e[pat[k]][b] += clip_pair(clip[b], cur, R(j, ROWP(i)), R(-j, ROWN(i)));
    }
    for (int i = -half; i < 0; ++i, ++k)
      for (int b = 0; b < 4; ++b) e[pat[k]][b] += clip_pair(clip[b], cur, R(0, ROWP(i)), R(0, ROWN(i)));
  }
#undef R
#undef ROWP
#undef ROWN
  for (int b = 0; b < 4; ++b) e[pat[k]][b] += cur;
}

/*
 * Covariance statistics of one rectangle (alf-generic.c:908-999), accumulated per class:
 * for class c: ee[c][13][13][4][4] (int64, full symmetric matrix as the reference leaves it),
 * yv[c][13][4] (int32), pix[c] (int64; the reference keeps the same integer in a double).
 * Luma: 25 classes from cls; chroma: 1 class, 7 coefficients (rows/cols 7..12 stay zero).
 */
ORC_EXPORT void ORC_FN(alf_stats_rect)(const orc_px *org, int org_stride, const orc_px *rec, int rec_stride, int pic_w, int pic_h,
                                       int x0, int y0, int w, int h, int chroma, const uint8_t *cls, int cls_stride,
                                       int vbh, int vb_pos, const int16_t clip[4], int64_t *ee, int32_t *yv, int64_t *pix)
{
  const int nc = chroma ? 7 : 13, ncls = chroma ? 1 : 25;
  memset(ee, 0, sizeof(int64_t) * (size_t)ncls * 13 * 13 * 16);
  memset(yv, 0, sizeof(int32_t) * (size_t)ncls * 13 * 4);
  memset(pix, 0, sizeof(int64_t) * (size_t)ncls);
  for (int y = y0; y < y0 + h; ++y) {
    const int vb_distance = (y % vbh) - vb_pos;
    for (int x = x0; x < x0 + w; ++x) {
      int e[13][4];
      memset(e, 0, sizeof e);
      int tr = 0, ci = 0;
      if (!chroma) { const int cl = cls[(y >> 2) * cls_stride + (x >> 2)]; ci = cl & 31; tr = cl >> 5; }
      const int yl = (int)org[(size_t)y * org_stride + x] - (int)rec[(size_t)y * rec_stride + x];
      covariance_sample(e, rec, rec_stride, pic_w, pic_h, x, y, chroma, tr, vb_distance, clip);
      int64_t *E = ee + (size_t)ci * 13 * 13 * 16;
      for (int k = 0; k < nc; ++k) {
        for (int l = k; l < nc; ++l)
          for (int b0 = 0; b0 < 4; ++b0)
            for (int b1 = 0; b1 < 4; ++b1)
              E[((k * 13 + l) * 4 + b0) * 4 + b1] += (int32_t)((int16_t)e[k][b0] * (double)(int16_t)e[l][b1]);
        for (int b = 0; b < 4; ++b) yv[(ci * 13 + k) * 4 + b] += (int32_t)((int16_t)e[k][b] * (double)(int16_t)yl);
      }
      pix[ci] += (int64_t)yl * yl;
    }
  }
  for (int ci = 0; ci < ncls; ++ci) {
    int64_t *E = ee + (size_t)ci * 13 * 13 * 16;
    for (int k = 1; k < nc; ++k)
      for (int l = 0; l < k; ++l)
        for (int b0 = 0; b0 < 4; ++b0)
          for (int b1 = 0; b1 < 4; ++b1) E[((k * 13 + l) * 4 + b0) * 4 + b1] = E[((l * 13 + k) * 4 + b1) * 4 + b0];
  }
}
