/*
 * oracle/orc_intra.c -- restatement of uvg266's intra prediction path:
 *   reference sample construction       intra.c:756-1063 (any), :1065-1342 (inner)
 *   [1 2 1]/4 reference smoothing       intra.c:190-225
 *   DC prediction                       intra.c:236-273
 *   wide-angle mapping + ref selection  intra.c:637-753 (intra_predict_regular)
 *   angular prediction                  strategies/generic/intra-generic.c:55-295
 *   planar                              strategies/generic/intra-generic.c:306-361
 *   PDPC for planar/DC                  strategies/generic/intra-generic.c:414-437
 *   rough-search cost min(SATD, 2 SAD)  search_intra.c:133-158
 * Restricted to multi_ref_idx = 0 in the builder (MRL is off in every preset
 * used); angular_pred itself keeps the multi_ref_idx / isp arguments.
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"

#define REF_LEN 400   /* >= INTRA_REF_LENGTH (intra.h:46 = 358) */

unsigned ORC_FN(satd_nxn)(const orc_px *, const orc_px *, int);
unsigned ORC_FN(sad_nxn)(const orc_px *, const orc_px *, int);
unsigned ORC_FN(satd_any_size)(int, int, const orc_px *, int, const orc_px *, int);
unsigned ORC_FN(reg_sad)(const orc_px *, const orc_px *, int, int, unsigned, unsigned);

/*
 * Reference construction on a whole reconstructed plane (frame coordinates).
 * (x,y): block position in the plane; pic_w/pic_h: plane's visible size.
 * avail_top / avail_left: number of already-reconstructed samples available
 * above-right / left-below, i.e. the reference's px_available_top/left after
 * all its MIN()s (count_available_edge_cus*4, cu+pu size, picture size, WPP
 * clamp) -- the caller (who knows the coding order) supplies them.
 * Writes REF_LEN entries of top[] and left[]: index 0 = top-left corner.
 */
ORC_EXPORT void ORC_FN(intra_build_refs)(const orc_px *rec, int stride, int pic_w, int pic_h,
                                         int x, int y, int w, int h, int avail_top, int avail_left,
                                         orc_px *top, orc_px *left)
{
  const orc_px dc = (orc_px)(1 << (ORC_BIT_DEPTH - 1));
  /* left column */
  if (x > 0) {
    if (avail_left < 1) avail_left = 1;          /* the do/while loops copy at least one sample (intra.c:1262) */
    for (int i = 0; i < avail_left; ++i) left[1 + i] = rec[(y + i) * stride + x - 1];
    for (int i = avail_left; i < REF_LEN - 1; ++i) left[1 + i] = left[avail_left];
  } else {
    const orc_px v = y > 0 ? rec[(y - 1) * stride + x] : dc;       /* intra.c:881 */
    for (int i = 0; i < REF_LEN - 1; ++i) left[1 + i] = v;
  }
  /* corner */
  if (x > 0 && y > 0) left[0] = top[0] = rec[(y - 1) * stride + x - 1];
  else left[0] = top[0] = left[1];                                  /* intra.c:1003-1005 */
  /* top row */
  if (y > 0) {
    if (avail_top < 1) avail_top = 1;
    for (int i = 0; i < avail_top; ++i) top[1 + i] = rec[(y - 1) * stride + x + i];
    for (int i = avail_top; i < REF_LEN - 1; ++i) top[1 + i] = top[avail_top];
  } else {
    const orc_px v = x > 0 ? rec[y * stride + x - 1] : dc;          /* intra.c:1053 */
    for (int i = 0; i < REF_LEN - 1; ++i) top[1 + i] = v;
  }
}

/* intra.c:190-225; entries beyond 2N are copied through so that the arrays stay fully defined */
ORC_EXPORT void ORC_FN(intra_filter_refs)(const orc_px *top, const orc_px *left, int w, int h,
                                          orc_px *ftop, orc_px *fleft)
{
  const int rw = 2 * w + 1, rh = 2 * h + 1;
  memcpy(ftop, top, REF_LEN * sizeof(orc_px));
  memcpy(fleft, left, REF_LEN * sizeof(orc_px));
  fleft[0] = ftop[0] = (orc_px)((left[1] + 2 * left[0] + top[1] + 2) >> 2);
  for (int i = 1; i < rh - 1; ++i) fleft[i] = (orc_px)((left[i - 1] + 2 * left[i] + left[i + 1] + 2) >> 2);
  for (int i = 1; i < rw - 1; ++i) ftop[i] = (orc_px)((top[i - 1] + 2 * top[i] + top[i + 1] + 2) >> 2);
}

static const int16_t k_sample_disp[32] = {0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51, 57,
                                          64, 73, 86, 102, 128, 171, 256, 341, 512, 1024};
static const int16_t k_inv_disp[32] = {0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712, 630,
                                       565, 512, 468, 420, 364, 321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16};
static const int8_t k_pre_scale[32] = {8, 7, 6, 5, 5, 4, 4, 4, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 0, 0, 0,
                                       -1, -1, -2, -3};
static const int k_dist_thres[8] = {24, 24, 24, 14, 2, 0, 0, 0};
/* H.266 table 25 (fC): 4-tap cubic intra interpolation filter, 1/32 phases */
static const int8_t k_cubic[32][4] = {
  {0, 64, 0, 0},    {-1, 63, 2, 0},   {-2, 62, 4, 0},   {-2, 60, 7, -1},  {-2, 58, 10, -2}, {-3, 57, 12, -2},
  {-4, 56, 14, -2}, {-4, 55, 15, -2}, {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3},
  {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4}, {-4, 36, 36, -4}, {-4, 33, 39, -4},
  {-4, 30, 42, -4}, {-4, 29, 44, -5}, {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
  {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3}, {-2, 10, 58, -2}, {-1, 7, 60, -2},
  {0, 4, 62, -2},   {0, 2, 63, -1}};

/*
 * uvg_angular_pred_generic (intra-generic.c:55-295).  intra_mode may be a
 * wide-angle mode (-14..80).  width/height are the block's; dst is w*h.
 */
ORC_EXPORT void ORC_FN(angular_pred)(int width, int height, int intra_mode, int is_chroma,
                                     const orc_px *ref_above, const orc_px *ref_left, orc_px *dst,
                                     int multi_ref_idx, int isp_mode)
{
  const int lw = orc_log2i(width), lh = orc_log2i(height);
  const int vertical = intra_mode >= 34;
  const int mode_disp = vertical ? intra_mode - 50 : -(intra_mode - 18);
  const int amd = orc_iabs(mode_disp);
  const int sample_disp = (mode_disp < 0 ? -1 : 1) * k_sample_disp[amd];
  const int side_log2 = vertical ? lh : lw;
  int scale = side_log2 - k_pre_scale[amd];
  if (scale > 2) scale = 2;

  orc_px tmp_above[REF_LEN + 64], tmp_left[REF_LEN + 64], work_buf[32 * 32];
  memset(tmp_above, 0, sizeof tmp_above);
  memset(tmp_left, 0, sizeof tmp_left);
  orc_px *work = (width == height || vertical) ? dst : work_buf;

  const orc_px *ref_main, *ref_side;
  if (sample_disp < 0) {
    memcpy(tmp_above + height, ref_above, (size_t)(width + 2 + multi_ref_idx) * sizeof(orc_px));
    memcpy(tmp_left + width, ref_left, (size_t)(height + 2 + multi_ref_idx) * sizeof(orc_px));
    orc_px *m = vertical ? tmp_above + height : tmp_left + width;
    const orc_px *s = vertical ? tmp_left + width : tmp_above + height;
    const int size_side = vertical ? height : width;
    for (int i = -size_side; i <= -1; ++i) {
      int idx = (-i * k_inv_disp[amd] + 256) >> 9;
      if (idx > size_side) idx = size_side;
      m[i] = s[idx];
    }
    ref_main = m; ref_side = s;
  } else {
    ref_main = vertical ? ref_above : ref_left;
    ref_side = vertical ? ref_left : ref_above;
  }
  ref_main += multi_ref_idx;
  ref_side += multi_ref_idx;
  if (!vertical) { const int t = width; width = height; height = t; }

  if (sample_disp != 0) {
    int use_cubic = 1;
    const int dist = orc_iabs(intra_mode - 50) < orc_iabs(intra_mode - 18) ? orc_iabs(intra_mode - 50)
                                                                            : orc_iabs(intra_mode - 18);
    if (dist > k_dist_thres[(lw + lh) >> 1] && (orc_iabs(sample_disp) & 31) != 0) use_cubic = 0;
    if (multi_ref_idx || isp_mode) use_cubic = 1;

    /* PDPC applicability (intra-generic.c:237-246).  The reference tests
     * "pred_mode > 1 && pred_mode < 67" on an UNSIGNED copy of the mode, so
     * negative wide-angle modes skip the per-angle restrictions. */
    int pdpc = (width >= 4 && height >= 4) && multi_ref_idx == 0;
    if (intra_mode > 1 && intra_mode < 67) {
      if (mode_disp < 0 || multi_ref_idx) pdpc = 0;
      else if (mode_disp > 0) pdpc = pdpc && scale >= 0;
    }

    int delta_pos = sample_disp * (1 + multi_ref_idx);
    for (int y = 0; y < height; ++y, delta_pos += sample_disp) {
      const int di = delta_pos >> 5, df = delta_pos & 31;
      if ((orc_iabs(sample_disp) & 31) != 0) {
        if (!is_chroma) {
          int f[4];
          if (use_cubic) { for (int k = 0; k < 4; ++k) f[k] = k_cubic[df][k]; }
          else { f[0] = 16 - (df >> 1); f[1] = 32 - (df >> 1); f[2] = 16 + (df >> 1); f[3] = df >> 1; }
          for (int x = 0; x < width; ++x) {
            const orc_px *p = ref_main + di + x;
            work[y * width + x] = orc_clip_px((f[0] * p[0] + f[1] * p[1] + f[2] * p[2] + f[3] * p[3] + 32) >> 6);
          }
        } else {
          for (int x = 0; x < width; ++x) {
            const int r1 = ref_main[x + di + 1], r2 = ref_main[x + di + 2];
            work[y * width + x] = (orc_px)(r1 + ((df * (r2 - r1) + 16) >> 5));
          }
        }
      } else {
        for (int x = 0; x < width; ++x) work[y * width + x] = ref_main[x + di + 1];
      }
      if (pdpc) {
        int inv_sum = 256;
        const int lim = (3 << scale) < width ? (3 << scale) : width;
        for (int x = 0; x < lim; ++x) {
          inv_sum += k_inv_disp[amd];
          const int wl = 32 >> ((2 * x) >> scale);
          const int l = ref_side[y + (inv_sum >> 9) + 1];
          const int c = work[y * width + x];
          work[y * width + x] = (orc_px)(c + ((wl * (l - c) + 32) >> 6));
        }
      }
    }
  } else {
    const int pdpc = (width >= 4 && height >= 4) && multi_ref_idx == 0;   /* sample_disp == 0 >= 0 */
    const int sc = (lw + lh - 2) >> 2;
    const int tl = ref_main[0];
    for (int y = 0; y < height; ++y) {
      for (int x = 0; x < width; ++x) work[y * width + x] = ref_main[1 + x];
      if (pdpc) {
        const int l = ref_side[1 + y];
        const int lim = (3 << sc) < width ? (3 << sc) : width;
        for (int x = 0; x < lim; ++x) {
          const int wl = 32 >> ((2 * x) >> sc);
          work[y * width + x] = orc_clip_px(work[y * width + x] + ((wl * (l - tl) + 32) >> 6));
        }
      }
    }
  }

  if (!vertical) {   /* transpose back: (width,height) are the swapped dims here */
    if (width == height) {
      for (int y = 0; y < height - 1; ++y)
        for (int x = y + 1; x < width; ++x) {
          const orc_px t = work[y * height + x]; work[y * height + x] = work[x * width + y]; work[x * width + y] = t;
        }
    } else {
      for (int y = 0; y < width; ++y)
        for (int x = 0; x < height; ++x) dst[x + y * height] = work[y + x * width];
    }
  }
}

/* intra-generic.c:306-361 (int_fast16_t is 64-bit on x86-64: no overflow to model) */
ORC_EXPORT void ORC_FN(intra_pred_planar)(int width, int height, const orc_px *ref_top, const orc_px *ref_left,
                                          orc_px *dst)
{
  const int lw = orc_log2i(width), lh = orc_log2i(height);
  const int64_t offset = 1 << (lw + lh);
  const int shift = 1 + lw + lh;
  const int tr = ref_top[width + 1], bl = ref_left[height + 1];
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const int64_t hor = ((int64_t)ref_left[y + 1] << lw) + (int64_t)(x + 1) * (tr - ref_left[y + 1]);
      const int64_t ver = ((int64_t)ref_top[x + 1] << lh) + (int64_t)(y + 1) * (bl - ref_top[x + 1]);
      dst[y * width + x] = (orc_px)(((hor << lh) + (ver << lw) + offset) >> shift);
    }
}

/* intra.c:236-273 */
ORC_EXPORT void ORC_FN(intra_pred_dc)(int width, int height, const orc_px *ref_top, const orc_px *ref_left,
                                      orc_px *dst, int multi_ref_idx)
{
  int sum = 0;
  if (width >= height) for (int i = 0; i < width; ++i) sum += ref_top[i + 1 + multi_ref_idx];
  if (width <= height) for (int j = 0; j < height; ++j) sum += ref_left[j + 1 + multi_ref_idx];
  const int denom = width == height ? width << 1 : (width > height ? width : height);
  const int shift = orc_log2i(denom);
  const orc_px dc = (orc_px)((sum + (denom >> 1)) >> shift);
  for (int i = 0; i < width * height; ++i) dst[i] = dc;
}

/*
 * intra_pred_filtered_dc (intra-generic.c:371-402; HEVC-style DC with [1 2 1]/[1 3] boundary smoothing --
 * registered upstream but without a caller).  Square 2^log2_width block; the DC sum honours the
 * multi-reference-line offset, the boundary filter does not (as upstream).  dc is stored as a pixel.
 */
ORC_EXPORT void ORC_FN(intra_pred_filtered_dc)(int log2_width, const orc_px *ref_top, const orc_px *ref_left,
                                               orc_px *dst, int multi_ref_idx)
{
  const int n = 1 << log2_width;
  long total = 0;
  for (int i = 0; i < n; ++i) total += ref_top[i + 1 + multi_ref_idx] + ref_left[i + 1 + multi_ref_idx];
  const orc_px dc = (orc_px)((total + n) >> (log2_width + 1));
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      int v = dc;
      if (x == 0 && y == 0) v = (ref_left[1] + 2 * dc + ref_top[1] + 2) / 4;
      else if (y == 0) v = (ref_top[x + 1] + 3 * dc + 2) / 4;
      else if (x == 0) v = (ref_left[y + 1] + 3 * dc + 2) / 4;
      dst[y * n + x] = (orc_px)v;
    }
}

/* intra-generic.c:414-437 */
ORC_EXPORT void ORC_FN(pdpc_planar_dc)(int width, int height, const orc_px *ref_top, const orc_px *ref_left,
                                       orc_px *dst)
{
  const int scale = (orc_log2i(width) + orc_log2i(height) - 2) >> 2;
  for (int y = 0; y < height; ++y) {
    const int sy = (y << 1) >> scale, wt = 32 >> (sy > 31 ? 31 : sy);
    for (int x = 0; x < width; ++x) {
      const int sx = (x << 1) >> scale, wl = 32 >> (sx > 31 ? 31 : sx);
      const int c = dst[y * width + x];
      dst[y * width + x] = (orc_px)(c + ((wl * (ref_left[y + 1] - c) + wt * (ref_top[x + 1] - c) + 32) >> 6));
    }
  }
}

/* uvg_wide_angle_correction (intra.c:637-658), account_for_dc_planar = false */
ORC_EXPORT int ORC_FN(wide_angle_correction)(int mode, int log2_w, int log2_h)
{
  static const int shift_tab[6] = {0, 6, 10, 12, 14, 15};
  int m = mode;
  if (log2_w != log2_h && mode > 1 && mode <= 66) {
    const int d = orc_iabs(log2_w - log2_h);
    if (log2_w > log2_h && mode < 2 + shift_tab[d]) m += 65;
    else if (log2_h > log2_w && mode > 66 - shift_tab[d]) m -= 65;
  }
  return (int8_t)m;
}

/*
 * intra_predict_regular (intra.c:660-753) for a PU == CU block, no ISP, no MRL:
 * chooses filtered/unfiltered references, predicts, applies PDPC.  Returns 1
 * if the filtered references were used.
 */
ORC_EXPORT int ORC_FN(intra_predict)(int mode, int is_chroma, int width, int height,
                                     const orc_px *top, const orc_px *left,
                                     const orc_px *ftop, const orc_px *fleft, orc_px *dst)
{
  const int lw = orc_log2i(width), lh = orc_log2i(height);
  const int pred_mode = ORC_FN(wide_angle_correction)(mode, lw, lh);
  int filtered = 0;
  if (is_chroma || mode == 1 || (width == 4 && height == 4)) {
    filtered = 0;
  } else if (mode == 0) {
    filtered = width * height > 32;
  } else {
    const int d50 = orc_iabs(pred_mode - 50), d18 = orc_iabs(pred_mode - 18);
    const int dist = d50 < d18 ? d50 : d18;
    if (dist > k_dist_thres[(lw + lh) >> 1]) {
      const int md = pred_mode >= 34 ? pred_mode - 50 : 18 - pred_mode;
      /* int_fast8_t sample_disp in the reference (intra.c:711): the table value is
       * truncated to 8 bits before the "& 31" test */
      const int8_t sd = (int8_t)((md < 0 ? -1 : 1) * k_sample_disp[orc_iabs(md)]);
      if ((orc_iabs(sd) & 31) == 0) filtered = 1;
    }
  }
  const orc_px *t = filtered ? ftop : top, *l = filtered ? fleft : left;
  if (mode == 0) ORC_FN(intra_pred_planar)(width, height, t, l, dst);
  else if (mode == 1) ORC_FN(intra_pred_dc)(width, height, t, l, dst, 0);
  else ORC_FN(angular_pred)(width, height, pred_mode, is_chroma, t, l, dst, 0, 0);
  if ((mode == 0 || mode == 1) && width >= 4 && height >= 4) ORC_FN(pdpc_planar_dc)(width, height, t, l, dst);
  return filtered;
}

/*
 * Rough-search cost of every listed mode for one square luma block
 * (search_intra.c:133-158): min(SATD, 2*SAD) of the prediction against the
 * original, with the N x N strategy functions (4x4 SATD unshifted).
 * orig: contiguous n*n block.
 */
ORC_EXPORT void ORC_FN(intra_mode_costs)(const orc_px *rec, int stride, int pic_w, int pic_h, int x, int y, int n,
                                         int avail_top, int avail_left, const orc_px *orig,
                                         const int8_t *modes, int n_modes, uint32_t *costs, orc_px *preds_out)
{
  orc_px top[REF_LEN], left[REF_LEN], ftop[REF_LEN], fleft[REF_LEN], pred[32 * 32];
  ORC_FN(intra_build_refs)(rec, stride, pic_w, pic_h, x, y, n, n, avail_top, avail_left, top, left);
  ORC_FN(intra_filter_refs)(top, left, n, n, ftop, fleft);
  for (int m = 0; m < n_modes; ++m) {
    ORC_FN(intra_predict)(modes[m], 0, n, n, top, left, ftop, fleft, pred);
    const unsigned satd = ORC_FN(satd_nxn)(orig, pred, n);
    const unsigned sad = ORC_FN(sad_nxn)(pred, orig, n);
    costs[m] = satd < 2 * sad ? satd : 2 * sad;
    if (preds_out) memcpy(preds_out + (size_t)m * n * n, pred, (size_t)n * n * sizeof(orc_px));
  }
}

/*
 * Frame-level driver used by bench.py's cpu_baseline leg and by tests: rough
 * search costs of every n x n block of a plane (raster order), OpenMP over
 * blocks.  blks: n_blks x 4 int32 (x, y, avail_top, avail_left).
 */
ORC_EXPORT void ORC_FN(intra_search_frame)(const orc_px *rec, int rec_stride, const orc_px *orig, int orig_stride,
                                           int pic_w, int pic_h, int n, const int32_t *blks, int n_blks,
                                           const int8_t *modes, int n_modes, uint32_t *costs)
{
#pragma omp parallel for schedule(dynamic, 16)
  for (int b = 0; b < n_blks; ++b) {
    orc_px o[32 * 32];
    const int x = blks[4 * b], y = blks[4 * b + 1];
    for (int yy = 0; yy < n; ++yy) memcpy(o + yy * n, orig + (size_t)(y + yy) * orig_stride + x, (size_t)n * sizeof(orc_px));
    ORC_FN(intra_mode_costs)(rec, rec_stride, pic_w, pic_h, x, y, n, blks[4 * b + 2], blks[4 * b + 3], o, modes, n_modes,
                             costs + (size_t)b * n_modes, NULL);
  }
}

/* Frame-level driver: predict every n x n block with its own mode into a plane (bench cpu_baseline). */
ORC_EXPORT void ORC_FN(intra_pred_plane_frame)(const orc_px *rec, int rec_stride, int pic_w, int pic_h, int n,
                                               const int32_t *blks, int n_blks, const int8_t *block_modes,
                                               orc_px *pred, int pred_stride)
{
#pragma omp parallel for schedule(dynamic, 16)
  for (int b = 0; b < n_blks; ++b) {
    orc_px top[REF_LEN], left[REF_LEN], ftop[REF_LEN], fleft[REF_LEN], p[32 * 32];
    const int x = blks[4 * b], y = blks[4 * b + 1];
    ORC_FN(intra_build_refs)(rec, rec_stride, pic_w, pic_h, x, y, n, n, blks[4 * b + 2], blks[4 * b + 3], top, left);
    ORC_FN(intra_filter_refs)(top, left, n, n, ftop, fleft);
    ORC_FN(intra_predict)(block_modes[b], 0, n, n, top, left, ftop, fleft, p);
    for (int yy = 0; yy < n; ++yy) memcpy(pred + (size_t)(y + yy) * pred_stride + x, p + yy * n, (size_t)n * sizeof(orc_px));
  }
}
