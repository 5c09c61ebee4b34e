/*
 * oracle/orc_sao.c -- restatement of the "sao" strategy group and the plain-C
 * helpers around it:
 *   calc_sao_edge_dir        strategies/generic/sao-generic.c:51-81
 *   sao_edge_ddistortion     strategies/generic/sao_shared_generics.h:53-88
 *   sao_band_ddistortion     strategies/generic/sao_shared_generics.h:90-127
 *   sao_reconstruct_color    strategies/generic/sao-generic.c:84-124
 *   uvg_calc_sao_offset_array / calc_sao_bands / uvg_sao_reconstruct   sao.c:180-201,268-285,302-361
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"

static const int k_eo_ofs[4][2][2] = {   /* sao.h:71-76: {a,b} x {dx,dy} */
  {{-1, 0}, {1, 0}}, {{0, -1}, {0, 1}}, {{-1, -1}, {1, 1}}, {{1, -1}, {-1, 1}}};

static int sgn3(int v) { return (v > 0) - (v < 0); }
static int eo_cat(int a, int b, int c)
{
  static const int map[5] = {1, 2, 0, 3, 4};   /* sao_shared_generics.h:45 */
  return map[2 + sgn3(c - a) + sgn3(c - b)];
}

/* orig/rec: packed block_width x block_height; accumulates into cat_sum_cnt[2][5] */
ORC_EXPORT void ORC_FN(calc_sao_edge_dir)(const orc_px *orig, const orc_px *rec, int eo_class, int bw, int bh,
                                          int cat_sum_cnt[2][5])
{
  const int ax = k_eo_ofs[eo_class][0][0], ay = k_eo_ofs[eo_class][0][1];
  const int bx = k_eo_ofs[eo_class][1][0], by = k_eo_ofs[eo_class][1][1];
  for (int y = 1; y < bh - 1; ++y)
    for (int x = 1; x < bw - 1; ++x) {
      const int c = rec[y * bw + x];
      const int cat = eo_cat(rec[(y + ay) * bw + x + ax], rec[(y + by) * bw + x + bx], c);
      cat_sum_cnt[0][cat] += orig[y * bw + x] - c;
      cat_sum_cnt[1][cat] += 1;
    }
}

ORC_EXPORT int ORC_FN(sao_edge_ddistortion)(const orc_px *orig, const orc_px *rec, int bw, int bh, int eo_class,
                                            const int32_t offsets[5])
{
  const int ax = k_eo_ofs[eo_class][0][0], ay = k_eo_ofs[eo_class][0][1];
  const int bx = k_eo_ofs[eo_class][1][0], by = k_eo_ofs[eo_class][1][1];
  int32_t sum = 0;
  for (int y = 1; y < bh - 1; ++y)
    for (int x = 1; x < bw - 1; ++x) {
      const int c = rec[y * bw + x];
      const int off = offsets[eo_cat(rec[(y + ay) * bw + x + ax], rec[(y + by) * bw + x + bx], c)];
      if (off) {
        const int diff = orig[y * bw + x] - c, delta = diff - off;
        sum += delta * delta - diff * diff;
      }
    }
  return sum;
}

ORC_EXPORT int ORC_FN(sao_band_ddistortion)(const orc_px *orig, const orc_px *rec, int bw, int bh, int band_pos,
                                            const int32_t sao_bands[4])
{
  const int shift = ORC_BIT_DEPTH - 5;
  int sum = 0;
  for (int i = 0; i < bw * bh; ++i) {
    const int band = (rec[i] >> shift) - band_pos;
    const int off = (band >= 0 && band <= 3) ? sao_bands[band] : 0;
    if (off) {
      const int diff = orig[i] - rec[i], delta = diff - off;
      sum += delta * delta - diff * diff;
    }
  }
  return sum;
}

/* sao.c:268-285 */
ORC_EXPORT void ORC_FN(calc_sao_bands)(const orc_px *orig, const orc_px *rec, int bw, int bh, int sao_bands[2][32])
{
  const int shift = ORC_BIT_DEPTH - 5;
  for (int i = 0; i < bw * bh; ++i) {
    const int b = rec[i] >> shift;
    sao_bands[0][b] += orig[i] - rec[i];
    sao_bands[1][b] += 1;
  }
}

/*
 * sao_reconstruct_color (sao-generic.c:84-124).  type: 1 band, 2 edge.
 * offsets: the 10-entry sao_info_t.offsets array; band_position[2]; is_v selects the V halves.
 * rec may be read up to one sample outside the w x h area for edge classes.
 */
ORC_EXPORT void ORC_FN(sao_reconstruct_color)(const orc_px *rec, orc_px *out, int type, int eo_class,
                                              const int32_t *band_position, const int32_t *offsets,
                                              int stride, int out_stride, int w, int h, int is_v)
{
  if (type == 1) {
    const int shift = ORC_BIT_DEPTH - 5, bp = band_position[is_v ? 1 : 0];
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int v = rec[y * stride + x], band = (v >> shift) - bp;
        out[y * out_stride + x] = (band >= 0 && band <= 3) ? orc_clip_px(v + offsets[band + 1 + (is_v ? 5 : 0)]) : (orc_px)v;
      }
  } else if (type == 2) {
    const int ax = k_eo_ofs[eo_class][0][0], ay = k_eo_ofs[eo_class][0][1];
    const int bx = k_eo_ofs[eo_class][1][0], by = k_eo_ofs[eo_class][1][1];
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int c = rec[y * stride + x];
        const int cat = eo_cat(rec[(y + ay) * stride + x + ax], rec[(y + by) * stride + x + bx], c);
        out[y * out_stride + x] = orc_clip_px(c + offsets[cat + (is_v ? 5 : 0)]);
      }
  }
}

/*
 * uvg_sao_reconstruct (sao.c:302-361) on whole planes: applies one CTU's parameters to the
 * rectangle (fx,fy,w,h) of `rec` (deblocked input plane), writing `out` (same geometry),
 * skipping picture-border rows/columns for edge classes exactly as :321-348.
 */
ORC_EXPORT void ORC_FN(sao_reconstruct_rect)(const orc_px *rec, orc_px *out, int stride, int pic_w, int pic_h,
                                             int fx, int fy, int w, int h, int type, int eo_class,
                                             const int32_t *band_position, const int32_t *offsets, int is_v)
{
  if (type == 2) {
    const int ax = k_eo_ofs[eo_class][0][0], ay = k_eo_ofs[eo_class][0][1];
    const int bx = k_eo_ofs[eo_class][1][0], by = k_eo_ofs[eo_class][1][1];
    if (fx + w + ax > pic_w || fx + w + bx > pic_w) w -= 1;
    if (fx + ax < 0 || fx + bx < 0) { fx += 1; w -= 1; }
    if (fy + h + ay > pic_h || fy + h + by > pic_h) h -= 1;
    if (fy + ay < 0 || fy + by < 0) { fy += 1; h -= 1; }
  }
  if (type != 0)
    ORC_FN(sao_reconstruct_color)(rec + (size_t)fy * stride + fx, out + (size_t)fy * stride + fx, type, eo_class,
                                  band_position, offsets, stride, stride, w, h, is_v);
}

/*
 * Frame-level statistics as the batched kernel produces them: for each rectangle r (x,y,w,h)
 * edge[r][class][2][5] (interior samples only, neighbours taken inside the rectangle = the
 * reference's packed copies) and bands[r][2][32].
 */
ORC_EXPORT void ORC_FN(sao_stats_rects)(const orc_px *orig, const orc_px *rec, int stride, const int32_t *rects, int n,
                                        int32_t *edge, int32_t *bands)
{
#pragma omp parallel for schedule(dynamic, 4)
  for (int r = 0; r < n; ++r) {
    const int x0 = rects[4 * r], y0 = rects[4 * r + 1], w = rects[4 * r + 2], h = rects[4 * r + 3];
    orc_px *po = malloc(sizeof(orc_px) * (size_t)w * h), *pr = malloc(sizeof(orc_px) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
      memcpy(po + y * w, orig + (size_t)(y0 + y) * stride + x0, sizeof(orc_px) * (size_t)w);
      memcpy(pr + y * w, rec + (size_t)(y0 + y) * stride + x0, sizeof(orc_px) * (size_t)w);
    }
    int32_t *e = edge + (size_t)r * 40, *b = bands + (size_t)r * 64;
    memset(e, 0, 40 * sizeof(int32_t)); memset(b, 0, 64 * sizeof(int32_t));
    for (int c = 0; c < 4; ++c) ORC_FN(calc_sao_edge_dir)(po, pr, c, w, h, (int(*)[5])(e + c * 10));
    ORC_FN(calc_sao_bands)(po, pr, w, h, (int(*)[32])b);
    free(po); free(pr);
  }
}

/* sao_search_edge_sao's arithmetic (sao.c:380-439) on the statistics; rate term supplied or zero.
 * Depth-independent, exported once (8-bit build). */
#if ORC_BIT_DEPTH == 8
ORC_EXPORT void orc_sao_edge_offsets(const int32_t *edge_stats, const int32_t *rate_cost, int n,
                                     int32_t *params /* [n][8] */, int32_t *ddist)
{
  for (int i = 0; i < n; ++i) {
    const int32_t *st = edge_stats + (size_t)i * 40;
    int best_dd = INT32_MAX, best_class = 0, best_off[5] = {0, 0, 0, 0, 0};
    for (int c = 0; c < 4; ++c) {
      int off[5] = {0, 0, 0, 0, 0}, dd = 0;
      for (int cat = 1; cat <= 4; ++cat) {
        const int sum = st[c * 10 + cat], cnt = st[c * 10 + 5 + cat];
        int o = 0;
        if (cnt != 0) {
          o = (sum + (cnt >> 1)) / cnt;                    /* sao.c:401 */
          o = o < -7 ? -7 : (o > 7 ? 7 : o);
        }
        if (cat <= 2 && o < 0) o = 0;                      /* sao.c:406-411 */
        if (cat >= 3 && o > 0) o = 0;
        off[cat] = o;
        dd += cnt * o * o - 2 * o * sum;                   /* sao.c:421 */
      }
      if (rate_cost) dd += rate_cost[(size_t)i * 4 + c];
      if (dd < best_dd) { best_dd = dd; best_class = c; for (int k = 0; k < 5; ++k) best_off[k] = off[k]; }
    }
    int32_t *P = params + (size_t)i * 8;
    P[0] = 2; P[1] = best_class; P[2] = 0;
    for (int k = 0; k < 5; ++k) P[3 + k] = best_off[k];
    if (ddist) ddist[i] = best_dd;
  }
}
#endif
