/*
 * oracle/orc_sao_search.c -- restatement of the SAO decision of an all-intra picture as the encoder takes it CTU by CTU:
 *   uvg_sao_search_lcu                    sao.c:670-742   (merge candidates, luma + chroma search, merge decision)
 *   sao_search_luma / _chroma             sao.c:605-668   (the CTU's block of frame->rec and of the source, packed)
 *   sao_search_best_mode                  sao.c:490-603   (edge vs band vs nothing, merge costs)
 *   sao_search_edge_sao / _band_sao       sao.c:362-488
 *   calc_sao_band_offsets                 sao.c:208-262
 *   sao_mode_bits_none/merge/edge/band    sao.c:52-178    (CABAC_FBITS_UPDATE on state->search_cabac: update is 0 there, so the
 *                                                          estimates read the two SAO models as the coder left them before this CTU)
 *   encode_sao                            encoderstate.c:523-608  (the bins that adapt those two models, CTU to CTU; WPP: a row
 *                                                          starts from the models after the first CTU of the row above, :966-975)
 * and of its place in the per-CTU worker (encoderstate.c:841-853): the CTU is searched, then deblocked by
 * uvg_filter_deblock_lcu (its own edges only -- orcN_deblock_lcu), then uvg_sao_search_lcu reads its block.  After the
 * picture: uvg_sao_reconstruct of every CTU on the fully deblocked picture (encoder_sao_reconstruct, encoderstate.c:256-343).
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"
#include "orc_ctx_init.h"
#include <limits.h>
#include <math.h>

typedef struct orc_scu orc_scu;
void ORC_FN(deblock_lcu)(orc_px *y, int y_stride, orc_px *u, orc_px *v, int c_stride, int width, int height, const void *scu,
                         int scu_stride, int beta_offset_div2, int tc_offset_div2, int slice_is_b, int frame_qp,
                         const int8_t *chroma_qp_map, int x_px, int y_px);
void ORC_FN(calc_sao_edge_dir)(const orc_px *orig, const orc_px *rec, int eo_class, int bw, int bh, int cat_sum_cnt[2][5]);
int ORC_FN(sao_edge_ddistortion)(const orc_px *orig, const orc_px *rec, int bw, int bh, int eo_class, const int32_t offsets[5]);
int ORC_FN(sao_band_ddistortion)(const orc_px *orig, const orc_px *rec, int bw, int bh, int band_pos, const int32_t sao_bands[4]);
void ORC_FN(calc_sao_bands)(const orc_px *orig, const orc_px *rec, int bw, int bh, int sao_bands[2][32]);
void ORC_FN(sao_reconstruct_rect)(const orc_px *rec, orc_px *out, int stride, int pic_w, int pic_h, int fx, int fy, int w, int h,
                                  int type, int eo_class, const int32_t *band_position, const int32_t *offsets, int is_v);

enum { SAO_NONE = 0, SAO_BAND = 1, SAO_EDGE = 2,
       SAO_MAX = (1 << ((ORC_BIT_DEPTH < 10 ? ORC_BIT_DEPTH : 10) - 5)) - 1 };      /* SAO_ABS_OFFSET_MAX (sao.h): 7 at 8 bit, 31 at 10 */
typedef struct { int32_t type, eo_class, ddistortion, merge_left, merge_up, band_position[2], offsets[10]; } sao_info;   /* sao_info_t, 17 ints */
typedef struct { uint16_t s0[2], s1[2]; uint8_t rate[2]; } sao_models;      /* [0] sao_merge_flag_model, [1] sao_type_idx_model */

static double fbits(const sao_models *m, int c, int bin)
{
  const int st = (m->s0[c] + m->s1[c]) >> 8;
  const double p1 = (2 * st + 1) / 512.0;
  return (float)(floor(-log2(bin ? p1 : 1.0 - p1) * 32768.0 + 0.5) / 32768.0);      /* uvg_f_entropy_bits */
}
static void code_bin(sao_models *m, int c, int bin)                                 /* CTX_UPDATE, cabac.h:182-193 */
{
  const int r0 = m->rate[c] >> 4, r1 = m->rate[c] & 15;
  const unsigned mask0 = (~(~0u << 10)) << 5, mask1 = (~(~0u << 14)) << 1;
  m->s0[c] = (uint16_t)(m->s0[c] - ((m->s0[c] >> r0) & mask0));
  m->s1[c] = (uint16_t)(m->s1[c] - ((m->s1[c] >> r1) & mask1));
  if (bin) { m->s0[c] = (uint16_t)(m->s0[c] + ((0x7fffu >> r0) & mask0)); m->s1[c] = (uint16_t)(m->s1[c] + ((0x7fffu >> r1) & mask1)); }
}
static void models_init(sao_models *m, int qp, int slice)
{
  for (int i = 0; i < 2; ++i) {
    const int v = k_ctx_init_sao[slice][i];
    const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
    int s = ((slope * (qp - 16)) >> 1) + offset;
    s = s < 1 ? 1 : (s > 127 ? 127 : s);
    const int p1 = s << 8;
    m->s0[i] = (uint16_t)(p1 & ((~(~0u << 10)) << 5)); m->s1[i] = (uint16_t)(p1 & ((~(~0u << 14)) << 1));
    m->rate[i] = k_ctx_init_sao[3][i];
  }
}

static double bits_prefix(const sao_models *m, const sao_info *top, const sao_info *left, int type_bin)
{
  double b = 0.0;
  if (left) b += fbits(m, 0, 0);
  if (top) b += fbits(m, 0, 0);
  b += fbits(m, 1, type_bin);
  return b;
}
static double bits_none(const sao_models *m, const sao_info *top, const sao_info *left) { return bits_prefix(m, top, left, 0); }
static double bits_merge(const sao_models *m, int cand)
{
  double b = fbits(m, 0, cand == 1);
  if (cand == 1) return b;
  return b + fbits(m, 0, cand == 2);
}
static double bits_edge(const sao_models *m, const int *offsets, const sao_info *top, const sao_info *left, int bufs)
{
  double b = bits_prefix(m, top, left, 1);
  b += 1.0;
  for (int i = 0; i < bufs; ++i)
    for (int cat = 1; cat <= 4; ++cat) {
      const int a = abs(offsets[cat + 5 * i]);
      b += (a == 0 || a == SAO_MAX) ? a + 1 : a + 2;
    }
  return b + 2.0;
}
static double bits_band(const sao_models *m, const int *offsets, const sao_info *top, const sao_info *left, int bufs)
{
  double b = bits_prefix(m, top, left, 1);
  b += 1.0;
  for (int i = 0; i < bufs; ++i)
    for (int k = 0; k < 4; ++k) {
      const int a = abs(offsets[k + 1 + i * 5]);
      b += a == 0 ? a + 1 : (a == SAO_MAX ? a + 1 + 1 : a + 2 + 1);
    }
  return b + 5.0 * bufs;
}

static int band_offsets(int sao_bands[2][32], int offsets[4], int *band_position)        /* calc_sao_band_offsets */
{
  int dist[32], temp_offsets[32], best_dist, best_pos = 0;
  for (int band = 0; band < 32; ++band) {
    best_dist = INT_MAX;
    int offset = 0;
    if (sao_bands[1][band] != 0) {
      offset = (sao_bands[0][band] + (sao_bands[1][band] >> 1)) / sao_bands[1][band];
      offset = offset < -SAO_MAX ? -SAO_MAX : (offset > SAO_MAX ? SAO_MAX : offset);
    }
    dist[band] = offset == 0 ? 0 : INT_MAX;
    temp_offsets[band] = 0;
    while (offset != 0) {
      const int t = sao_bands[1][band] * offset * offset - 2 * offset * sao_bands[0][band];
      if (t < best_dist) { dist[band] = t; temp_offsets[band] = offset; }      /* sic: best_dist is never lowered (sao.c:236-240) */
      offset += offset > 0 ? -1 : 1;
    }
  }
  best_dist = INT_MAX;
  for (int band = 0; band < 28; ++band) {
    const int t = (int)((unsigned)dist[band] + (unsigned)dist[band + 1] + (unsigned)dist[band + 2] + (unsigned)dist[band + 3]);
    if (t < best_dist) { best_dist = t; best_pos = band; }
  }
  memcpy(offsets, &temp_offsets[best_pos], 4 * sizeof(int));
  *band_position = best_pos;
  return best_dist;
}

static void search_best(const sao_models *m, double lambda, int sao_type, const orc_px *const *data, const orc_px *const *rec, int bw, int bh,
                        int bufs, sao_info *out, const sao_info *top, const sao_info *left, int32_t merge_cost[3])
{
  sao_info edge, band;
  memset(&edge, 0, sizeof edge); memset(&band, 0, sizeof band);
  if (sao_type & 1) {
    /* sao_search_edge_sao */
    edge.type = SAO_EDGE; edge.ddistortion = INT_MAX;
    for (int cls = 0; cls < 4; ++cls) {
      int eo[10], sum = 0;
      memset(eo, 0, sizeof eo);
      for (int i = 0; i < bufs; ++i) {
        int sc[2][5];
        memset(sc, 0, sizeof sc);
        ORC_FN(calc_sao_edge_dir)(data[i], rec[i], cls, bw, bh, sc);
        for (int cat = 1; cat <= 4; ++cat) {
          const int cs = sc[0][cat], cc = sc[1][cat];
          int o = 0;
          if (cc != 0) { o = (cs + (cc >> 1)) / cc; o = o < -SAO_MAX ? -SAO_MAX : (o > SAO_MAX ? SAO_MAX : o); }
          if (cat <= 2 && o < 0) o = 0;
          if (cat >= 3 && o > 0) o = 0;
          eo[cat + 5 * i] = o;
          sum += cc * o * o - 2 * o * cs;
        }
      }
      sum += (int)(bits_edge(m, eo, top, left, bufs) * lambda + 0.5);
      eo[0] = 0; eo[5] = 0;
      if (sum < edge.ddistortion) { edge.eo_class = cls; edge.ddistortion = sum; memcpy(edge.offsets, eo, sizeof eo); }
    }
    int dd = (int)(bits_edge(m, edge.offsets, top, left, bufs) * lambda + 0.5);
    for (int i = 0; i < bufs; ++i) dd += ORC_FN(sao_edge_ddistortion)(data[i], rec[i], bw, bh, edge.eo_class, &edge.offsets[5 * i]);
    edge.ddistortion = dd;
  } else edge.ddistortion = INT_MAX;
  if (sao_type & 2) {
    /* sao_search_band_sao */
    band.type = SAO_BAND;
    int temp[10], dd = 0;
    memset(temp, 0, sizeof temp);
    for (int i = 0; i < bufs; ++i) {
      int sb[2][32];
      memset(sb, 0, sizeof sb);
      ORC_FN(calc_sao_bands)(data[i], rec[i], bw, bh, sb);
      dd += band_offsets(sb, &temp[1 + 5 * i], &band.band_position[i]);
    }
    dd += (int)(bits_band(m, temp, top, left, bufs) * lambda + 0.5);
    if (dd < INT_MAX) { band.ddistortion = dd; memcpy(band.offsets, temp, sizeof(int) * (size_t)bufs * 5); }
    dd = (int)(bits_band(m, band.offsets, top, left, bufs) * lambda + 0.5);
    for (int i = 0; i < bufs; ++i) dd += ORC_FN(sao_band_ddistortion)(data[i], rec[i], bw, bh, band.band_position[i], &band.offsets[1 + 5 * i]);
    band.ddistortion = dd;
  } else band.ddistortion = INT_MAX;
  if (edge.ddistortion <= band.ddistortion) { *out = edge; merge_cost[0] = edge.ddistortion; }
  else { *out = band; merge_cost[0] = band.ddistortion; }
  {
    const int nothing = (int)(bits_none(m, top, left) * lambda + 0.5);
    if (out->ddistortion >= nothing) { out->type = SAO_NONE; merge_cost[0] = nothing; }
  }
  const sao_info *cands[2] = {left, top};
  for (int i = 0; i < 2; ++i) {
    const sao_info *c = cands[i];
    if (!c) continue;
    int dd = (int)(bits_merge(m, i + 1) * lambda + 0.5);
    if (c->type == SAO_EDGE)
      for (int b = 0; b < bufs; ++b) dd += ORC_FN(sao_edge_ddistortion)(data[b], rec[b], bw, bh, c->eo_class, &c->offsets[5 * b]);
    else if (c->type == SAO_BAND)
      for (int b = 0; b < bufs; ++b) dd += ORC_FN(sao_band_ddistortion)(data[b], rec[b], bw, bh, c->band_position[b], &c->offsets[1 + 5 * b]);
    merge_cost[i + 1] = dd;
  }
}

static void pack_block(const orc_px *plane, int stride, int x, int y, int w, int h, orc_px *dst)
{
  for (int r = 0; r < h; ++r) memcpy(dst + (size_t)r * w, plane + (size_t)(y + r) * stride + x, sizeof(orc_px) * (size_t)w);
}

static void encode_sao(sao_models *m, int cx, int cy, const sao_info *luma, const sao_info *chroma)
{
  if (cx > 0) code_bin(m, 0, luma->merge_left);
  if (cy > 0 && !luma->merge_left) code_bin(m, 0, luma->merge_up);
  if (!luma->merge_left && !luma->merge_up) {
    code_bin(m, 1, luma->type != SAO_NONE);         /* Y */
    code_bin(m, 1, chroma->type != SAO_NONE);       /* U (V codes no type) */
  }
}

/*
 * One all-intra picture.  rec_y/u/v: the reconstruction before the in-loop filters (what the CTU search leaves), deblocked IN
 * PLACE; scu: the side information (orc_scu per 4x4).  Out: info[ctu][2][17] (luma, chroma sao_info_t), models[ctu][6] (the two
 * SAO models after the CTU's SAO syntax: s0, s1, rate each), snap_y/u/v (optional, picture-sized): every CTU's block as the
 * decision saw it, out_y/u/v: the picture after SAO.
 */
/* slice: 2 = I, 1 = P, 0 = B (the row of the context initialisation table; a B slice also switches the deblocking filter's
 * boundary-strength rule, filter.c:734-818) */
ORC_EXPORT void ORC_FN(sao_search_picture_slice)(const orc_px *src_y, const orc_px *src_u, const orc_px *src_v, orc_px *rec_y, orc_px *rec_u,
                                                 orc_px *rec_v, int width, int height, const void *scu, int scu_stride, int qp, double lambda,
                                                 int sao_type, int slice, int32_t *info_out, uint16_t *models_out, orc_px *snap_y, orc_px *snap_u,
                                                 orc_px *snap_v, orc_px *out_y, orc_px *out_u, orc_px *out_v);
ORC_EXPORT void ORC_FN(sao_search_picture)(const orc_px *src_y, const orc_px *src_u, const orc_px *src_v, orc_px *rec_y, orc_px *rec_u,
                                           orc_px *rec_v, int width, int height, const void *scu, int scu_stride, int qp, double lambda,
                                           int sao_type, int32_t *info_out, uint16_t *models_out, orc_px *snap_y, orc_px *snap_u,
                                           orc_px *snap_v, orc_px *out_y, orc_px *out_u, orc_px *out_v)
{
  ORC_FN(sao_search_picture_slice)(src_y, src_u, src_v, rec_y, rec_u, rec_v, width, height, scu, scu_stride, qp, lambda, sao_type, 2, info_out,
                                   models_out, snap_y, snap_u, snap_v, out_y, out_u, out_v);
}
ORC_EXPORT void ORC_FN(sao_search_picture_slice)(const orc_px *src_y, const orc_px *src_u, const orc_px *src_v, orc_px *rec_y, orc_px *rec_u,
                                                 orc_px *rec_v, int width, int height, const void *scu, int scu_stride, int qp, double lambda,
                                                 int sao_type, int slice, int32_t *info_out, uint16_t *models_out, orc_px *snap_y, orc_px *snap_u,
                                                 orc_px *snap_v, orc_px *out_y, orc_px *out_u, orc_px *out_v)
{
  const int wc = (width + 63) / 64, hc = (height + 63) / 64, cw = width / 2, ch = height / 2;
  sao_info *luma = calloc((size_t)wc * hc, sizeof(sao_info)), *chroma = calloc((size_t)wc * hc, sizeof(sao_info));
  sao_models *after = calloc((size_t)wc * hc, sizeof(sao_models));
  orc_px *po = malloc(sizeof(orc_px) * 64 * 64 * 3), *pr = malloc(sizeof(orc_px) * 64 * 64 * 3);
  for (int cy = 0; cy < hc; ++cy)
    for (int cx = 0; cx < wc; ++cx) {
      const int k = cy * wc + cx, x = cx * 64, y = cy * 64;
      ORC_FN(deblock_lcu)(rec_y, width, rec_u, rec_v, cw, width, height, scu, scu_stride, 0, 0, slice == 0, qp, NULL, x, y);
      sao_models m;
      if (cx > 0) m = after[k - 1];
      else if (cy > 0) m = after[(cy - 1) * wc];
      else models_init(&m, qp, slice);
      const sao_info *top_l = cy ? &luma[k - wc] : NULL, *left_l = cx ? &luma[k - 1] : NULL;
      const sao_info *top_c = cy ? &chroma[k - wc] : NULL, *left_c = cx ? &chroma[k - 1] : NULL;
      int32_t mc_l[3] = {INT_MAX, 0, 0}, mc_c[3] = {INT_MAX, 0, 0};
      {
        const int bw = x + 64 >= width ? width - x : 64, bh = y + 64 >= height ? height - y : 64;
        pack_block(src_y, width, x, y, bw, bh, po); pack_block(rec_y, width, x, y, bw, bh, pr);
        if (snap_y) for (int r = 0; r < bh; ++r) memcpy(snap_y + (size_t)(y + r) * width + x, pr + (size_t)r * bw, sizeof(orc_px) * (size_t)bw);
        const orc_px *d[1] = {po}, *r[1] = {pr};
        search_best(&m, lambda, sao_type, d, r, bw, bh, 1, &luma[k], top_l, left_l, mc_l);
      }
      {
        const int bw = x / 2 + 32 >= cw ? cw - x / 2 : 32, bh = y / 2 + 32 >= ch ? ch - y / 2 : 32;
        pack_block(src_u, cw, x / 2, y / 2, bw, bh, po); pack_block(src_v, cw, x / 2, y / 2, bw, bh, po + 1024);
        pack_block(rec_u, cw, x / 2, y / 2, bw, bh, pr); pack_block(rec_v, cw, x / 2, y / 2, bw, bh, pr + 1024);
        if (snap_u) for (int r = 0; r < bh; ++r) {
          memcpy(snap_u + (size_t)(y / 2 + r) * cw + x / 2, pr + (size_t)r * bw, sizeof(orc_px) * (size_t)bw);
          memcpy(snap_v + (size_t)(y / 2 + r) * cw + x / 2, pr + 1024 + (size_t)r * bw, sizeof(orc_px) * (size_t)bw);
        }
        const orc_px *d[2] = {po, po + 1024}, *r[2] = {pr, pr + 1024};
        search_best(&m, lambda, sao_type, d, r, bw, bh, 2, &chroma[k], top_c, left_c, mc_c);
      }
      luma[k].merge_up = luma[k].merge_left = 0;
      if (top_l && mc_l[2] + mc_c[2] <= mc_l[0] + mc_c[0]) {
        luma[k] = *top_l; chroma[k] = *top_c;
        luma[k].merge_up = 1; luma[k].merge_left = 0;
      }
      if (left_l && mc_l[1] + mc_c[1] <= mc_l[0] + mc_c[0]) {
        if (!luma[k].merge_up || mc_l[1] + mc_c[1] < mc_l[2] + mc_c[2]) {
          luma[k] = *left_l; chroma[k] = *left_c;
          luma[k].merge_left = 1; luma[k].merge_up = 0;
        }
      }
      encode_sao(&m, cx, cy, &luma[k], &chroma[k]);
      after[k] = m;
      memcpy(info_out + (size_t)k * 34, &luma[k], 17 * sizeof(int32_t));
      memcpy(info_out + (size_t)k * 34 + 17, &chroma[k], 17 * sizeof(int32_t));
      uint16_t *mo = models_out + (size_t)k * 6;
      mo[0] = m.s0[0]; mo[1] = m.s1[0]; mo[2] = m.rate[0]; mo[3] = m.s0[1]; mo[4] = m.s1[1]; mo[5] = m.rate[1];
    }
  /* the picture is fully deblocked now: SAO of every CTU on it */
  memcpy(out_y, rec_y, sizeof(orc_px) * (size_t)width * height);
  memcpy(out_u, rec_u, sizeof(orc_px) * (size_t)cw * ch); memcpy(out_v, rec_v, sizeof(orc_px) * (size_t)cw * ch);
  for (int cy = 0; cy < hc; ++cy)
    for (int cx = 0; cx < wc; ++cx) {
      const int k = cy * wc + cx, x = cx * 64, y = cy * 64;
      const int bw = x + 64 >= width ? width - x : 64, bh = y + 64 >= height ? height - y : 64;
      ORC_FN(sao_reconstruct_rect)(rec_y, out_y, width, width, height, x, y, bw, bh, luma[k].type, luma[k].eo_class, luma[k].band_position, luma[k].offsets, 0);
      ORC_FN(sao_reconstruct_rect)(rec_u, out_u, cw, cw, ch, x / 2, y / 2, bw / 2, bh / 2, chroma[k].type, chroma[k].eo_class, chroma[k].band_position, chroma[k].offsets, 0);
      ORC_FN(sao_reconstruct_rect)(rec_v, out_v, cw, cw, ch, x / 2, y / 2, bw / 2, bh / 2, chroma[k].type, chroma[k].eo_class, chroma[k].band_position, chroma[k].offsets, 1);
    }
  free(luma); free(chroma); free(after); free(po); free(pr);
}
