/*
 * oracle/orc_alf_picture.c -- the reconstruction half of the per-picture ALF process, given the encoder's decisions:
 *   alf_reconstruct_coeff / alf_reconstruct_coeff_aps   src/alf.c:2925-2986, 4332-4368   (an APS's coded filters -> a table per class)
 *   the fixed filter sets and the clipping values        src/alf.c:5244-5279               (uvg_alf_enc_process's arr_vars)
 *   alf_reconstruct                                      src/alf.c:5032-5137               (per CTU: luma 7x7 by filter set, chroma 5x5 by alternative)
 *   apply_cc_alf_filter / filter_blk_cc_alf              src/alf.c:1726-1775, 1626-1725    (cross-component filter of both chroma planes)
 * on whole planes, with the block functions of orc_alf.c.  Pinned by records around uvg_alf_enc_process in runs of the real encoder
 * (tools/refcheck/ctu_dump.c "alf"; tests/golden/ref_alf_*.npz; tests/test_oracle_alf_picture.py): the picture the process got -> the
 * picture it left.  The DERIVATION of the decisions (alf_encoder, alf_encoder_ctb, derive_cc_alf_filter) is not restated.
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"
#include <string.h>
#include <stdlib.h>

void ORC_FN(alf_classify_frame)(const orc_px *rec, int stride, int w, int h, int shift, int vbh, int vb_pos, uint8_t *cls, int cls_stride);
void ORC_FN(alf_filter_rect)(const orc_px *src, orc_px *dst, int stride, int pic_w, int pic_h, int x0, int y0, int w, int h, int chroma,
                             const int16_t *coef_set, const int16_t *clip_set, const uint8_t *cls, int cls_stride, int vbh, int vb_pos);

enum { CLASSES = 25, LC = 13, CC = 7, LN = CLASSES * LC, APS_WORDS = 2 * LN + CLASSES + 2, N_APS = 8, N_FIXED = 16, N_ALT = 8, N_CCF = 4, CCW = 8 };

static inline int at(const orc_px *p, int stride, int w, int h, int x, int y)
{
  return p[(size_t)orc_clip3(0, h - 1, y) * stride + orc_clip3(0, w - 1, x)];
}

/* alf_clipping_values (alf.c:5248-5260): [0] = 1 << depth, [i] = 1 << (7 - 2 i + depth - 8) */
static void clip_values(int16_t v[4])
{
  v[0] = (int16_t)(1 << ORC_BIT_DEPTH);
  for (int i = 1; i < 4; ++i) v[i] = (int16_t)(1 << (7 - 2 * i + ORC_BIT_DEPTH - 8));
}

/* one coded luma APS (coefficients and clip indices of its filters, the class -> filter map) as the 25 x 13 tables the block filter
 * takes (alf_reconstruct_coeff with is_rdo = 0, alf.c:2964-2984) */
static void expand_luma_aps(const int16_t *aps, int16_t *coef, int16_t *clip)
{
  int16_t cv[4];
  clip_values(cv);
  const int16_t *c = aps, *k = aps + LN, *map = aps + 2 * LN;
  const int non_linear = aps[2 * LN + CLASSES + 1];
  for (int cl = 0; cl < CLASSES; ++cl) {
    const int f = map[cl];
    for (int i = 0; i < LC - 1; ++i) {
      coef[cl * LC + i] = c[f * LC + i];
      clip[cl * LC + i] = cv[non_linear ? k[f * LC + i] : 0];
    }
    coef[cl * LC + LC - 1] = (int16_t)(1 << (ORC_BIT_DEPTH - 1));
    clip[cl * LC + LC - 1] = cv[0];
  }
}

/* filter_blk_cc_alf (alf.c:1626-1725), 4:2:0: the chroma rectangle (x0, y0, w, h) (chroma samples) of `dst` gets a 7-tap filter of
 * the luma plane added; vb: the luma virtual boundary row 60 of every 64 */
static void cc_alf_rect(orc_px *dst, int cstride, const orc_px *luma, int lstride, int W, int H, const int16_t *f, int x0, int y0, int w, int h)
{
  const int vb_pos = 60, vbh = 64;
  for (int y = y0; y < y0 + h; ++y) {
    const int ly = y << 1, pos = ly & (vbh - 1);
    int o1 = 1, o2 = -1, o3 = 2;
    if (pos == vb_pos - 2 || pos == vb_pos + 1) o3 = o1;
    else if (pos == vb_pos - 1 || pos == vb_pos) { o1 = 0; o2 = 0; o3 = 0; }
    for (int x = x0; x < x0 + w; ++x) {
      const int lx = x << 1;
#define L(dx, dy) at(luma, lstride, W, H, lx + (dx), ly + (dy))
      const int cur = L(0, 0);
      int sum = 0;
      sum += f[0] * (L(0, o2) - cur);
      sum += f[1] * (L(-1, 0) - cur);
      sum += f[2] * (L(1, 0) - cur);
      sum += f[3] * (L(-1, o1) - cur);
      sum += f[4] * (L(0, o1) - cur);
      sum += f[5] * (L(1, o1) - cur);
      sum += f[6] * (L(0, o3) - cur);
#undef L
      sum = (sum + 64) >> 7;
      const int offset = 1 << ORC_BIT_DEPTH >> 1;
      sum = (int)orc_clip_px(sum + offset) - offset;
      dst[(size_t)y * cstride + x] = orc_clip_px(sum + dst[(size_t)y * cstride + x]);
    }
  }
}

/*
 * get_blk_stats_cc_alf / calc_covariance_cc_alf (alf.c:2583-2779), 4:2:0: the CC-ALF covariance of the chroma rectangle (x0, y0, w, h)
 * (chroma samples; a CTU: y0 a multiple of 32) -- per sample e[k] = the seven luma tap differences around (2x, 2y) of the picture BEFORE
 * ALF, d = org - rec of the chroma plane AFTER its ALF; ee[k][l] = sum e[k] e[l] (full symmetric 7x7), y[k] = sum e[k] d, pix = sum d d.
 * The rows bend at the luma virtual boundary like the filter's -- except in the picture's last CTU row, where the reference moves the
 * boundary out of reach (vb_pos = frame_height, :2652-2655).  Pinned function by function: tools/refcheck/rc_alfstatic.c reaches the
 * reference's static functions by compiling its alf.c into the dev tool.
 */
ORC_EXPORT void ORC_FN(cc_alf_stats_rect)(const orc_px *org, int ostride, const orc_px *rec_c, int cstride, const orc_px *luma, int lstride, int W, int H,
                                          int x0, int y0, int w, int h, int64_t *ee, int32_t *yv, int64_t *pix)
{
  int64_t e2[7][7] = {{0}}, ys[7] = {0}, pa = 0;
  const int last_row = (y0 << 1) + 64 >= H;
  for (int i = 0; i < h; ++i) {
    const int vbd = last_row ? -(1 << 20) : (((i << 1) % 64) - 60);
    int o_m1 = -1, o_p1 = 1, o_p2 = 2;
    if (vbd == -2 || vbd == 1) o_p2 = o_p1;
    else if (vbd == -1 || vbd == 0) { o_m1 = 0; o_p1 = 0; o_p2 = 0; }
    for (int j = 0; j < w; ++j) {
      const int lx = (x0 + j) << 1, ly = (y0 + i) << 1;
#define L(dx, dy) at(luma, lstride, W, H, lx + (dx), ly + (dy))
      const int c = L(0, 0);
      const int e[7] = {L(0, o_m1) - c, L(-1, 0) - c, L(1, 0) - c, L(-1, o_p1) - c, L(0, o_p1) - c, L(1, o_p1) - c, L(0, o_p2) - c};
#undef L
      const int d = (int)org[(size_t)(y0 + i) * ostride + x0 + j] - (int)rec_c[(size_t)(y0 + i) * cstride + x0 + j];
      for (int k = 0; k < 7; ++k) {
        for (int l = k; l < 7; ++l) e2[k][l] += (int64_t)e[k] * e[l];
        ys[k] += (int64_t)e[k] * d;
      }
      pa += (int64_t)d * d;
    }
  }
  for (int k = 0; k < 7; ++k) {
    for (int l = 0; l < 7; ++l) ee[k * 7 + l] = k <= l ? e2[k][l] : e2[l][k];
    yv[k] = (int32_t)ys[k];
  }
  *pix = pa;
}

/*
 * meta: the record's header (ctu_dump.c): [3] alf_type, [4..6] slice enable Y / Cb / Cr, [7] number of luma APSs, [17..18] CC-ALF on for Cb / Cr,
 * [28] cfg.input_bitdepth (the classification's activity shift is that + 4 -- the depth of the INPUT, which the runs behind the goldens leave at 8).
 * flags[7][n]: CTU enable Y / Cb / Cr, chroma alternative Cb / Cr, CC-ALF control Cb / Cr.  set_idx[n]: < 16 a fixed set, else luma APS set_idx - 16.
 * luma_aps[8][APS_WORDS], chroma_aps[2 * 8 * 7 + 2], cc_coeff[2][4][8] as recorded; fixed: 64 x 13 coefficients then 16 x 25 class -> filter.
 * Planes are tight (stride = width).  -> 0, or -1 for a combination the reference itself leaves undefined (CC-ALF without luma ALF: its luma
 * source alf_tmp_y is only filled by alf_reconstruct, alf.c:5066).
 */
ORC_EXPORT int ORC_FN(alf_reconstruct_picture)(const orc_px *pre_y, const orc_px *pre_u, const orc_px *pre_v, int W, int H, orc_px *out_y, orc_px *out_u,
                                               orc_px *out_v, const int32_t *meta, const uint8_t *flags, const int16_t *set_idx, const int16_t *luma_aps,
                                               const int16_t *chroma_aps, const int16_t *cc_coeff, const int16_t *fixed)
{
  const int wc = (W + 63) / 64, hc = (H + 63) / 64, n = wc * hc, CW = W / 2, CH = H / 2;
  memcpy(out_y, pre_y, sizeof(orc_px) * (size_t)W * H);
  memcpy(out_u, pre_u, sizeof(orc_px) * (size_t)CW * CH);
  memcpy(out_v, pre_v, sizeof(orc_px) * (size_t)CW * CH);
  const int luma_on = meta[4];
  if (luma_on) {            /* alf_reconstruct returns at once otherwise -- chroma included (alf.c:5035-5038) */
    int16_t cv[4];
    clip_values(cv);
    int16_t (*coef)[LN] = malloc(sizeof(int16_t) * LN * (N_FIXED + N_APS)), (*clip)[LN] = malloc(sizeof(int16_t) * LN * (N_FIXED + N_APS));
    for (int s = 0; s < N_FIXED; ++s)
      for (int cl = 0; cl < CLASSES; ++cl) {
        const int fi = fixed[64 * LC + s * CLASSES + cl];
        for (int i = 0; i < LC - 1; ++i) coef[s][cl * LC + i] = fixed[fi * LC + i];
        coef[s][cl * LC + LC - 1] = (int16_t)(1 << (ORC_BIT_DEPTH - 1));
        for (int i = 0; i < LC; ++i) clip[s][cl * LC + i] = cv[0];
      }
    for (int i = 0; i < meta[7] && i < N_APS; ++i) expand_luma_aps(luma_aps + (size_t)i * APS_WORDS, coef[N_FIXED + i], clip[N_FIXED + i]);
    int16_t ccoef[N_ALT][CC], cclip[N_ALT][CC];
    {
      const int non_linear = chroma_aps[2 * N_ALT * CC + 1];
      for (int t = 0; t < N_ALT; ++t) {
        for (int i = 0; i < CC - 1; ++i) { ccoef[t][i] = chroma_aps[t * CC + i]; cclip[t][i] = cv[non_linear ? chroma_aps[(N_ALT + t) * CC + i] : 0]; }
        ccoef[t][CC - 1] = (int16_t)(1 << (ORC_BIT_DEPTH - 1)); cclip[t][CC - 1] = cv[0];
      }
    }
    const int cls_stride = (W + 3) / 4;
    uint8_t *cls = malloc((size_t)cls_stride * ((H + 3) / 4));
    ORC_FN(alf_classify_frame)(pre_y, W, W, H, meta[28] + 4, 64, 60, cls, cls_stride);      /* cfg.input_bitdepth + 4 (alf.c:5185) */
    for (int k = 0; k < n; ++k) {
      const int x = (k % wc) * 64, y = (k / wc) * 64, w = x + 64 > W ? W - x : 64, h = y + 64 > H ? H - y : 64;
      if (flags[k]) {
        const int s = set_idx[k];
        ORC_FN(alf_filter_rect)(pre_y, out_y, W, W, H, x, y, w, h, 0, coef[s], clip[s], cls, cls_stride, 64, 60);
      }
      for (int c = 1; c < 3; ++c)
        if (flags[c * n + k]) {
          const int alt = flags[(2 + c) * n + k];
          ORC_FN(alf_filter_rect)(c == 1 ? pre_u : pre_v, c == 1 ? out_u : out_v, CW, CW, CH, x / 2, y / 2, w / 2, h / 2, 1, ccoef[alt], cclip[alt], NULL, 0, 32, 30);
        }
    }
    free(cls); free(coef); free(clip);
  }
  if (meta[3] != 2) return 0;            /* UVG_ALF_FULL only (alf.c:5363-5366) */
  for (int c = 0; c < 2; ++c) {
    if (!meta[17 + c]) continue;
    if (!luma_on) return -1;
    for (int k = 0; k < n; ++k) {
      const int ctl = flags[(5 + c) * n + k];
      if (!ctl) continue;
      const int x = (k % wc) * 64, y = (k / wc) * 64, w = x + 64 > W ? W - x : 64, h = y + 64 > H ? H - y : 64;
      cc_alf_rect(c ? out_v : out_u, CW, pre_y, W, W, H, cc_coeff + ((size_t)c * N_CCF + (ctl - 1)) * CCW, x / 2, y / 2, w / 2, h / 2);
    }
  }
  return 0;
}
