/*
 * oracle/orc_deblock.c -- restatement of uvg266's deblocking filter
 * (src/filter.c, plain C upstream, not behind the strategy API) as a
 * frame-level two-pass filter: every vertical edge of the picture, then every
 * horizontal edge.
 *
 * Why two whole-picture passes reproduce the reference: uvg_filter_deblock_lcu
 * (filter.c:1372-1380) filters, per CTU, the CTU's vertical edges, then the
 * horizontal edges of the previous CTU's rightmost 8 columns, then its own
 * horizontal edges minus the rightmost 8 columns.  VVC chooses the filter
 * lengths from the transform sizes on both sides (filter.c:587-644) so that
 * the samples one edge modifies are never read or written by another edge of
 * the same direction, and the 8-column delay guarantees that a horizontal
 * edge only ever sees samples whose vertical filtering is complete.  So the
 * per-CTU schedule and the two-pass schedule give the same picture;
 * tools/refcheck verifies this against uvg_filter_deblock_lcu run CTU by CTU.
 *
 * Reference followed:
 *   tc / beta tables                        filter.c:47-60   (H.266 table 43)
 *   strong / weak / chroma sample filters   filter.c:127-257
 *   large-block filter                      filter.c:406-524
 *   strong-filter decision                  filter.c:529-585
 *   max filter length                       filter.c:587-644
 *   luma edge (Bs, decisions)               filter.c:671-1009
 *   chroma edge                             filter.c:1036-1194
 *   which units are edges / skipped         filter.c:1207-1300
 * The per-4x4 side information the reference reads from cu_info_t is passed
 * as a table of orc_scu (the same 32-byte layout as uvghip_scu_t).
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"

typedef struct {
  uint8_t luma_edges, chroma_edges;   /* bit0: left edge (EDGE_VER), bit1: top edge (EDGE_HOR) */
  uint8_t type;                        /* 1 intra, 2 inter, 4 ibc */
  uint8_t cbf;                         /* bit0 Y, bit1 U, bit2 V */
  int8_t qp;
  uint8_t log2_width, log2_height, log2_chroma_width, log2_chroma_height;
  uint8_t isp_mode, mv_dir, pad;
  int16_t ref_id[2];
  int32_t mv[2][2];
} orc_scu;

typedef struct {
  int beta_offset_div2, tc_offset_div2;
  int slice_is_b;
  int frame_qp;                 /* >= 0: constant QP (max_qp_delta_depth < 0), -1: per-CU average */
  const int8_t *chroma_qp_map;  /* qp_map[0] or NULL */
} orc_dbk_cfg;

static const uint16_t k_tc[66] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 4, 4, 4, 4, 5, 5, 5, 5, 7, 7, 8, 9, 10, 10,
                                  11, 13, 14, 15, 17, 19, 21, 24, 25, 29, 33, 36, 41, 45, 51, 57, 64, 71, 80, 89, 100, 112,
                                  125, 141, 157, 177, 198, 222, 250, 280, 314, 352, 395};
static const uint8_t k_beta[64] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20,
                                   22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64, 66,
                                   68, 70, 72, 74, 76, 78, 80, 82, 84, 86, 88};

static int tc_from_index(int idx)
{
  return ORC_BIT_DEPTH < 10 ? (k_tc[idx] + (1 << (9 - ORC_BIT_DEPTH))) >> (10 - ORC_BIT_DEPTH)
                            : k_tc[idx] << (ORC_BIT_DEPTH - 10);
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* luma TU size seen from an edge of direction dir (filter.c:846-882) */
static int tu_size_luma(const orc_scu *c, int dir_hor)
{
  const int cw = 1 << c->log2_width, ch = 1 << c->log2_height;
  if (c->type == 1 && c->isp_mode) {
    if (c->isp_mode == 2 && !dir_hor) return imax(4, cw >> 2);
    if (c->isp_mode == 1 && dir_hor) return imax(4, ch >> 2);
  }
  return imin(dir_hor ? ch : cw, 32);
}

/* boundary strength of a luma edge segment (filter.c:734-818) */
static int luma_strength(const orc_scu *p, const orc_scu *q, int tu_boundary, const orc_dbk_cfg *cfg)
{
  if (q->type == 1 || p->type == 1) return 2;
  if (tu_boundary && ((q->cbf & 1) || (p->cbf & 1))) return 1;
  const int thr = 1 << 3;   /* half a sample in 1/16 units */
  if (p->mv_dir == 3 || q->mv_dir == 3 || cfg->slice_is_b) {
    int32_t mq[2][2], mp[2][2];
    for (int l = 0; l < 2; ++l)
      for (int k = 0; k < 2; ++k) {
        mq[l][k] = (q->mv_dir & (1 << l)) ? q->mv[l][k] : 0;
        mp[l][k] = (p->mv_dir & (1 << l)) ? p->mv[l][k] : 0;
      }
    const int rp0 = p->type == 4 ? -2 : (p->mv_dir & 1) ? p->ref_id[0] : -1;
    const int rp1 = p->type == 4 ? -2 : (p->mv_dir & 2) ? p->ref_id[1] : -1;
    const int rq0 = q->type == 4 ? -2 : (q->mv_dir & 1) ? q->ref_id[0] : -1;
    const int rq1 = q->type == 4 ? -2 : (q->mv_dir & 2) ? q->ref_id[1] : -1;
#define FAR(a, b) (orc_iabs((a)[0] - (b)[0]) >= thr || orc_iabs((a)[1] - (b)[1]) >= thr)
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0)) {
      if (rp0 != rp1) {
        if (rp0 == rq0) return (FAR(mq[0], mp[0]) || FAR(mq[1], mp[1])) ? 1 : 0;
        return (FAR(mq[1], mp[0]) || FAR(mq[0], mp[1])) ? 1 : 0;
      }
      return ((FAR(mq[0], mp[0]) || FAR(mq[1], mp[1])) && (FAR(mq[1], mp[0]) || FAR(mq[0], mp[1]))) ? 1 : 0;
    }
    return 1;
  }
  {
    const int rp = p->type == 4 ? -2 : p->ref_id[0], rq = q->type == 4 ? -2 : q->ref_id[0];
    if (rp != rq) return 1;
    return FAR(q->mv[0], p->mv[0]) ? 1 : 0;
#undef FAR
  }
}

static int qp_pred(const orc_scu *p, const orc_scu *q, const orc_dbk_cfg *cfg)
{
  if (cfg->frame_qp >= 0) return cfg->frame_qp;       /* filter.c:341-343 */
  return (p->qp + q->qp + 1) >> 1;
}

/* filter.c:529-585, normal (non-large) decision; ctb = chroma CTB-boundary variant */
static int strong_normal(const int *p0l, const int *q0l, const int *p3l, const int *q3l, int dp0, int dq0, int dp3, int dq3,
                         int tc, int beta, int ctb)
{
  const int sp0 = ctb ? orc_iabs(p0l[1] - p0l[0]) : orc_iabs(p0l[3] - p0l[0]);
  const int sp3 = ctb ? orc_iabs(p3l[1] - p3l[0]) : orc_iabs(p3l[3] - p3l[0]);
  return 2 * (dp0 + dq0) < (beta >> 2) && 2 * (dp3 + dq3) < (beta >> 2) &&
         orc_iabs(p0l[0] - q0l[0]) < ((5 * tc + 1) >> 1) && orc_iabs(p3l[0] - q3l[0]) < ((5 * tc + 1) >> 1) &&
         sp0 + orc_iabs(q0l[0] - q0l[3]) < (beta >> 3) && sp3 + orc_iabs(q3l[0] - q3l[3]) < (beta >> 3);
}

/* one line of the large-block filter (filter.c:406-524). P[k] = p_k, Q[k] = q_k, k = 0..7 */
static void large_block_line(int *P, int *Q, int tc, int lenP, int lenQ)
{
  static const int c7[7] = {59, 50, 41, 32, 23, 14, 5}, c5[5] = {58, 45, 32, 19, 6}, c3[3] = {53, 32, 11};
  static const int t7[7] = {6, 5, 4, 3, 2, 1, 1}, t3[3] = {6, 4, 2};
  const int *cP = lenP == 7 ? c7 : lenP == 5 ? c5 : c3, *cQ = lenQ == 7 ? c7 : lenQ == 5 ? c5 : c3;
  const int refP = (P[lenP - 1] + P[lenP] + 1) >> 1, refQ = (Q[lenQ - 1] + Q[lenQ] + 1) >> 1;
  int mid;
  if (lenP == lenQ) {
    if (lenP == 7) mid = (P[6] + P[5] + P[4] + P[3] + P[2] + P[1] + 2 * (P[0] + Q[0]) + Q[1] + Q[2] + Q[3] + Q[4] + Q[5] + Q[6] + 8) >> 4;
    else mid = (P[4] + P[3] + 2 * (P[2] + P[1] + P[0] + Q[0] + Q[1] + Q[2]) + Q[3] + Q[4] + 8) >> 4;
  } else {
    const int lenS = imin(lenP, lenQ), lenL = imax(lenP, lenQ);
    const int *S = lenP < lenQ ? P : Q, *L = lenP < lenQ ? Q : P;
    if (lenL == 7 && lenS == 5) mid = (P[5] + P[4] + P[3] + P[2] + 2 * (P[1] + P[0] + Q[0] + Q[1]) + Q[2] + Q[3] + Q[4] + Q[5] + 8) >> 4;
    else if (lenL == 7 && lenS == 3) mid = (3 * S[0] + 2 * L[0] + 3 * S[1] + L[1] + 2 * S[2] + L[2] + L[3] + L[4] + L[5] + L[6] + 8) >> 4;
    else mid = (P[3] + P[2] + P[1] + P[0] + Q[0] + Q[1] + Q[2] + Q[3] + 4) >> 3;
  }
  int nP[7], nQ[7];
  for (int i = 0; i < lenP; ++i) {
    const int r = (tc * (lenP == 3 ? t3[i] : t7[i])) >> 1;
    nP[i] = orc_clip3(P[i] - r, P[i] + r, (mid * cP[i] + refP * (64 - cP[i]) + 32) >> 6);
  }
  for (int i = 0; i < lenQ; ++i) {
    const int r = (tc * (lenQ == 3 ? t3[i] : t7[i])) >> 1;
    nQ[i] = orc_clip3(Q[i] - r, Q[i] + r, (mid * cQ[i] + refQ * (64 - cQ[i]) + 32) >> 6);
  }
  for (int i = 0; i < lenP; ++i) P[i] = nP[i];
  for (int i = 0; i < lenQ; ++i) Q[i] = nQ[i];
}

/* One 4-sample luma edge segment whose first Q sample is (x,y) (filter.c:671-1009). */
static void luma_segment(orc_px *plane, int stride, const orc_scu *scu, int scu_stride, int x, int y, int dir_hor,
                         const orc_dbk_cfg *cfg)
{
  const orc_scu *q = scu + (y >> 2) * scu_stride + (x >> 2);
  const orc_scu *p = dir_hor ? q - scu_stride : q - 1;
  const int tu_boundary = (q->luma_edges & (dir_hor ? 2 : 1)) != 0;
  const int qp = qp_pred(p, q, cfg);
  const int beta = k_beta[orc_clip3(0, 63, qp + (cfg->beta_offset_div2 << 1))] * (1 << (ORC_BIT_DEPTH - 8));
  const int side_thr = (beta + (beta >> 1)) >> 3;
  const int bs = luma_strength(p, q, tu_boundary, cfg);
  if (!bs) return;
  const int tc = tc_from_index(orc_clip3(0, 65, qp + 2 * (bs - 1) + (cfg->tc_offset_div2 << 1)));

  const int tq = tu_size_luma(q, dir_hor), tp = tu_size_luma(p, dir_hor);
  int lenP, lenQ;
  if (tp <= 4 || tq <= 4) lenP = lenQ = 1;
  else { lenP = tp >= 32 ? 7 : 3; lenQ = tq >= 32 ? 7 : 3; }
  const int largeP = lenP > 3 && !(dir_hor && (y % 64) == 0);   /* CTU-boundary line-buffer restriction (filter.c:890) */
  const int largeQ = lenQ > 3;

  const int xs = dir_hor ? stride : 1, ys = dir_hor ? 1 : stride;   /* across / along the edge */
  orc_px *e = plane + (size_t)y * stride + x;
  int P[4][8], Q[4][8];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 8; ++k) {
      /* samples further than 4 from the edge are only read on a large side, where they exist */
      P[i][k] = (k < 4 || largeP) ? e[i * ys - (k + 1) * xs] : 0;
      Q[i][k] = (k < 4 || largeQ) ? e[i * ys + k * xs] : 0;
    }
  const int dp0 = orc_iabs(P[0][2] - 2 * P[0][1] + P[0][0]), dq0 = orc_iabs(Q[0][0] - 2 * Q[0][1] + Q[0][2]);
  const int dp3 = orc_iabs(P[3][2] - 2 * P[3][1] + P[3][0]), dq3 = orc_iabs(Q[3][0] - 2 * Q[3][1] + Q[3][2]);
  const int dp = dp0 + dp3, dq = dq0 + dq3;
  int sw = 0, done = 0;

  if (largeP || largeQ) {
    int dp0L = dp0, dq0L = dq0, dp3L = dp3, dq3L = dq3;
    if (largeP) {
      dp0L = (dp0L + orc_iabs(P[0][5] - 2 * P[0][4] + P[0][3]) + 1) >> 1;
      dp3L = (dp3L + orc_iabs(P[3][5] - 2 * P[3][4] + P[3][3]) + 1) >> 1;
    }
    if (largeQ) {
      dq0L = (dq0L + orc_iabs(Q[0][3] - 2 * Q[0][4] + Q[0][5]) + 1) >> 1;
      dq3L = (dq3L + orc_iabs(Q[3][3] - 2 * Q[3][4] + Q[3][5]) + 1) >> 1;
    }
    if (dp0L + dp3L + dq0L + dq3L < beta) {
      int sp[2], sq[2];
      for (int j = 0; j < 2; ++j) {
        const int *Pl = P[j * 3], *Ql = Q[j * 3];
        sp[j] = orc_iabs(Pl[3] - Pl[0]);
        sq[j] = orc_iabs(Ql[0] - Ql[3]);
        if (largeP) {
          int t;
          if (lenP == 7) { t = Pl[7]; sp[j] += orc_iabs(Pl[4] - Pl[5] - Pl[6] + t); } else t = Pl[5];
          sp[j] = (sp[j] + orc_iabs(Pl[3] - t) + 1) >> 1;
        }
        if (largeQ) {
          int t;
          if (lenQ == 7) { t = Ql[7]; sq[j] += orc_iabs(Ql[4] - Ql[5] - Ql[6] + t); } else t = Ql[5];
          sq[j] = (sq[j] + orc_iabs(t - Ql[3]) + 1) >> 1;
        }
      }
      sw = 2 * (dp0L + dq0L) < (beta >> 4) && 2 * (dp3L + dq3L) < (beta >> 4) &&
           orc_iabs(P[0][0] - Q[0][0]) < ((5 * tc + 1) >> 1) && orc_iabs(P[3][0] - Q[3][0]) < ((5 * tc + 1) >> 1) &&
           sp[0] + sq[0] < ((beta * 3) >> 5) && sp[1] + sq[1] < ((beta * 3) >> 5);
      if (sw) {
        for (int i = 0; i < 4; ++i) large_block_line(P[i], Q[i], tc, largeP ? lenP : 3, largeQ ? lenQ : 3);
        done = 1;
      }
    }
  }
  if (!sw && dp + dq < beta) {
    if (lenP > 2 && lenQ > 2) sw = strong_normal(P[0], Q[0], P[3], Q[3], dp0, dq0, dp3, dq3, tc, beta, 0);
    for (int i = 0; i < 4; ++i) {
      int *Pl = P[i], *Ql = Q[i];
      if (sw) {   /* filter.c:127-149 */
        const int m0 = Pl[3], m1 = Pl[2], m2 = Pl[1], m3 = Pl[0], m4 = Ql[0], m5 = Ql[1], m6 = Ql[2], m7 = Ql[3];
        Pl[2] = orc_clip3(m1 - tc, m1 + tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
        Pl[1] = orc_clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
        Pl[0] = orc_clip3(m3 - 3 * tc, m3 + 3 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
        Ql[0] = orc_clip3(m4 - 3 * tc, m4 + 3 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
        Ql[1] = orc_clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
        Ql[2] = orc_clip3(m6 - tc, m6 + tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
      } else {    /* filter.c:159-198 */
        const int p2nd = (lenP > 1 && lenQ > 1) ? dp < side_thr : 0, q2nd = (lenP > 1 && lenQ > 1) ? dq < side_thr : 0;
        const int m1 = Pl[2], m2 = Pl[1], m3 = Pl[0], m4 = Ql[0], m5 = Ql[1], m6 = Ql[2];
        int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
        if (orc_iabs(delta) < tc * 10) {
          const int tc2 = tc >> 1;
          delta = orc_clip3(-tc, tc, delta);
          Pl[0] = orc_clip3(0, ORC_PX_MAX, m3 + delta);
          Ql[0] = orc_clip3(0, ORC_PX_MAX, m4 - delta);
          if (p2nd) Pl[1] = orc_clip3(0, ORC_PX_MAX, m2 + orc_clip3(-tc2, tc2, (((m1 + m3 + 1) >> 1) - m2 + delta) >> 1));
          if (q2nd) Ql[1] = orc_clip3(0, ORC_PX_MAX, m5 + orc_clip3(-tc2, tc2, (((m6 + m4 + 1) >> 1) - m5 - delta) >> 1));
        }
      }
    }
    done = 1;
  }
  if (!done) return;
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 8; ++k) {
      if (k < 4 || largeP) e[i * ys - (k + 1) * xs] = (orc_px)P[i][k];
      if (k < 4 || largeQ) e[i * ys + k * xs] = (orc_px)Q[i][k];
    }
}

/* One 2-sample chroma edge segment, both chroma planes; (xc,yc) = first Q sample in chroma
 * coordinates (filter.c:1036-1194 with length 2, 4:2:0). */
static void chroma_segment(orc_px *pu, orc_px *pv, int stride, const orc_scu *scu, int scu_stride, int xc, int yc,
                           int dir_hor, const orc_dbk_cfg *cfg)
{
  const int x = xc << 1, y = yc << 1;
  const orc_scu *q = scu + (y >> 2) * scu_stride + (x >> 2);
  const orc_scu *p = dir_hor ? scu + ((y - 1) >> 2) * scu_stride + (x >> 2) : scu + (y >> 2) * scu_stride + ((x - 1) >> 2);
  const int tu_boundary = (q->luma_edges & (dir_hor ? 2 : 1)) != 0;     /* the caller's luma tu_boundary (filter.c:1247-1256) */
  const int luma_qp = qp_pred(p, q, cfg);
  int QP;
  if (cfg->chroma_qp_map) QP = cfg->chroma_qp_map[luma_qp]; else QP = orc_clip3(0, 57, luma_qp);   /* transform.c:150-165, offset 0 */
  const int tp = imin(1 << (dir_hor ? p->log2_chroma_height : p->log2_chroma_width), 32);
  const int tq = imin(1 << (dir_hor ? q->log2_chroma_height : q->log2_chroma_width), 32);
  const int large = tp >= 8 && tq >= 8;
  const int ctb = dir_hor && (y % 64) == 0;
  int bs[2] = {0, 0};
  if (q->type == 1 || p->type == 1) bs[0] = bs[1] = 2;
  else if (tu_boundary) { bs[0] = ((q->cbf | p->cbf) & 2) ? 1 : 0; bs[1] = ((q->cbf | p->cbf) & 4) ? 1 : 0; }
  const int xs = dir_hor ? stride : 1, ys = dir_hor ? 1 : stride;
  for (int comp = 0; comp < 2; ++comp) {
    if (!(bs[comp] == 2 || (large && bs[comp] == 1))) continue;
    const int tc = tc_from_index(orc_clip3(0, 65, QP + 2 * (bs[comp] - 1) + (cfg->tc_offset_div2 << 1)));
    orc_px *e = (comp ? pv : pu) + (size_t)yc * stride + xc;
    int P[2][4], Q[2][4];
    for (int i = 0; i < 2; ++i)
      for (int k = 0; k < 4; ++k) {
        P[i][k] = (k < 2 || large) ? e[i * ys - (k + 1) * xs] : 0;   /* p2,p3 only exist/are read on 8+ sample sides */
        Q[i][k] = (k < 2 || large) ? e[i * ys + k * xs] : 0;
      }
    int sw = 0, long_filter = 0;
    if (large) {
      const int beta = k_beta[orc_clip3(0, 63, QP + (cfg->beta_offset_div2 << 1))] * (1 << (ORC_BIT_DEPTH - 8));
      const int pi = ctb ? 1 : 2;    /* b[p_ind]: p1 at the CTB boundary, else p2 (filter.c:1153) */
      const int dp0 = orc_iabs(P[0][pi] - 2 * P[0][1] + P[0][0]), dq0 = orc_iabs(Q[0][0] - 2 * Q[0][1] + Q[0][2]);
      const int dp3 = orc_iabs(P[1][pi] - 2 * P[1][1] + P[1][0]), dq3 = orc_iabs(Q[1][0] - 2 * Q[1][1] + Q[1][2]);
      if (dp0 + dp3 + dq0 + dq3 < beta) {
        long_filter = 1;
        sw = strong_normal(P[0], Q[0], P[1], Q[1], dp0, dq0, dp3, dq3, tc, beta, ctb);
      }
    }
    (void)long_filter;
    for (int i = 0; i < 2; ++i) {
      const int m0 = P[i][3], m1 = P[i][2], m2 = P[i][1], m3 = P[i][0], m4 = Q[i][0], m5 = Q[i][1], m6 = Q[i][2], m7 = Q[i][3];
      if (sw) {   /* filter.c:220-236 */
        if (ctb) {
          e[i * ys - xs] = (orc_px)orc_clip3(m3 - tc, m3 + tc, (3 * m2 + 2 * m3 + m4 + m5 + m6 + 4) >> 3);
          e[i * ys] = (orc_px)orc_clip3(m4 - tc, m4 + tc, (2 * m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4) >> 3);
        } else {
          e[i * ys - 3 * xs] = (orc_px)orc_clip3(m1 - tc, m1 + tc, (3 * m0 + 2 * m1 + m2 + m3 + m4 + 4) >> 3);
          e[i * ys - 2 * xs] = (orc_px)orc_clip3(m2 - tc, m2 + tc, (2 * m0 + m1 + 2 * m2 + m3 + m4 + m5 + 4) >> 3);
          e[i * ys - xs] = (orc_px)orc_clip3(m3 - tc, m3 + tc, (m0 + m1 + m2 + 2 * m3 + m4 + m5 + m6 + 4) >> 3);
          e[i * ys] = (orc_px)orc_clip3(m4 - tc, m4 + tc, (m1 + m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4) >> 3);
        }
        e[i * ys + xs] = (orc_px)orc_clip3(m5 - tc, m5 + tc, (m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4) >> 3);
        e[i * ys + 2 * xs] = (orc_px)orc_clip3(m6 - tc, m6 + tc, (m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4) >> 3);
      } else {    /* filter.c:237-241 */
        const int delta = orc_clip3(-tc, tc, (((m4 - m3) * 4) + m2 - m5 + 4) >> 3);
        e[i * ys - xs] = (orc_px)orc_clip3(0, ORC_PX_MAX, m3 + delta);
        e[i * ys] = (orc_px)orc_clip3(0, ORC_PX_MAX, m4 - delta);
      }
    }
  }
}

/*
 * Deblock a whole picture in place.  y/u/v: planes (4:2:0), scu: one orc_scu per 4x4 luma
 * block, row-major with scu_stride entries per row.  u/v may be NULL (luma only).
 */
ORC_EXPORT void ORC_FN(deblock_frame)(orc_px *y, int y_stride, orc_px *u, orc_px *v, int c_stride, int width, int height,
                                      const orc_scu *scu, int scu_stride, int beta_offset_div2, int tc_offset_div2,
                                      int slice_is_b, int frame_qp, const int8_t *chroma_qp_map)
{
  const orc_dbk_cfg cfg = {beta_offset_div2, tc_offset_div2, slice_is_b, frame_qp, chroma_qp_map};
  for (int dir_hor = 0; dir_hor < 2; ++dir_hor) {
    const int bit = dir_hor ? 2 : 1;
    for (int by = 0; by < height; by += 4)
      for (int bx = 0; bx < width; bx += 4) {
        if ((!dir_hor && bx == 0) || (dir_hor && by == 0)) continue;          /* filter.c:1219-1220 */
        const orc_scu *c = scu + (by >> 2) * scu_stride + (bx >> 2);
        if (c->luma_edges & bit) luma_segment(y, y_stride, scu, scu_stride, bx, by, dir_hor, &cfg);
        /* chroma: needs a luma-flagged unit (filter_deblock_lcu_inside only visits those), the chroma flag,
         * and the 8x8 chroma grid (filter.c:1247-1256) */
        if (u && (c->luma_edges & bit) && (c->chroma_edges & bit)) {
          const int xc = bx >> 1, yc = by >> 1;
          if (dir_hor ? (yc & 7) == 0 : (xc & 7) == 0) chroma_segment(u, v, c_stride, scu, scu_stride, xc, yc, dir_hor, &cfg);
        }
      }
  }
}

/*
 * uvg_filter_deblock_lcu (filter.c:1372-1380) for the CTU at (x_px, y_px), in place, in the reference's own order:
 * the CTU's vertical edges (left boundary included), the horizontal edges of the last 8 luma columns of the CTU to the
 * left (filter_deblock_lcu_rightmost, :1303-1340), then the CTU's horizontal edges minus its own last 8 columns unless it
 * is the picture's last CTU of the row (filter_deblock_unit, :1224-1238).  Calling it for every CTU in raster order gives the
 * picture of deblock_frame; what matters here is the state in between: uvg_sao_search_lcu (encoderstate.c:849) reads the CTU's
 * block right after this call.
 */
static void deblock_unit(orc_px *y, int y_stride, orc_px *u, orc_px *v, int c_stride, const orc_scu *scu, int scu_stride,
                         int bx, int by, int dir_hor, const orc_dbk_cfg *cfg)
{
  const int bit = dir_hor ? 2 : 1;
  if ((!dir_hor && bx == 0) || (dir_hor && by == 0)) return;
  const orc_scu *c = scu + (by >> 2) * scu_stride + (bx >> 2);
  if (c->luma_edges & bit) luma_segment(y, y_stride, scu, scu_stride, bx, by, dir_hor, cfg);
  if (u && (c->luma_edges & bit) && (c->chroma_edges & bit)) {
    const int xc = bx >> 1, yc = by >> 1;
    if (dir_hor ? (yc & 7) == 0 : (xc & 7) == 0) chroma_segment(u, v, c_stride, scu, scu_stride, xc, yc, dir_hor, cfg);
  }
}
ORC_EXPORT void ORC_FN(deblock_lcu)(orc_px *y, int y_stride, orc_px *u, orc_px *v, int c_stride, int width, int height,
                                    const orc_scu *scu, int scu_stride, int beta_offset_div2, int tc_offset_div2,
                                    int slice_is_b, int frame_qp, const int8_t *chroma_qp_map, int x_px, int y_px)
{
  const orc_dbk_cfg cfg = {beta_offset_div2, tc_offset_div2, slice_is_b, frame_qp, chroma_qp_map};
  const int end_x = imin(x_px + 64, width), end_y = imin(y_px + 64, height);
  for (int by = y_px; by < end_y; by += 4)
    for (int bx = x_px; bx < end_x; bx += 4) deblock_unit(y, y_stride, u, v, c_stride, scu, scu_stride, bx, by, 0, &cfg);
  if (x_px > 0)
    for (int bx = x_px - 8; bx < x_px; bx += 4)
      for (int by = y_px; by < end_y; by += 4) deblock_unit(y, y_stride, u, v, c_stride, scu, scu_stride, bx, by, 1, &cfg);
  for (int by = y_px; by < end_y; by += 4)
    for (int bx = x_px; bx < end_x; bx += 4) {
      if ((bx & 63) >= 56 && bx < width - 8) continue;        /* "the last 8 pixels will be deblocked when processing the next LCU" */
      deblock_unit(y, y_stride, u, v, c_stride, scu, scu_stride, bx, by, 1, &cfg);
    }
}
