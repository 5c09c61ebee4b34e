// Deblocking filter device functions (src/filter.c): boundary strength, luma / chroma edge segments.  Shared by the whole-picture
// kernel (deblock.hip) and the per-CTU filter job of the P / B picture pipeline (ctu_filter.h).  See deblock.hip for the citations.
#pragma once
#include "uvghip_common.h"

__device__ static const uint16_t kTc[66] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 4, 4, 4, 4, 5, 5, 5, 5, 7, 7, 8, 9,
                                            10, 10, 11, 13, 14, 15, 17, 19, 21, 24, 25, 29, 33, 36, 41, 45, 51, 57, 64, 71, 80,
                                            89, 100, 112, 125, 141, 157, 177, 198, 222, 250, 280, 314, 352, 395};
__device__ static const uint8_t kBeta[64] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17,
                                             18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58,
                                             60, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80, 82, 84, 86, 88};

struct dbk_cfg {
  int beta_offset_div2, tc_offset_div2, slice_is_b, frame_qp;
  int has_qp_map;
  int snapshot;        // the picture as uvg_sao_search_lcu sees each CTU: an edge on a CTU boundary leaves the CTU before it alone,
                       // and horizontal edges skip a CTU's last 8 luma columns unless they are the picture's (filter.c:1224-1238, 1341-1380)
  int8_t qp_map[64];
};

template <int DEPTH> __device__ __forceinline__ int tc_from_index(int idx)
{
  return DEPTH < 10 ? (kTc[idx] + (1 << (9 - DEPTH))) >> (10 - DEPTH) : kTc[idx] << (DEPTH - 10);
}

__device__ __forceinline__ int tu_size_luma(const uvghip_scu_t &c, bool dir_hor)
{
  const int cw = 1 << c.log2_width, ch = 1 << c.log2_height;
  if (c.type == 1 && c.isp_mode) {
    if (c.isp_mode == 2 && !dir_hor) return max(4, cw >> 2);
    if (c.isp_mode == 1 && dir_hor) return max(4, ch >> 2);
  }
  return min(dir_hor ? ch : cw, 32);
}

__device__ __forceinline__ bool mv_far(const int32_t *a, const int32_t *b)
{
  return abs(a[0] - b[0]) >= 8 || abs(a[1] - b[1]) >= 8;   // half a sample at 1/16 precision
}

// filter.c:734-818
__device__ inline int luma_strength(const uvghip_scu_t &p, const uvghip_scu_t &q, bool tu_boundary, const dbk_cfg &cfg)
{
  if (q.type == 1 || p.type == 1) return 2;
  if (tu_boundary && ((q.cbf | p.cbf) & 1)) return 1;
  if (p.mv_dir == 3 || q.mv_dir == 3 || cfg.slice_is_b) {
    int32_t mq[2][2], mp[2][2];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        mq[l][k] = (q.mv_dir & (1 << l)) ? q.mv[l][k] : 0;
        mp[l][k] = (p.mv_dir & (1 << l)) ? p.mv[l][k] : 0;
      }
    const int rp0 = p.type == 4 ? -2 : (p.mv_dir & 1) ? p.ref_id[0] : -1;
    const int rp1 = p.type == 4 ? -2 : (p.mv_dir & 2) ? p.ref_id[1] : -1;
    const int rq0 = q.type == 4 ? -2 : (q.mv_dir & 1) ? q.ref_id[0] : -1;
    const int rq1 = q.type == 4 ? -2 : (q.mv_dir & 2) ? q.ref_id[1] : -1;
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0)) {
      if (rp0 != rp1) {
        if (rp0 == rq0) return (mv_far(mq[0], mp[0]) || mv_far(mq[1], mp[1])) ? 1 : 0;
        return (mv_far(mq[1], mp[0]) || mv_far(mq[0], mp[1])) ? 1 : 0;
      }
      return ((mv_far(mq[0], mp[0]) || mv_far(mq[1], mp[1])) && (mv_far(mq[1], mp[0]) || mv_far(mq[0], mp[1]))) ? 1 : 0;
    }
    return 1;
  }
  const int rp = p.type == 4 ? -2 : p.ref_id[0], rq = q.type == 4 ? -2 : q.ref_id[0];
  if (rp != rq) return 1;
  return mv_far(q.mv[0], p.mv[0]) ? 1 : 0;
}

__device__ __forceinline__ int qp_pred(const uvghip_scu_t &p, const uvghip_scu_t &q, const dbk_cfg &cfg)
{
  return cfg.frame_qp >= 0 ? cfg.frame_qp : (p.qp + q.qp + 1) >> 1;
}

// filter.c:529-585, normal decision.  P/Q: samples of one line, index = distance from the edge.
__device__ __forceinline__ bool strong_normal(const int *P0, const int *Q0, const int *P3, const int *Q3, int dp0, int dq0,
                                              int dp3, int dq3, int tc, int beta, bool ctb)
{
  const int sp0 = ctb ? abs(P0[1] - P0[0]) : abs(P0[3] - P0[0]);
  const int sp3 = ctb ? abs(P3[1] - P3[0]) : abs(P3[3] - P3[0]);
  return 2 * (dp0 + dq0) < (beta >> 2) && 2 * (dp3 + dq3) < (beta >> 2) &&
         abs(P0[0] - Q0[0]) < ((5 * tc + 1) >> 1) && abs(P3[0] - Q3[0]) < ((5 * tc + 1) >> 1) &&
         sp0 + abs(Q0[0] - Q0[3]) < (beta >> 3) && sp3 + abs(Q3[0] - Q3[3]) < (beta >> 3);
}

// filter.c:406-524 for the length pairs that occur (7/7, 7/3, 3/7; 5 is kept for completeness)
__device__ inline void large_block_line(int *P, int *Q, int tc, int lenP, int lenQ)
{
  const int c7[7] = {59, 50, 41, 32, 23, 14, 5}, c5[5] = {58, 45, 32, 19, 6}, c3[3] = {53, 32, 11};
  const int t7[7] = {6, 5, 4, 3, 2, 1, 1}, t3[3] = {6, 4, 2};
  const int refP = (P[lenP - 1] + P[lenP] + 1) >> 1, refQ = (Q[lenQ - 1] + Q[lenQ] + 1) >> 1;
  int mid;
  if (lenP == lenQ) {
    if (lenP == 7) mid = (P[6] + P[5] + P[4] + P[3] + P[2] + P[1] + 2 * (P[0] + Q[0]) + Q[1] + Q[2] + Q[3] + Q[4] + Q[5] + Q[6] + 8) >> 4;
    else mid = (P[4] + P[3] + 2 * (P[2] + P[1] + P[0] + Q[0] + Q[1] + Q[2]) + Q[3] + Q[4] + 8) >> 4;
  } else {
    const int lenS = min(lenP, lenQ), lenL = max(lenP, lenQ);
    const int *S = lenP < lenQ ? P : Q, *L = lenP < lenQ ? Q : P;
    if (lenL == 7 && lenS == 5) mid = (P[5] + P[4] + P[3] + P[2] + 2 * (P[1] + P[0] + Q[0] + Q[1]) + Q[2] + Q[3] + Q[4] + Q[5] + 8) >> 4;
    else if (lenL == 7 && lenS == 3) mid = (3 * S[0] + 2 * L[0] + 3 * S[1] + L[1] + 2 * S[2] + L[2] + L[3] + L[4] + L[5] + L[6] + 8) >> 4;
    else mid = (P[3] + P[2] + P[1] + P[0] + Q[0] + Q[1] + Q[2] + Q[3] + 4) >> 3;
  }
  int nP[7], nQ[7];
  for (int i = 0; i < lenP; ++i) {
    const int c = lenP == 7 ? c7[i] : lenP == 5 ? c5[i] : c3[i];
    const int r = (tc * (lenP == 3 ? t3[i] : t7[i])) >> 1;
    nP[i] = clampi((mid * c + refP * (64 - c) + 32) >> 6, P[i] - r, P[i] + r);
  }
  for (int i = 0; i < lenQ; ++i) {
    const int c = lenQ == 7 ? c7[i] : lenQ == 5 ? c5[i] : c3[i];
    const int r = (tc * (lenQ == 3 ? t3[i] : t7[i])) >> 1;
    nQ[i] = clampi((mid * c + refQ * (64 - c) + 32) >> 6, Q[i] - r, Q[i] + r);
  }
  for (int i = 0; i < lenP; ++i) P[i] = nP[i];
  for (int i = 0; i < lenQ; ++i) Q[i] = nQ[i];
}

template <typename PX>
__device__ inline void luma_segment(PX *plane, int stride, const uvghip_scu_t *scu, int scu_stride, int x, int y, bool dir_hor,
                                    const dbk_cfg &cfg, bool keep_p)
{
  constexpr int DEPTH = px_traits<PX>::depth;
  constexpr int MAXV = px_traits<PX>::maxv;
  const uvghip_scu_t q = scu[(y >> 2) * scu_stride + (x >> 2)];
  const uvghip_scu_t p = dir_hor ? scu[((y >> 2) - 1) * scu_stride + (x >> 2)] : scu[(y >> 2) * scu_stride + (x >> 2) - 1];
  const bool tu_boundary = (q.luma_edges & (dir_hor ? 2 : 1)) != 0;
  const int qp = qp_pred(p, q, cfg);
  const int bs = luma_strength(p, q, tu_boundary, cfg);
  if (!bs) return;
  const int beta = kBeta[clampi(qp + (cfg.beta_offset_div2 << 1), 0, 63)] * (1 << (DEPTH - 8));
  const int side_thr = (beta + (beta >> 1)) >> 3;
  const int tc = tc_from_index<DEPTH>(clampi(qp + 2 * (bs - 1) + (cfg.tc_offset_div2 << 1), 0, 65));
  const int tq = tu_size_luma(q, dir_hor), tp = tu_size_luma(p, dir_hor);
  int lenP, lenQ;
  if (tp <= 4 || tq <= 4) lenP = lenQ = 1;
  else { lenP = tp >= 32 ? 7 : 3; lenQ = tq >= 32 ? 7 : 3; }
  const bool largeP = lenP > 3 && !(dir_hor && (y & 63) == 0);
  const bool largeQ = lenQ > 3;

  const int xs = dir_hor ? stride : 1, ys = dir_hor ? 1 : stride;
  PX *e = plane + (size_t)y * stride + x;
  int P[4][8], Q[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      P[i][k] = (k < 4 || largeP) ? (int)e[i * ys - (k + 1) * xs] : 0;
      Q[i][k] = (k < 4 || largeQ) ? (int)e[i * ys + k * xs] : 0;
    }
  const int dp0 = abs(P[0][2] - 2 * P[0][1] + P[0][0]), dq0 = abs(Q[0][0] - 2 * Q[0][1] + Q[0][2]);
  const int dp3 = abs(P[3][2] - 2 * P[3][1] + P[3][0]), dq3 = abs(Q[3][0] - 2 * Q[3][1] + Q[3][2]);
  const int dp = dp0 + dp3, dq = dq0 + dq3;
  bool sw = false;
  int wP[4] = {0, 0, 0, 0}, wQ[4] = {0, 0, 0, 0};   // samples to write back per line and side

  if (largeP || largeQ) {
    int dp0L = dp0, dq0L = dq0, dp3L = dp3, dq3L = dq3;
    if (largeP) {
      dp0L = (dp0L + abs(P[0][5] - 2 * P[0][4] + P[0][3]) + 1) >> 1;
      dp3L = (dp3L + abs(P[3][5] - 2 * P[3][4] + P[3][3]) + 1) >> 1;
    }
    if (largeQ) {
      dq0L = (dq0L + abs(Q[0][3] - 2 * Q[0][4] + Q[0][5]) + 1) >> 1;
      dq3L = (dq3L + abs(Q[3][3] - 2 * Q[3][4] + Q[3][5]) + 1) >> 1;
    }
    if (dp0L + dp3L + dq0L + dq3L < beta) {
      int sp[2], sq[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int *Pl = P[j * 3], *Ql = Q[j * 3];
        sp[j] = abs(Pl[3] - Pl[0]);
        sq[j] = abs(Ql[0] - Ql[3]);
        if (largeP) {
          int t;
          if (lenP == 7) { t = Pl[7]; sp[j] += abs(Pl[4] - Pl[5] - Pl[6] + t); } else t = Pl[5];
          sp[j] = (sp[j] + abs(Pl[3] - t) + 1) >> 1;
        }
        if (largeQ) {
          int t;
          if (lenQ == 7) { t = Ql[7]; sq[j] += abs(Ql[4] - Ql[5] - Ql[6] + t); } else t = Ql[5];
          sq[j] = (sq[j] + abs(t - Ql[3]) + 1) >> 1;
        }
      }
      sw = 2 * (dp0L + dq0L) < (beta >> 4) && 2 * (dp3L + dq3L) < (beta >> 4) &&
           abs(P[0][0] - Q[0][0]) < ((5 * tc + 1) >> 1) && abs(P[3][0] - Q[3][0]) < ((5 * tc + 1) >> 1) &&
           sp[0] + sq[0] < ((beta * 3) >> 5) && sp[1] + sq[1] < ((beta * 3) >> 5);
      if (sw) {
        const int lp = largeP ? lenP : 3, lq = largeQ ? lenQ : 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) { large_block_line(P[i], Q[i], tc, lp, lq); wP[i] = lp; wQ[i] = lq; }
      }
    }
  }
  if (!sw && dp + dq < beta) {
    if (lenP > 2 && lenQ > 2) sw = strong_normal(P[0], Q[0], P[3], Q[3], dp0, dq0, dp3, dq3, tc, beta, false);
    const bool second = lenP > 1 && lenQ > 1;
    const bool p2nd = second && dp < side_thr, q2nd = second && dq < side_thr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int *Pl = P[i], *Ql = Q[i];
      if (sw) {
        const int m0 = Pl[3], m1 = Pl[2], m2 = Pl[1], m3 = Pl[0], m4 = Ql[0], m5 = Ql[1], m6 = Ql[2], m7 = Ql[3];
        Pl[2] = clampi((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3, m1 - tc, m1 + tc);
        Pl[1] = clampi((m1 + m2 + m3 + m4 + 2) >> 2, m2 - 2 * tc, m2 + 2 * tc);
        Pl[0] = clampi((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3, m3 - 3 * tc, m3 + 3 * tc);
        Ql[0] = clampi((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3, m4 - 3 * tc, m4 + 3 * tc);
        Ql[1] = clampi((m3 + m4 + m5 + m6 + 2) >> 2, m5 - 2 * tc, m5 + 2 * tc);
        Ql[2] = clampi((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3, m6 - tc, m6 + tc);
        wP[i] = 3; wQ[i] = 3;
      } else {
        const int m1 = Pl[2], m2 = Pl[1], m3 = Pl[0], m4 = Ql[0], m5 = Ql[1], m6 = Ql[2];
        int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
        if (abs(delta) < tc * 10) {
          const int tc2 = tc >> 1;
          delta = clampi(delta, -tc, tc);
          Pl[0] = clampi(m3 + delta, 0, MAXV);
          Ql[0] = clampi(m4 - delta, 0, MAXV);
          wP[i] = 1; wQ[i] = 1;
          if (p2nd) { Pl[1] = clampi(m2 + clampi((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1, -tc2, tc2), 0, MAXV); wP[i] = 2; }
          if (q2nd) { Ql[1] = clampi(m5 + clampi((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1, -tc2, tc2), 0, MAXV); wQ[i] = 2; }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (k < wP[i] && !keep_p) e[i * ys - (k + 1) * xs] = (PX)P[i][k];
      if (k < wQ[i]) e[i * ys + k * xs] = (PX)Q[i][k];
    }
}

template <typename PX>
__device__ inline void chroma_segment(PX *pu, PX *pv, int stride, const uvghip_scu_t *scu, int scu_stride, int xc, int yc,
                                      bool dir_hor, const dbk_cfg &cfg, bool keep_p)
{
  constexpr int DEPTH = px_traits<PX>::depth;
  constexpr int MAXV = px_traits<PX>::maxv;
  const int x = xc << 1, y = yc << 1;
  const uvghip_scu_t q = scu[(y >> 2) * scu_stride + (x >> 2)];
  const uvghip_scu_t p = dir_hor ? scu[((y - 1) >> 2) * scu_stride + (x >> 2)] : scu[(y >> 2) * scu_stride + ((x - 1) >> 2)];
  const bool tu_boundary = (q.luma_edges & (dir_hor ? 2 : 1)) != 0;
  const int luma_qp = qp_pred(p, q, cfg);
  const int QP = cfg.has_qp_map ? cfg.qp_map[luma_qp] : clampi(luma_qp, 0, 57);
  const int tp = min(1 << (dir_hor ? p.log2_chroma_height : p.log2_chroma_width), 32);
  const int tq = min(1 << (dir_hor ? q.log2_chroma_height : q.log2_chroma_width), 32);
  const bool large = tp >= 8 && tq >= 8;
  const bool ctb = dir_hor && (y & 63) == 0;
  int bs[2] = {0, 0};
  if (q.type == 1 || p.type == 1) bs[0] = bs[1] = 2;
  else if (tu_boundary) { bs[0] = ((q.cbf | p.cbf) & 2) ? 1 : 0; bs[1] = ((q.cbf | p.cbf) & 4) ? 1 : 0; }
  const int xs = dir_hor ? stride : 1, ys = dir_hor ? 1 : stride;
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    if (!(bs[comp] == 2 || (large && bs[comp] == 1))) continue;
    const int tc = tc_from_index<DEPTH>(clampi(QP + 2 * (bs[comp] - 1) + (cfg.tc_offset_div2 << 1), 0, 65));
    PX *e = (comp ? pv : pu) + (size_t)yc * stride + xc;
    int P[2][4], Q[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        P[i][k] = (k < 2 || large) ? (int)e[i * ys - (k + 1) * xs] : 0;
        Q[i][k] = (k < 2 || large) ? (int)e[i * ys + k * xs] : 0;
      }
    bool sw = false;
    if (large) {
      const int beta = kBeta[clampi(QP + (cfg.beta_offset_div2 << 1), 0, 63)] * (1 << (DEPTH - 8));
      const int pi = ctb ? 1 : 2;
      const int dp0 = abs(P[0][pi] - 2 * P[0][1] + P[0][0]), dq0 = abs(Q[0][0] - 2 * Q[0][1] + Q[0][2]);
      const int dp3 = abs(P[1][pi] - 2 * P[1][1] + P[1][0]), dq3 = abs(Q[1][0] - 2 * Q[1][1] + Q[1][2]);
      if (dp0 + dp3 + dq0 + dq3 < beta) sw = strong_normal(P[0], Q[0], P[1], Q[1], dp0, dq0, dp3, dq3, tc, beta, ctb);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m0 = P[i][3], m1 = P[i][2], m2 = P[i][1], m3 = P[i][0], m4 = Q[i][0], m5 = Q[i][1], m6 = Q[i][2], m7 = Q[i][3];
      if (sw) {
        if (ctb) {
          if (!keep_p) e[i * ys - xs] = (PX)clampi((3 * m2 + 2 * m3 + m4 + m5 + m6 + 4) >> 3, m3 - tc, m3 + tc);
          e[i * ys] = (PX)clampi((2 * m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4) >> 3, m4 - tc, m4 + tc);
        } else {
          if (!keep_p) {
            e[i * ys - 3 * xs] = (PX)clampi((3 * m0 + 2 * m1 + m2 + m3 + m4 + 4) >> 3, m1 - tc, m1 + tc);
            e[i * ys - 2 * xs] = (PX)clampi((2 * m0 + m1 + 2 * m2 + m3 + m4 + m5 + 4) >> 3, m2 - tc, m2 + tc);
            e[i * ys - xs] = (PX)clampi((m0 + m1 + m2 + 2 * m3 + m4 + m5 + m6 + 4) >> 3, m3 - tc, m3 + tc);
          }
          e[i * ys] = (PX)clampi((m1 + m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4) >> 3, m4 - tc, m4 + tc);
        }
        e[i * ys + xs] = (PX)clampi((m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4) >> 3, m5 - tc, m5 + tc);
        e[i * ys + 2 * xs] = (PX)clampi((m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4) >> 3, m6 - tc, m6 + tc);
      } else {
        const int delta = clampi((((m4 - m3) * 4) + m2 - m5 + 4) >> 3, -tc, tc);
        if (!keep_p) e[i * ys - xs] = (PX)clampi(m3 + delta, 0, MAXV);
        e[i * ys] = (PX)clampi(m4 - delta, 0, MAXV);
      }
    }
  }
}
