// Rate-distortion optimised quantisation on gfx950: uvg_rdoq (src/rdo.c:1449-1870) for n transform blocks of one shape.
//
// What is sequential in the reference's walk (reverse scan order) and what is not:
//   * a level decision (uvg_get_coded_level, :597-640) reads only the levels already decided to the right / below
//     (context selection, :1400-1438) -- all of them lie on LATER anti-diagonals of the block, so the positions of one
//     anti-diagonal of a 4x4 coefficient group are independent of each other;
//   * the Rice parameter a position sees is context-free while regular bins remain: templateAbsSum runs over the INPUT
//     coefficients there (:1697), so it is computed up front for every position in parallel;
//   * the running double-precision costs (:1689, :1705-1716) and the coefficient-group decision that tests them
//     (:1719-1772) must be summed in scan order -- floating-point addition does not re-associate and the result has to be
//     bit-exact; they do not feed back into the decisions inside the group;
//   * the regular-bin budget (:1696) only matters once it is nearly spent: a group that could reach "fewer than 4 bins
//     left" is walked position by position instead.
// Mapping: 16 lanes = the 16 scan positions of the current coefficient group of one block, four blocks per wave.  Per
// group: seven anti-diagonal phases of parallel decisions (levels exchanged through LDS), then every lane replays the
// 16 costs in scan order (same values in all 16 lanes, no broadcast needed) and takes the group decision.  Per-position
// cost_coeff / cost_sig for the last-position search (:1786-1823) stay in LDS (cost_sig as the code of the table entry
// it was built from).  All cost arithmetic is IEEE double in the reference's operation order; this file is compiled with
// -ffp-contract=off, and the one transcendental (pow(2, -2 * transform_shift), :1527) is evaluated on the host.
#include "uvghip_common.h"
#include "vvc_rdoq_tables.h"
#include <cmath>
#include <mutex>

namespace {

enum : int { O_SIGGRP = 0, O_SIG = 4, O_PAR = 28, O_GT1 = 70, O_GT2 = 112, O_LASTX = 154, O_LASTY = 194, O_CBF_Y = 234, O_CBF_CB = 238,
             O_CBF_CR = 240, O_ROOT = 243, N_CTX = 244 };
static_assert(sizeof(uvghip_rdoq_ctx_t) == N_CTX, "uvghip_rdoq_ctx_t layout");

struct rdoq_params {
  int width, height, l2w, l2h, n;
  int color, block_type, cbf_u, lfnst_idx, mts_idx;
  int q_bits, q;
  double lambda, error_scale;
  uvghip_rdoq_ctx_t ctx;
};

__device__ __forceinline__ int group_idx(int pos)
{
  if (pos < 4) return pos;
  const int l = 31 - __clz(pos);
  return 2 * l + ((pos >> (l - 1)) & 1);
}

__device__ __forceinline__ int go_rice_par(int s) { return (s >= 7) + (s >= 14) + (s >= 28); }

// rdo.c:465-581 with use_limited_prefix_length = true (the only way uvg_get_coded_level calls it)
__device__ __forceinline__ int ic_rate(const uint32_t (*B)[2], int t, uint32_t abs_level, int ctx, int go_rice, uint32_t reg_bins)
{
  int rate = 1 << 15;
  const int thr = 5, max_log2 = 15;
  if (reg_bins < 4) {
    const uint32_t zero = 1u << go_rice;
    const uint32_t symbol = (abs_level == 0 ? zero : abs_level <= zero ? abs_level - 1 : abs_level);
    if (symbol < ((uint32_t)thr << go_rice)) {
      rate += (int)(((symbol >> go_rice) + 1 + go_rice) << 15);
    } else {
      const uint32_t max_prefix = 32 - (thr + max_log2);
      uint32_t prefix = 0;
      const uint32_t suffix = (symbol >> go_rice) - thr;
      while (prefix < max_prefix && (int)suffix > ((2 << prefix) - 2)) prefix++;
      const uint32_t suffix_len = prefix == max_prefix ? (uint32_t)(max_log2 - go_rice) : prefix + 1;
      rate += (int)((thr + prefix + suffix_len + go_rice) << 15);
    }
    return rate;
  }
  const int par = O_PAR + 21 * t + ctx, gt1 = O_GT1 + 21 * t + ctx, gt2 = O_GT2 + 21 * t + ctx;
  if (abs_level >= 4) {
    const int symbol = (int)abs_level - 4;
    if (symbol < (thr << go_rice)) {
      rate += ((symbol >> go_rice) + 1 + go_rice) << 15;
    } else {
      const uint32_t max_prefix = 32 - (thr + max_log2);
      uint32_t prefix = 0;
      const uint32_t suffix = (uint32_t)(symbol >> go_rice) - thr;
      while (prefix < max_prefix && (int)suffix > ((2 << prefix) - 2)) prefix++;
      const uint32_t suffix_len = prefix == max_prefix ? (uint32_t)(max_log2 - go_rice) : prefix + 1;
      rate += (int)((thr + prefix + suffix_len + go_rice) << 15);
    }
    rate += (int)B[par][(abs_level - 2) & 1];
    rate += (int)B[gt1][1];
    rate += (int)B[gt2][1];
  } else if (abs_level == 1) {
    rate += (int)B[gt1][0];
  } else if (abs_level == 2) {
    rate += (int)B[par][0]; rate += (int)B[gt1][1]; rate += (int)B[gt2][0];
  } else if (abs_level == 3) {
    rate += (int)B[par][1]; rate += (int)B[gt1][1]; rate += (int)B[gt2][0];
  } else {
    rate = 0;
  }
  return rate;
}

struct rdoq_decision { int level; int sig_code; double coded_cost, coded_sig; };

// uvg_get_coded_level (rdo.c:597-640) + the context derivation in front of it (:1630-1651) for one position.
// nb / has: levels of the neighbours right, right+1, below-right, below, below+1 (0 where outside the block).
__device__ __forceinline__ rdoq_decision rdoq_decide(const rdoq_params &P, const uint32_t (*B)[2], int t, bool is_last, int level_double, uint32_t max_abs_level,
                                            double c0, const int (&nb)[5], const bool (&has)[5], uint32_t pos_x, uint32_t pos_y, int go_rice,
                                            uint32_t reg_bins)
{
  const int mts = P.mts_idx;
  const double lambda = P.lambda, error_scale = P.error_scale;
  int ctx_sig = 0, ctx_set = 0;
  if (!is_last) {
    // zero-out tests of context_get_sig_ctx_idx_abs (the "below" terms test pos_x, as the reference does, rdo.c:1425)
    const bool zz[5] = {mts && pos_x + 1 >= 16, mts && pos_x + 2 >= 16, mts && (pos_y + 1 >= 16 || pos_x + 1 >= 16),
                        mts && pos_x + 1 >= 16, mts && pos_x + 2 >= 16};
    int num_pos = 0, sum_abs = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (has[k]) { const int a = zz[k] ? 0 : abs(nb[k]); sum_abs += min(4 + (a & 1), a); num_pos += a ? 1 : 0; }
    const int diag = (int)(pos_x + pos_y);
    ctx_sig = min((sum_abs + 1) >> 1, 3) + (diag < 2 ? 4 : 0);
    if (P.color == 0) ctx_sig += diag < 5 ? 4 : 0;
    const int temp_sum = sum_abs - num_pos;
    ctx_set = (min(temp_sum, 4) + 1) + (!diag ? ((P.color == 0) ? 15 : 5) : (P.color == 0) ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
  }                                               // the last significant position is the first one visited: temp_diag == -1, ctx_set = 0
  if (reg_bins < 4) {                             // templateAbsSum over the decided levels (its zero-out tests use pos_y for "below")
    const bool zz[5] = {mts && pos_x + 1 >= 16, mts && pos_x + 2 >= 16, mts && (pos_y + 1 >= 16 || pos_x + 1 >= 16),
                        mts && pos_y + 1 >= 16, mts && pos_y + 2 >= 16};
    int16_t sum = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) if (has[k]) sum = (int16_t)(sum + (zz[k] ? 0 : abs(nb[k])));
    go_rice = go_rice_par(clampi((int)sum, 0, 31));
  }
  rdoq_decision d;
  d.level = 0; d.sig_code = 0; d.coded_sig = 0;
  double cur_cost_sig = 0;
  bool done = false;
  if (!is_last && max_abs_level < 3) {
    d.coded_sig = lambda * (double)B[O_SIG + 12 * t + ctx_sig][0];
    d.sig_code = 1 + 2 * ctx_sig;
    d.coded_cost = c0 + d.coded_sig;
    if (max_abs_level == 0) done = true;
  } else {
    d.coded_cost = 1.7e+308;
  }
  if (!done) {
    if (!is_last) cur_cost_sig = lambda * (double)B[O_SIG + 12 * t + ctx_sig][1];
    const int min_abs = max_abs_level > 1 ? (int)max_abs_level - 1 : 1;
    for (int a = (int)max_abs_level; a >= min_abs; a--) {
      const double err = (double)(level_double - (a * (1 << P.q_bits)));
      double cur = err * err * error_scale + lambda * (double)ic_rate(B, t, (uint32_t)a, ctx_set, go_rice, reg_bins);
      cur += cur_cost_sig;
      if (cur < d.coded_cost) { d.level = a; d.coded_cost = cur; d.coded_sig = cur_cost_sig; d.sig_code = is_last ? 0 : 2 + 2 * ctx_sig; }
    }
  }
  return d;
}

// Four lanes per block (an anti-diagonal of a 4x4 group has at most four positions), TUS blocks per wave: 16 for the small
// shapes (plenty of blocks: lane utilisation matters), 4 / 1 for 256 / 512+ coefficients (few blocks, long chains: more
// waves per SIMD hide the latency of the dependent steps).
template <int TUS>
__global__ void __launch_bounds__(64)
rdoq_kernel(const rdoq_params Pk, const int16_t *__restrict__ coef, int16_t *__restrict__ q_coef, double *__restrict__ ws,
            uint32_t *__restrict__ abs_sum_out, uint8_t *__restrict__ has_coeffs)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char sDyn[];
  // the parameter block is read through LDS: kept in scalar registers for the whole kernel it pushes the SGPR file over
  // its limit (spills + a scratch slot, which makes every dispatch bind scratch memory)
  __shared__ rdoq_params sP;
  for (int i = threadIdx.x; i < (int)(sizeof(rdoq_params) / 4); i += 64)
    reinterpret_cast<uint32_t *>(&sP)[i] = reinterpret_cast<const uint32_t *>(&Pk)[i];
  __syncthreads();
  const rdoq_params &P = sP;
  __shared__ uint32_t sB[N_CTX][2];
  __shared__ int sLastX[32], sLastY[32];
  __shared__ uint8_t sScanCg[64];
  __shared__ double sStage[TUS][16][3];                                // per group: coded_cost, coded_sig, cost0 of the 16 positions
  __shared__ int sStageLv[TUS][16];
  const int tid = threadIdx.x, grp = tid >> 2, j = tid & 3;
  const int width = P.width, height = P.height, wh = width * height, l2w = P.l2w;
  const int n = P.n;
  const int tu0 = blockIdx.x * TUS;
  const int here = min(TUS, n - tu0);
  const bool live = grp < here;
  const int gq = live ? grp : 0;                                       // idle lanes alias block 0 for address arithmetic only
  const int tu = tu0 + gq;
  // per block in LDS: coef int16[wh] | level int16[wh] | meta uint8[wh] (padded to 8 bytes); then 64 group costs per block.
  // cost_coeff[] (a double per position, written once during the walk, re-read only for non-zero levels by the last-position
  // search) goes to the caller's workspace: it would be 60 % of the LDS footprint and cap the blocks per wave.
  const size_t per_tu = ((size_t)wh * 5 + 7) & ~(size_t)7;
  double *gCost = ws + (size_t)tu * wh;
  int16_t *sCoef = reinterpret_cast<int16_t *>(sDyn + gq * per_tu);
  int16_t *sLev = sCoef + wh;
  uint8_t *sMeta = reinterpret_cast<uint8_t *>(sLev + wh);
  double *sCgCost = reinterpret_cast<double *>(sDyn + TUS * per_tu) + gq * 64;
  const int t = P.color ? 1 : 0;
  const int mts = P.mts_idx;

  // ---- per-workgroup tables ----
  {
    const uint8_t *st = reinterpret_cast<const uint8_t *>(&P.ctx);
    for (int i = tid; i < N_CTX; i += 64) { const int s = st[i]; sB[i][0] = kEntropyBits[2 * s]; sB[i][1] = kEntropyBits[2 * s + 1]; }
  }
  const int l2cgw = l2w - 2, cgw = 1 << l2cgw, cgh = height >> 2;
  if (tid == 0) {                                                      // H.266 6.5.2 on the coefficient-group grid
    int i = 0, x = 0, y = 0;
    while (i < cgw * cgh) {
      while (y >= 0) { if (x < cgw && y < cgh) sScanCg[i++] = (uint8_t)(y * cgw + x); y--; x++; }
      y = x; x = 0;
    }
  }
  // ---- stage the coefficients (coalesced: the blocks of a workgroup are contiguous), clear the levels ----
  for (int e = tid; e < here * wh; e += 64) {
    const int b = e / wh, pos = e - b * wh;
    int16_t *base = reinterpret_cast<int16_t *>(sDyn + b * per_tu);
    base[pos] = coef[(size_t)tu0 * wh + e];
    base[wh + pos] = 0;
  }
  __syncthreads();
  if (tid == 0) {                                                      // calc_last_bits, rdo.c:667-700
    auto prefix_ctx = [](int l2) { return l2 <= 2 ? 0 : l2 == 3 ? 3 : l2 == 4 ? 6 : 10; };   // {0,0,0,3,6,10,15,21}[log2 size]
    const int l2h = P.l2h;
    const int ox = t ? 0 : prefix_ctx(l2w), oy = t ? 0 : prefix_ctx(l2h);
    const int sx = t ? clampi(width >> 3, 0, 2) : ((l2w + 1) >> 2), sy = t ? clampi(height >> 3, 0, 2) : ((l2h + 1) >> 2);
    int bits = 0, c;
    for (c = 0; c < group_idx(width - 1); ++c) {
      const int o = O_LASTX + 20 * t + ox + (c >> sx);
      sLastX[c] = bits + (int)sB[o][0]; bits += (int)sB[o][1];
    }
    sLastX[c] = bits;
    bits = 0;
    for (c = 0; c < group_idx(height - 1); ++c) {
      const int o = O_LASTY + 20 * t + oy + (c >> sy);
      sLastY[c] = bits + (int)sB[o][0]; bits += (int)sB[o][1];
    }
    sLastY[c] = bits;
  }
  // the Rice parameter after each position: templateAbsSum(coef, 4, ...) over the input block (rdo.c:846-871, 1697)
  if (live)
    for (int pos = j; pos < wh; pos += 4) {
      const uint32_t pos_y = (uint32_t)pos >> l2w, pos_x = (uint32_t)pos - (pos_y << l2w);
      const int16_t *c0p = sCoef + pos;
      int16_t sum = 0;                                                 // coeff_t accumulator: wraps like the reference's
      if (pos_x < (uint32_t)width - 1) {
        sum = (int16_t)(sum + ((mts && pos_x + 1 >= 16) ? 0 : abs((int)c0p[1])));
        if (pos_x < (uint32_t)width - 2) sum = (int16_t)(sum + ((mts && pos_x + 2 >= 16) ? 0 : abs((int)c0p[2])));
        if (pos_y < (uint32_t)height - 1) sum = (int16_t)(sum + ((mts && (pos_y + 1 >= 16 || pos_x + 1 >= 16)) ? 0 : abs((int)c0p[width + 1])));
      }
      if (pos_y < (uint32_t)height - 1) {
        sum = (int16_t)(sum + ((mts && pos_y + 1 >= 16) ? 0 : abs((int)c0p[width])));
        if (pos_y < (uint32_t)height - 2) sum = (int16_t)(sum + ((mts && pos_y + 2 >= 16) ? 0 : abs((int)c0p[2 * width])));
      }
      sMeta[pos] = (uint8_t)go_rice_par(clampi((int)sum - 20, 0, 31));
    }
  __syncthreads();

  // in-group up-right diagonal order of a 4x4 group as (y * 4 + x) nibbles, scan position 0 first
  constexpr unsigned long long kDiag4 = 0xFBE7AD369C258140ull;
  auto in_cg = [&](int s4) { return (int)((kDiag4 >> (4 * s4)) & 15); };
  auto blk_in = [&](int g, int k) { return ((((g >> l2cgw) << 2) + (k >> 2)) << l2w) + ((g & (cgw - 1)) << 2) + (k & 3); };
  const uint32_t (*B)[2] = sB;
  const double lambda = P.lambda, error_scale = P.error_scale;
  const int q_bits = P.q_bits, q = P.q;
  const int cap = 0x7fffffff - (1 << (q_bits - 1));
  const uint32_t cg_width = (uint32_t)min(width, 32) >> 2, cg_height = (uint32_t)min(height, 32) >> 2;
  const int cg_num = P.lfnst_idx > 0 ? 1 : wh >> 4;
  const int max_group = P.lfnst_idx > 0 ? (((height == 4 && width == 4) || (height == 8 && width == 8)) ? 7 : 15) : 15;
  auto level_double_at = [&](int blkpos) {
    const long long prod = (long long)abs((int)sCoef[blkpos]) * q;
    return (int)(prod < cap ? prod : cap);
  };
  auto cost0_of = [&](int level_double) { const double err = (double)level_double; return err * err * error_scale; };   // cost_coeff0[]
  auto sig_cost_of = [&](int code) { return code ? lambda * (double)B[O_SIG + 12 * t + ((code - 1) >> 1)][(code - 1) & 1] : 0.0; };   // cost_sig[]
  auto cg_skipped = [&](int g) { return mts != 0 && ((g >> l2cgw) >= 4 || (g & (cgw - 1)) >= 4); };
  auto quad_or = [&](unsigned v) { v |= __shfl_xor(v, 1, 64); v |= __shfl_xor(v, 2, 64); return v; };
  auto quad_sum = [&](unsigned v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); return v; };
  auto neighbours = [&](int blkpos, uint32_t pos_x, uint32_t pos_y, int (&nb)[5], bool (&has)[5]) __attribute__((always_inline)) {
    const int16_t *D = sLev + blkpos;
#pragma unroll
    for (int k = 0; k < 5; ++k) { nb[k] = 0; has[k] = false; }
    if (pos_x < (uint32_t)width - 1) {
      has[0] = true; nb[0] = D[1];
      if (pos_x < (uint32_t)width - 2) { has[1] = true; nb[1] = D[2]; }
      if (pos_y < (uint32_t)height - 1) { has[2] = true; nb[2] = D[width + 1]; }
    }
    if (pos_y < (uint32_t)height - 1) {
      has[3] = true; nb[3] = D[width];
      if (pos_y < (uint32_t)height - 2) { has[4] = true; nb[4] = D[2 * width]; }
    }
  };

  // ---- the last significant position (rdo.c:1561-1592): highest scan position whose rounded level is non-zero ----
  int last_scanpos = -1;
  for (int cgs = cg_num - 1; cgs >= 0; --cgs) {                        // uniform trip count; the blocks differ only in predicates
    const int g = sScanCg[cgs];
    unsigned m = 0;
    if (live && last_scanpos < 0 && !cg_skipped(g))
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r;
        if (s4 <= max_group) {
          const int ld = level_double_at(blk_in(g, in_cg(s4)));
          if (((uint32_t)(ld + (1 << (q_bits - 1))) >> q_bits) > 0) m |= 1u << s4;
        }
      }
    m = quad_or(m);
    if (last_scanpos < 0 && m) last_scanpos = cgs * 16 + (31 - __clz((int)m));
  }
  const int cg_last_scanpos = last_scanpos >> 4;                        // -1 >> 4 == -1

  // ---- the walk, coefficient group by coefficient group (rdo.c:1604-1773) ----
  unsigned long long sig_cg = 0;                                       // sig_coeffgroup_flag as a bit set over group raster positions
  double block_uncoded_cost = 0, base_cost = 0;
  uint32_t reg_bins = (uint32_t)(wh * 28) >> 4;
  int go_rice_state = 0;                                               // only tracked once a group is walked sequentially
  bool slow = false;
  for (int cgs = cg_num - 1; cgs >= 0; --cgs) {
    const int g = sScanCg[cgs];
    const bool has_last = live && last_scanpos >= 0;
    const bool in_tail = has_last && cgs > cg_last_scanpos && !cg_skipped(g);   // beyond the last significant position: cost0 sums only (:1585)
    const bool in_walk = has_last && cgs <= cg_last_scanpos && !cg_skipped(g);
    // every position of the group: rounded level into dest_coeff (:1620), cost0 staged, bins this group can spend at most
    unsigned spend = 0;
    if (in_tail || in_walk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r;
        if (s4 > max_group) continue;
        const int blkpos = blk_in(g, in_cg(s4));
        const int ld = level_double_at(blkpos);
        const uint32_t mx = (uint32_t)(ld + (1 << (q_bits - 1))) >> q_bits;
        sStage[gq][s4][2] = cost0_of(ld);
        sStageLv[gq][s4] = 0;
        if (in_walk && cgs * 16 + s4 <= last_scanpos) { sLev[blkpos] = (int16_t)mx; spend += (mx < 2 ? mx : 3) + 1; }
      }
    spend = quad_sum(spend);
    // a group in which the regular-bin budget could fall below 4 is walked position by position, and so is everything after it
    if (in_walk && !slow && reg_bins < 4 + spend) slow = true;
    const bool any_slow = __any(in_walk && slow);
    __syncthreads();
    // -- fast path: the positions of one anti-diagonal decide together (their neighbours lie on later anti-diagonals) --
    for (int dg = 6; dg >= 0; --dg) {
      const int cnt = dg <= 3 ? dg + 1 : 7 - dg;
      const int s4 = (dg <= 3 ? dg * (dg + 1) / 2 : 16 - (7 - dg) * (8 - dg) / 2) + j;   // scan index inside the group
      const int scanpos = cgs * 16 + s4;
      if (in_walk && !slow && j < cnt && s4 <= max_group && scanpos <= last_scanpos) {
        const int blkpos = blk_in(g, in_cg(s4));
        const uint32_t pos_y = (uint32_t)blkpos >> l2w, pos_x = (uint32_t)blkpos - (pos_y << l2w);
        const int ld = level_double_at(blkpos);
        const uint32_t mx = (uint32_t)(ld + (1 << (q_bits - 1))) >> q_bits;
        const bool is_last = scanpos == last_scanpos;
        int nb[5]; bool has[5];
        neighbours(blkpos, pos_x, pos_y, nb, has);
        // Rice parameter left by the previously visited position (scanpos + 1): reset after every 16th (:1692), else the
        // context-free value of that position; the last significant position starts with 0
        const int go_rice = (is_last || s4 == 15) ? 0 : (sMeta[blk_in(g, in_cg(s4 + 1))] & 3);
        const rdoq_decision d = rdoq_decide(P, B, t, is_last, ld, mx, cost0_of(ld), nb, has, pos_x, pos_y, go_rice, 4);
        sLev[blkpos] = (int16_t)d.level;
        sStage[gq][s4][0] = d.coded_cost; sStage[gq][s4][1] = d.coded_sig;
        sStageLv[gq][s4] = d.level;
        gCost[blkpos] = d.coded_cost;
        sMeta[blkpos] = (uint8_t)((sMeta[blkpos] & 3) | (d.sig_code << 2));
      }
      __syncthreads();
    }
    // -- slow path: the lanes of the block walk the group's positions in order (identical results in all four) --
    if (any_slow) {
      for (int s2 = max_group; s2 >= 0; --s2) {
        const int sc2 = cgs * 16 + s2;
        const bool act2 = in_walk && slow && sc2 <= last_scanpos;
        rdoq_decision d2;
        d2.level = 0; d2.sig_code = 0; d2.coded_cost = 0; d2.coded_sig = 0;
        const int b2 = blk_in(g, in_cg(s2));
        if (act2) {
          const uint32_t py = (uint32_t)b2 >> l2w, px = (uint32_t)b2 - (py << l2w);
          const int ld2 = level_double_at(b2);
          const uint32_t mx2 = (uint32_t)(ld2 + (1 << (q_bits - 1))) >> q_bits;
          int nb[5]; bool has[5];
          neighbours(b2, px, py, nb, has);
          const bool last2 = sc2 == last_scanpos;
          d2 = rdoq_decide(P, B, t, last2, ld2, mx2, cost0_of(ld2), nb, has, px, py, go_rice_state, reg_bins);
          // context set update (rdo.c:1691-1699), tracked here because the budget is nearly spent
          if ((sc2 % 16 == 0) && sc2 > 0) go_rice_state = 0;
          else if (reg_bins >= 4) {
            reg_bins -= (uint32_t)((d2.level < 2 ? d2.level : 3) + (last2 ? 0 : 1));
            go_rice_state = sMeta[b2] & 3;
          }
        }
        __syncthreads();
        if (act2 && j == 0) {
          sLev[b2] = (int16_t)d2.level;
          sStage[gq][s2][0] = d2.coded_cost; sStage[gq][s2][1] = d2.coded_sig;
          sStageLv[gq][s2] = d2.level;
          gCost[b2] = d2.coded_cost;
          sMeta[b2] = (uint8_t)((sMeta[b2] & 3) | (d2.sig_code << 2));
        }
        __syncthreads();
      }
    }
    // -- replay the group's costs in scan order: every lane of the block computes the same sums --
    if (in_tail) {
      for (int s2 = max_group; s2 >= 0; --s2) { const double v = sStage[gq][s2][2]; block_uncoded_cost += v; base_cost += v; }
    } else if (in_walk) {
      double rd_coded = 0, rd_uncoded = 0, rd_sig = 0, rd_sig0 = 0;
      int nnz_before_pos0 = 0;
      bool any_level = false;
      for (int s2 = max_group; s2 >= 0; --s2) {
        const int sc2 = cgs * 16 + s2;
        const double v0 = sStage[gq][s2][2];
        if (sc2 > last_scanpos) {                                        // trailing zeros of the last group (:1585-1586)
          block_uncoded_cost += v0; base_cost += v0;
          continue;
        }
        const double cc = sStage[gq][s2][0], cs = sStage[gq][s2][1];
        const int lv = sStageLv[gq][s2];
        block_uncoded_cost += v0;
        base_cost += cc;
        if (!slow) {                                                     // budget bookkeeping of the fast path (:1692-1697)
          if (!((sc2 % 16 == 0) && sc2 > 0) && reg_bins >= 4) reg_bins -= (uint32_t)((lv < 2 ? lv : 3) + (sc2 == last_scanpos ? 0 : 1));
        }
        rd_sig += cs;
        if (s2 == 0) rd_sig0 = cs;
        if (lv) {
          any_level = true;
          rd_coded += cc - cs;
          rd_uncoded += v0;
          if (s2 != 0) nnz_before_pos0++;
        }
      }
      if (any_level) sig_cg |= 1ull << g;
      // coefficient-group decision (rdo.c:1719-1772)
      double cg_cost = 0;
      bool zero_it = false;
      if (cgs) {
        const uint32_t cg_pos_y = (uint32_t)g >> l2cgw, cg_pos_x = (uint32_t)g & (cgw - 1);
        uint32_t right = 0, lower = 0;
        if (cg_pos_x + 1 < cg_width) right = (uint32_t)(sig_cg >> (g + 1)) & 1;
        if (cg_pos_y + 1 < cg_height) lower = (uint32_t)(sig_cg >> (g + cg_width)) & 1;
        const int csx = O_SIGGRP + 2 * t + ((right || lower) ? 1 : 0);
        if (!any_level) {
          cg_cost = lambda * (double)B[csx][0];
          base_cost += cg_cost - rd_sig;
        } else if (cgs < cg_last_scanpos) {
          if (nnz_before_pos0 == 0) { base_cost -= rd_sig0; rd_sig -= rd_sig0; }
          double cost_zero_cg = base_cost;
          cg_cost = lambda * (double)B[csx][1];
          base_cost += cg_cost;
          cost_zero_cg += lambda * (double)B[csx][0];
          cost_zero_cg += rd_uncoded;
          cost_zero_cg -= rd_coded;
          cost_zero_cg -= rd_sig;
          if (cost_zero_cg < base_cost) {
            sig_cg &= ~(1ull << g);
            base_cost = cost_zero_cg;
            cg_cost = lambda * (double)B[csx][0];
            zero_it = true;
          }
        }
      } else {
        sig_cg |= 1ull << g;
      }
      if (j == 0) sCgCost[cgs] = cg_cost;                                // cost_coeffgroup_sig[cgs]
      if (zero_it)                                                       // reset the group (:1752-1762): cost_coeff = cost_coeff0, cost_sig = 0
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s4 = j + 4 * r;
          const int blkpos = blk_in(g, in_cg(s4));
          if (s4 <= max_group && sLev[blkpos]) { sLev[blkpos] = 0; gCost[blkpos] = sStage[gq][s4][2]; sMeta[blkpos] &= 3; }
        }
    } else if (has_last && j == 0) {
      sCgCost[cgs] = 0;                                                  // groups skipped by the MTS zero-out keep a zero flag cost
    }
    __syncthreads();
  }

  // ---- last position (rdo.c:1775-1829): one short sequential pass, identical in the lanes of a block ----
  int best_last_idx_p1 = 0;
  if (live && last_scanpos >= 0) {
    double best_cost;
    if (P.block_type != 1 && !P.color) {
      best_cost = block_uncoded_cost + lambda * (double)B[O_ROOT][0];
      base_cost += lambda * (double)B[O_ROOT][1];
    } else {
      const int m = P.color == 0 ? O_CBF_Y : P.color == 1 ? O_CBF_CB : O_CBF_CR + (P.cbf_u ? 1 : 0);
      best_cost = block_uncoded_cost + lambda * (double)B[m][0];
      base_cost += lambda * (double)B[m][1];
    }
    bool found_last = false;
    for (int cgs = cg_last_scanpos; cgs >= 0 && !found_last; cgs--) {
      const int g = sScanCg[cgs];
      base_cost -= sCgCost[cgs];
      if ((sig_cg >> g) & 1) {
        // the group's cost_coeff values: fetched by the four lanes together, then read from LDS by each of them
        // (same-wave LDS traffic is in order: no barrier inside this block-divergent loop)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int s4 = j + 4 * r; if (s4 <= max_group) sStage[gq][s4][0] = gCost[blk_in(g, in_cg(s4))]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int s2 = max_group; s2 >= 0; s2--) {
          const int sc2 = cgs * 16 + s2;
          if (sc2 > last_scanpos) continue;
          const int b2 = blk_in(g, in_cg(s2));
          const int lv = sLev[b2];
          const double csig = sig_cost_of(sMeta[b2] >> 2);
          if (lv) {
            const uint32_t py = (uint32_t)b2 >> l2w, px = (uint32_t)b2 - (py << l2w);
            const uint32_t cx = (uint32_t)group_idx((int)px), cy = (uint32_t)group_idx((int)py);
            double ui = (double)(sLastX[cx] + sLastY[cy]);
            if (cx > 3) ui += (double)(32768u * ((cx - 2) >> 1));
            if (cy > 3) ui += (double)(32768u * ((cy - 2) >> 1));
            const double cost_last = lambda * ui;
            const double total = base_cost + cost_last - csig;
            if (total < best_cost) { best_last_idx_p1 = sc2 + 1; best_cost = total; }
            if (lv > 1) { found_last = true; break; }
            base_cost -= sStage[gq][s2][0];
            base_cost += cost0_of(level_double_at(b2));
          } else {
            base_cost -= csig;
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- signs, clean-up, outputs (rdo.c:1831-1858): parallel over the positions ----
  uint32_t my_abs = 0;
  if (live) {
    const bool reduce = mts && !(width < 32 && height < 32);
    for (int cgs = 0; cgs < (wh >> 4); ++cgs)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r, scanpos = cgs * 16 + s4;
        const int b = blk_in(sScanCg[cgs], in_cg(s4));
        int level = sLev[b];
        if (last_scanpos < 0 || scanpos >= best_last_idx_p1) level = 0;
        else if (reduce) { const int bx = b & (width - 1), by = b >> l2w; if (bx >= 16 || by >= 16) level = 0; }
        my_abs += (uint32_t)level;
        sLev[b] = (int16_t)((level != 0 && sCoef[b] < 0) ? -level : level);
      }
  }
  my_abs = quad_sum(my_abs);
  if (live && j == 0) {
    if (abs_sum_out) abs_sum_out[tu] = my_abs;
    if (has_coeffs) has_coeffs[tu] = my_abs ? 1 : 0;
  }
  __syncthreads();
  for (int e = tid; e < here * wh; e += 64) {
    const int b = e / wh, pos = e - b * wh;
    q_coef[(size_t)tu0 * wh + e] = reinterpret_cast<const int16_t *>(sDyn + b * per_tu)[wh + pos];
  }
}

}  // namespace

extern "C" size_t uvghip_rdoq_workspace_bytes(int width, int height, int n)
{
  if (width <= 0 || height <= 0 || n <= 0) return 0;
  return (size_t)width * height * (size_t)n * sizeof(double);      // cost_coeff[] of every block
}

extern "C" int uvghip_rdoq_batch(int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n, int color,
                                 int block_type, int cbf_u, int lfnst_idx, int mts_idx, int qp_scaled, double lambda,
                                 const uvghip_rdoq_ctx_t *ctx_host, void *workspace, size_t workspace_bytes,
                                 uint32_t *abs_sum_out, uint8_t *has_coeffs, void *stream)
{
  UVGHIP_REQUIRE_READY();
  auto pow2 = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32; };
  if ((bitdepth != 8 && bitdepth != 10) || !pow2(width) || !pow2(height) || color < 0 || color > 2 || !ctx_host || !coef || !q_coef ||
      qp_scaled < 0 || lfnst_idx < 0 || lfnst_idx > 2 || mts_idx < 0 || !(lambda >= 0))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  if (!workspace || workspace_bytes < uvghip_rdoq_workspace_bytes(width, height, n)) return uvghip_set_error(hipErrorInvalidValue, "uvghip_rdoq_batch: workspace");
  if (((uintptr_t)coef | (uintptr_t)q_coef) & 1) return uvghip_set_error(hipErrorInvalidValue, __func__);
  rdoq_params P;
  P.width = width; P.height = height; P.n = n;
  P.l2w = 31 - __builtin_clz(width); P.l2h = 31 - __builtin_clz(height);
  P.color = color; P.block_type = block_type; P.cbf_u = cbf_u; P.lfnst_idx = lfnst_idx; P.mts_idx = mts_idx;
  const int sqrt2 = (P.l2w + P.l2h) & 1;
  const int transform_shift = 15 - bitdepth - ((P.l2w + P.l2h) >> 1);
  P.q_bits = 14 + qp_scaled / 6 + transform_shift - sqrt2;
  static const int scales[2][6] = {{26214, 23302, 20560, 18396, 16384, 14564}, {18396, 16384, 14564, 13107, 11651, 10280}};   // uvg_g_quant_scales (scalinglist.c:91)
  P.q = scales[sqrt2][qp_scaled % 6];
  if (P.q_bits < 1 || P.q_bits > 30) return uvghip_set_error(hipErrorInvalidValue, "uvghip_rdoq_batch: q_bits");
  // rdo.c:1523-1529: evaluated on the host in double, exactly as the reference does
  const double d_trans_shift = (double)transform_shift + (sqrt2 ? -0.5 : 0.0);
  double scale = 32768;
  scale = scale * pow(2.0, -2.0 * d_trans_shift);
  P.error_scale = scale / P.q / P.q;
  P.lambda = lambda;
  P.ctx = *ctx_host;
  const int wh = width * height;
  // per block: cost_coeff (double) + coefficients + levels (int16) + meta (byte) per position; then 64 group costs per block
  // blocks per wave (four lanes each).  The kernel is issue-bound (SQ counters, tools/dev/rdoq_pmc.sh: ~1900 instructions per
  // coefficient group and wave, VALU active > 50 %); a wave's work does not depend on how many blocks share it, so more blocks
  // per wave = the same latency at a fraction of the GPU time.  16 while the LDS allows (<= 256 coefficients), else 4.
  const int tus = wh <= 512 ? 16 : 8;
  const size_t per_tu = ((size_t)wh * 5 + 7) & ~(size_t)7;
  const size_t lds = (size_t)tus * per_tu + (size_t)tus * 64 * sizeof(double);
  double *w = static_cast<double *>(workspace);
  hipStream_t st = uvghip_stream(stream);
  const int grid = (n + tus - 1) / tus;
  if (tus == 16) rdoq_kernel<16><<<grid, 64, lds, st>>>(P, coef, q_coef, w, abs_sum_out, has_coeffs);
  else rdoq_kernel<8><<<grid, 64, lds, st>>>(P, coef, q_coef, w, abs_sum_out, has_coeffs);
  UVGHIP_CHECK_LAUNCH();
}
