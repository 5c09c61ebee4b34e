// Rate-distortion optimised quantisation on gfx950: uvg_rdoq (src/rdo.c:1449-1870) for n transform blocks of one shape.
//
// The algorithm is a strictly sequential walk over a block's coefficients in reverse scan order: every decision reads
// the levels already decided to the right / below (context selection, rdo.c:1400-1438), the running regular-bin budget
// and Rice parameter (:1692-1699), and a running double-precision cost that later comparisons test against (:1689,
// :1724-1752).  Floating-point addition does not re-associate, so a bit-exact result needs that order.  Blocks are
// independent of each other (the context models are read-only inside a call), hence:
//     one lane = one transform block, one wave = 64 blocks walking the same scan position in lockstep
//   * levels live in LDS as [position][lane] (neighbour reads are lane-private, conflict-free);
//   * the three per-position cost arrays the last-position search re-reads (:1786-1823) live in a caller-provided
//     workspace as [position][block] -- lockstep lanes make every access a coalesced row;
//   * bin costs: CTX_ENTROPY_BITS of the 244 context models the routine touches are looked up once per workgroup
//     (generated table, vvc_rdoq_tables.h) into LDS; the last-position prefix costs (:667-700) likewise.
// All arithmetic on costs is IEEE double in the reference's order; this file is compiled with -ffp-contract=off, and
// the one transcendental (pow(2, -2 * transform_shift), :1527) is evaluated on the host like the reference does.
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
__device__ inline int ic_rate(const uint32_t (*B)[2], int t, uint32_t abs_level, int ctx, int go_rice, uint32_t reg_bins)
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

// One wave per workgroup; lane = block.  Dynamic LDS: levels[wh][64] int16.
__global__ void __launch_bounds__(64)
rdoq_kernel(const rdoq_params P, const int16_t *__restrict__ coef, int16_t *__restrict__ q_coef, double *__restrict__ ws,
            uint32_t *__restrict__ abs_sum_out, uint8_t *__restrict__ has_coeffs)
{
  extern __shared__ __attribute__((aligned(16))) int16_t sLev[];        // [wh][64]
  __shared__ uint32_t sB[N_CTX][2];
  __shared__ int sLastX[32], sLastY[32];
  __shared__ uint8_t sScanCg[64];
  const int lane = threadIdx.x;
  const int width = P.width, height = P.height, wh = width * height, l2w = P.l2w;
  const int n = P.n;
  const int tu = blockIdx.x * 64 + lane;
  const bool live = tu < n;
  const int t = P.color ? 1 : 0;

  // ---- per-workgroup tables ----
  {
    const uint8_t *st = reinterpret_cast<const uint8_t *>(&P.ctx);
    for (int i = lane; i < N_CTX; i += 64) { const int s = st[i]; sB[i][0] = kEntropyBits[2 * s]; sB[i][1] = kEntropyBits[2 * s + 1]; }
  }
  const int cgw = width >> 2, cgh = height >> 2;
  if (lane == 0) {                                                     // H.266 6.5.2 on the coefficient-group grid
    int i = 0, x = 0, y = 0;
    while (i < cgw * cgh) {
      while (y >= 0) { if (x < cgw && y < cgh) sScanCg[i++] = (uint8_t)(y * cgw + x); y--; x++; }
      y = x; x = 0;
    }
  }
  __syncthreads();
  if (lane == 0) {                                                     // calc_last_bits, rdo.c:667-700
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
  __syncthreads();
  if (!live) return;

  // in-group up-right diagonal order of a 4x4 group as (y * 4 + x) nibbles, scan position 0 first
  constexpr unsigned long long kDiag4 = 0xFBE7AD369C258140ull;
  auto blk_of = [&](int scanpos) {
    const int g = sScanCg[scanpos >> 4], k = (int)((kDiag4 >> (4 * (scanpos & 15))) & 15);
    return (((g / cgw) * 4 + (k >> 2)) << l2w) + (g % cgw) * 4 + (k & 3);
  };
  const int16_t *C = coef + (size_t)tu * wh;
  int16_t *L = sLev + lane;                                            // level of position p: L[p * 64]
  for (int p = 0; p < wh; ++p) L[p * 64] = 0;
  double *cost_coeff = ws + tu, *cost_sig = ws + (size_t)wh * n + tu, *cost_coeff0 = ws + (size_t)2 * wh * n + tu;
  double *cost_cg = ws + (size_t)3 * wh * n + tu;                       // [64][n]
#define CC(a, i) a[(size_t)(i) * n]

  const double lambda = P.lambda, error_scale = P.error_scale;
  const int q_bits = P.q_bits, q = P.q;
  const int cap = 0x7fffffff - (1 << (q_bits - 1));
  const uint32_t cg_width = (uint32_t)min(width, 32) >> 2, cg_height = (uint32_t)min(height, 32) >> 2;
  const uint32_t num_blk_side = max(width >> 2, 1);
  const int cg_num = P.lfnst_idx > 0 ? 1 : wh >> 4;
  const int max_group = P.lfnst_idx > 0 ? (((height == 4 && width == 4) || (height == 8 && width == 8)) ? 7 : 15) : 15;
  const int mts = P.mts_idx;
  unsigned long long sig_cg = 0;                                       // sig_coeffgroup_flag as a bit set over group raster positions
  double block_uncoded_cost = 0, base_cost = 0;
  int cg_last_scanpos = -1, last_scanpos = -1;
  uint32_t reg_bins = (uint32_t)(wh * 28) >> 4;
  int go_rice_param = 0;
  int temp_diag = -1, temp_sum = -1;

  auto level_double_of = [&](int blkpos) {
    const long long prod = (long long)abs((int)C[blkpos]) * q;
    return (int)(prod < cap ? prod : cap);
  };

  // ---- find the last significant position (rdo.c:1561-1592) ----
  int cg_scanpos;
  for (cg_scanpos = cg_num - 1; cg_scanpos >= 0; cg_scanpos--) {
    const uint32_t cg_blkpos = sScanCg[cg_scanpos];
    const uint32_t cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
    if (mts != 0 && (cg_pos_y >= 4 || cg_pos_x >= 4)) continue;
    for (int sp = max_group; sp >= 0; sp--) {
      const int scanpos = cg_scanpos * 16 + sp;
      const int blkpos = blk_of(scanpos);
      const int level_double = level_double_of(blkpos);
      const uint32_t max_abs_level = (uint32_t)(level_double + (1 << (q_bits - 1))) >> q_bits;
      const double err = (double)level_double;
      const double c0 = err * err * error_scale;
      CC(cost_coeff0, scanpos) = c0;
      L[blkpos * 64] = (int16_t)max_abs_level;
      if (max_abs_level > 0) { last_scanpos = scanpos; cg_last_scanpos = cg_scanpos; break; }
      block_uncoded_cost += c0;
      base_cost += c0;
    }
    if (last_scanpos != -1) break;
  }
  int16_t *Q = q_coef + (size_t)tu * wh;
  if (last_scanpos == -1) {
    for (int p = 0; p < wh; ++p) Q[p] = 0;
    if (abs_sum_out) abs_sum_out[tu] = 0;
    if (has_coeffs) has_coeffs[tu] = 0;
    return;
  }
  for (; cg_scanpos >= 0; cg_scanpos--) CC(cost_cg, cg_scanpos) = 0;

  const uint32_t (*B)[2] = sB;
  // ---- level decisions, coefficient group by coefficient group (rdo.c:1604-1773) ----
  for (int cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const uint32_t cg_blkpos = sScanCg[cgs];
    const uint32_t cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
    double rd_coded = 0, rd_uncoded = 0, rd_sig = 0, rd_sig0 = 0;
    int nnz_before_pos0 = 0;
    if (mts != 0 && (cg_pos_y >= 4 || cg_pos_x >= 4)) continue;
    for (int sp = max_group; sp >= 0; sp--) {
      const int scanpos = cgs * 16 + sp;
      if (scanpos > last_scanpos) continue;
      const int blkpos = blk_of(scanpos);
      const int level_double = level_double_of(blkpos);
      const uint32_t max_abs_level = (uint32_t)(level_double + (1 << (q_bits - 1))) >> q_bits;
      L[blkpos * 64] = (int16_t)max_abs_level;
      const double err0 = (double)level_double;
      const double c0 = err0 * err0 * error_scale;
      CC(cost_coeff0, scanpos) = c0;
      block_uncoded_cost += c0;

      const uint32_t pos_y = (uint32_t)blkpos >> l2w, pos_x = (uint32_t)blkpos - (pos_y << l2w);
      const bool is_last = scanpos == last_scanpos;
      // neighbourhood of already decided levels: right, right+1, below-right, below, below+1 (rdo.c:1400-1438, 846-871)
      const int16_t *D = L + blkpos * 64;
      int nb[5] = {0, 0, 0, 0, 0};
      bool has[5] = {false, false, false, false, false};
      if (pos_x < (uint32_t)width - 1) {
        has[0] = true; nb[0] = D[64];
        if (pos_x < (uint32_t)width - 2) { has[1] = true; nb[1] = D[128]; }
        if (pos_y < (uint32_t)height - 1) { has[2] = true; nb[2] = D[(width + 1) * 64]; }
      }
      if (pos_y < (uint32_t)height - 1) {
        has[3] = true; nb[3] = D[width * 64];
        if (pos_y < (uint32_t)height - 2) { has[4] = true; nb[4] = D[2 * width * 64]; }
      }
      int ctx_sig = 0;
      if (!is_last) {
        // the zero-out tests of context_get_sig_ctx_idx_abs (note: the "below" terms test pos_x, as the reference does)
        const bool z0 = mts && pos_x + 1 >= 16, z1 = mts && pos_x + 2 >= 16, z2 = mts && (pos_y + 1 >= 16 || pos_x + 1 >= 16);
        const bool z3 = mts && pos_x + 1 >= 16, z4 = mts && pos_x + 2 >= 16;
        const bool zz[5] = {z0, z1, z2, z3, z4};
        int num_pos = 0, sum_abs = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (has[k]) { const int a = zz[k] ? 0 : abs(nb[k]); sum_abs += min(4 + (a & 1), a); num_pos += a ? 1 : 0; }
        const int diag = (int)(pos_x + pos_y);
        ctx_sig = min((sum_abs + 1) >> 1, 3) + (diag < 2 ? 4 : 0);
        if (P.color == 0) ctx_sig += diag < 5 ? 4 : 0;
        temp_diag = diag; temp_sum = sum_abs - num_pos;
      }
      int ctx_set = 0;
      if (temp_diag != -1)
        ctx_set = (min(temp_sum, 4) + 1) + (!temp_diag ? ((P.color == 0) ? 15 : 5) : (P.color == 0) ? (temp_diag < 3 ? 10 : (temp_diag < 10 ? 5 : 0)) : 0);
      // templateAbsSum over the decided levels (base 0) -- its zero-out tests differ from the context function's
      if (reg_bins < 4) {
        const bool zz[5] = {mts && pos_x + 1 >= 16, mts && pos_x + 2 >= 16, mts && (pos_y + 1 >= 16 || pos_x + 1 >= 16),
                            mts && pos_y + 1 >= 16, mts && pos_y + 2 >= 16};
        int16_t sum = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) if (has[k]) sum = (int16_t)(sum + (zz[k] ? 0 : abs(nb[k])));
        go_rice_param = go_rice_par(clampi((int)sum, 0, 31));
      }

      // uvg_get_coded_level (rdo.c:597-640)
      double coded_cost, coded_sig = 0, cur_cost_sig = 0;
      uint32_t best = 0;
      bool done = false;
      if (!is_last && max_abs_level < 3) {
        coded_sig = lambda * (double)B[O_SIG + 12 * t + ctx_sig][0];
        coded_cost = c0 + coded_sig;
        if (max_abs_level == 0) done = true;
      } else {
        coded_cost = 1.7e+308;
      }
      if (!done) {
        if (!is_last) cur_cost_sig = lambda * (double)B[O_SIG + 12 * t + ctx_sig][1];
        const int min_abs = max_abs_level > 1 ? (int)max_abs_level - 1 : 1;
        for (int a = (int)max_abs_level; a >= min_abs; a--) {
          const double err = (double)(level_double - (a * (1 << q_bits)));
          double cur = err * err * error_scale + lambda * (double)ic_rate(B, t, (uint32_t)a, ctx_set, go_rice_param, reg_bins);
          cur += cur_cost_sig;
          if (cur < coded_cost) { best = (uint32_t)a; coded_cost = cur; coded_sig = cur_cost_sig; }
        }
      }
      const int level = (int)best;
      CC(cost_coeff, scanpos) = coded_cost;
      CC(cost_sig, scanpos) = coded_sig;
      L[blkpos * 64] = (int16_t)level;
      base_cost += coded_cost;

      // context set update (rdo.c:1691-1699)
      if ((scanpos % 16 == 0) && scanpos > 0) go_rice_param = 0;
      else if (reg_bins >= 4) {
        reg_bins -= (uint32_t)((level < 2 ? level : 3) + (is_last ? 0 : 1));
        // sic: templateAbsSum over the INPUT coefficients with base level 4 (rdo.c:1697)
        const bool zz[5] = {mts && pos_x + 1 >= 16, mts && pos_x + 2 >= 16, mts && (pos_y + 1 >= 16 || pos_x + 1 >= 16),
                            mts && pos_y + 1 >= 16, mts && pos_y + 2 >= 16};
        const int off[5] = {1, 2, width + 1, width, 2 * width};
        int16_t sum = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) if (has[k]) sum = (int16_t)(sum + (zz[k] ? 0 : abs((int)C[blkpos + off[k]])));
        go_rice_param = go_rice_par(clampi((int)sum - 20, 0, 31));
      }

      rd_sig += coded_sig;
      if (sp == 0) rd_sig0 = coded_sig;
      if (level) {
        sig_cg |= 1ull << cg_blkpos;
        rd_coded += coded_cost - coded_sig;
        rd_uncoded += c0;
        if (sp != 0) nnz_before_pos0++;
      }
    }
    // coefficient-group decision (rdo.c:1719-1772)
    if (cgs) {
      const uint32_t pos = cg_pos_y * cg_width + cg_pos_x;
      uint32_t right = 0, lower = 0;
      if (cg_pos_x + 1 < cg_width) right = (uint32_t)(sig_cg >> (pos + 1)) & 1;
      if (cg_pos_y + 1 < cg_height) lower = (uint32_t)(sig_cg >> (pos + cg_width)) & 1;
      const int cs = O_SIGGRP + 2 * t + ((right || lower) ? 1 : 0);
      if (!((sig_cg >> cg_blkpos) & 1)) {
        const double v = lambda * (double)B[cs][0];
        CC(cost_cg, cgs) = v;
        base_cost += v - rd_sig;
      } else if (cgs < cg_last_scanpos) {
        if (nnz_before_pos0 == 0) { base_cost -= rd_sig0; rd_sig -= rd_sig0; }
        double cost_zero_cg = base_cost;
        double v = lambda * (double)B[cs][1];
        base_cost += v;
        cost_zero_cg += lambda * (double)B[cs][0];
        cost_zero_cg += rd_uncoded;
        cost_zero_cg -= rd_coded;
        cost_zero_cg -= rd_sig;
        if (cost_zero_cg < base_cost) {
          sig_cg &= ~(1ull << cg_blkpos);
          base_cost = cost_zero_cg;
          v = lambda * (double)B[cs][0];
          for (int sp = max_group; sp >= 0; sp--) {
            const int scanpos = cgs * 16 + sp;
            const int blkpos = blk_of(scanpos);
            if (L[blkpos * 64]) { L[blkpos * 64] = 0; CC(cost_coeff, scanpos) = CC(cost_coeff0, scanpos); CC(cost_sig, scanpos) = 0; }
          }
        }
        CC(cost_cg, cgs) = v;
      }
    } else {
      sig_cg |= 1ull << cg_blkpos;
    }
  }

  // ---- last position (rdo.c:1775-1829) ----
  double best_cost;
  int best_last_idx_p1 = 0;
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
    const uint32_t cg_blkpos = sScanCg[cgs];
    base_cost -= CC(cost_cg, cgs);
    if ((sig_cg >> cg_blkpos) & 1) {
      for (int sp = max_group; sp >= 0; sp--) {
        const int scanpos = cgs * 16 + sp;
        if (scanpos > last_scanpos) continue;
        const int blkpos = blk_of(scanpos);
        const int lv = L[blkpos * 64];
        if (lv) {
          const uint32_t pos_y = (uint32_t)blkpos >> l2w, pos_x = (uint32_t)blkpos - (pos_y << l2w);
          const uint32_t cx = (uint32_t)group_idx((int)pos_x), cy = (uint32_t)group_idx((int)pos_y);
          double ui = (double)(sLastX[cx] + sLastY[cy]);
          if (cx > 3) ui += (double)(32768u * ((cx - 2) >> 1));
          if (cy > 3) ui += (double)(32768u * ((cy - 2) >> 1));
          const double cost_last = lambda * ui;
          const double total = base_cost + cost_last - CC(cost_sig, scanpos);
          if (total < best_cost) { best_last_idx_p1 = scanpos + 1; best_cost = total; }
          if (lv > 1) { found_last = true; break; }
          base_cost -= CC(cost_coeff, scanpos);
          base_cost += CC(cost_coeff0, scanpos);
        } else {
          base_cost -= CC(cost_sig, scanpos);
        }
      }
    }
  }

  // ---- signs, clean-up, output (rdo.c:1831-1858) ----
  uint32_t abs_sum = 0;
  const bool reduce = mts && !(width < 32 && height < 32);
  for (int scanpos = 0; scanpos < best_last_idx_p1; scanpos++) {
    const int b = blk_of(scanpos);
    int level = L[b * 64];
    if (reduce) { const int bx = b & (width - 1), by = b >> l2w; if (bx >= 16 || by >= 16) level = 0; }
    abs_sum += (uint32_t)level;
    L[b * 64] = (int16_t)((level != 0 && C[b] < 0) ? -level : level);
  }
  for (int scanpos = best_last_idx_p1; scanpos <= last_scanpos; scanpos++) L[blk_of(scanpos) * 64] = 0;
  bool any = false;
  for (int p = 0; p < wh; ++p) { const int16_t v = L[p * 64]; Q[p] = v; any |= v != 0; }
  if (abs_sum_out) abs_sum_out[tu] = abs_sum;
  if (has_coeffs) has_coeffs[tu] = any ? 1 : 0;
#undef CC
}

}  // namespace

extern "C" size_t uvghip_rdoq_workspace_bytes(int width, int height, int n)
{
  if (width <= 0 || height <= 0 || n <= 0) return 0;
  return ((size_t)3 * width * height + 64) * (size_t)n * sizeof(double);
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
  const size_t lds = (size_t)width * height * 64 * sizeof(int16_t);
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(rdoq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  });
  UVGHIP_TRY(attr_err);
  rdoq_kernel<<<(n + 63) / 64, 64, lds, uvghip_stream(stream)>>>(P, coef, q_coef, static_cast<double *>(workspace), abs_sum_out, has_coeffs);
  UVGHIP_CHECK_LAUNCH();
}
