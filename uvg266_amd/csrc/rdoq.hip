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
// Mapping: four lanes per block (an anti-diagonal of a 4x4 coefficient group has at most four positions), 16 blocks per
// wave; see the kernel's comment for the per-group stages.  All cost arithmetic is IEEE double in the reference's operation
// order; this file is compiled with -ffp-contract=off, and the one transcendental (pow(2, -2 * transform_shift), :1527) is
// evaluated on the host.  Where the time goes (tools/dev/rdoq_pmc.sh, tools/dev/rdoq_real.py on the 1080p bench data): the
// wave issues an instruction ~50 % of its lifetime at <= 1 wave per SIMD (LDS-limited), so the instruction count per
// coefficient group is what is optimised: context-free halves of the candidates' costs staged in parallel, lane-role replay
// chains, last-position search on staged values.
#include "uvghip_common.h"
#include "vvc_rdoq_tables.h"
#include <cmath>
#include <mutex>
#include <cstdio>

namespace {

enum : int { O_SIGGRP = 0, O_SIG = 4, O_PAR = 28, O_GT1 = 70, O_GT2 = 112, O_LASTX = 154, O_LASTY = 194, O_CBF_Y = 234, O_CBF_CB = 238,
             O_CBF_CR = 240, O_ROOT = 243, N_CTX = 244 };
static_assert(sizeof(uvghip_rdoq_ctx_t) == N_CTX, "uvghip_rdoq_ctx_t layout");
// the kernel's LDS table of the models' bit costs leaves out the 80 last-position models (they only feed the two small
// per-coordinate tables built at kernel start): the models behind them move down
enum : int { S_DROP = O_CBF_Y - O_LASTX, S_CBF_Y = O_CBF_Y - S_DROP, S_CBF_CB = O_CBF_CB - S_DROP, S_CBF_CR = O_CBF_CR - S_DROP,
             S_ROOT = O_ROOT - S_DROP, N_CTXS = N_CTX - S_DROP };

struct rdoq_params {
  int width, height, l2w, l2h, n;
  int color, block_type, cbf_u, lfnst_idx, mts_idx;
  int q_bits, q, signhide;
  long long rd_factor;          // uvg_rdoq_sign_hiding's rd_factor (host, rdo.c:724-726)
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

// uvg_get_ic_rate with use_limited_prefix_length = false: what the sign-hiding bookkeeping calls (rdo.c:1674-1681)
__device__ __noinline__ int ic_rate_sh(const uint32_t (*B)[2], int t, uint32_t abs_level, int ctx, int go_rice, uint32_t reg_bins)
{
  int rate = 1 << 15;
  const int thr = 5;
  if (reg_bins < 4) {
    const uint32_t zero = 1u << go_rice;
    uint32_t symbol = (abs_level == 0 ? zero : abs_level <= zero ? abs_level - 1 : abs_level);
    if (symbol < ((uint32_t)thr << go_rice)) {
      rate += (int)(((symbol >> go_rice) + 1 + go_rice) << 15);
    } else {
      uint32_t length = (uint32_t)go_rice;
      symbol = symbol - ((uint32_t)thr << go_rice);
      while ((int)symbol >= (1 << length)) symbol -= (1u << (length++));
      rate += (int)((thr + length + 1 - go_rice + length) << 15);
    }
    return rate;
  }
  const int par = O_PAR + 21 * t + ctx, gt1 = O_GT1 + 21 * t + ctx, gt2 = O_GT2 + 21 * t + ctx;
  if (abs_level >= 4) {
    int symbol = (int)abs_level - 4;
    if (symbol < (thr << go_rice)) {
      rate += ((symbol >> go_rice) + 1 + go_rice) << 15;
    } else {
      int length = go_rice;
      symbol = symbol - (thr << go_rice);
      while (symbol >= (1 << length)) symbol -= (1 << (length++));
      rate += (thr + length + 1 - go_rice + length) << 15;
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

// struct sh_rates_t of one position (rdo.c:214-223, filled at :1660-1687): the caller's workspace holds them as
// [block][inc | dec | sig_coeff_inc | quant_delta][position]
__device__ __forceinline__ void sh_record(int32_t *gSh, int wh, int blkpos, const uint32_t (*B)[2], int t, bool is_last, int level,
                                          int level_double, int q_bits, int ctx_sig, int ctx_set, int go_rice, uint32_t reg_bins)
{
  int inc, dec = 0;
  if (level > 0) {
    const int now = ic_rate_sh(B, t, (uint32_t)level, ctx_set, go_rice, reg_bins);
    inc = ic_rate_sh(B, t, (uint32_t)level + 1, ctx_set, go_rice, reg_bins) - now;
    dec = ic_rate_sh(B, t, (uint32_t)level - 1, ctx_set, go_rice, reg_bins) - now;
  } else if (reg_bins < 4) {
    inc = ic_rate_sh(B, t, 1, ctx_set, go_rice, reg_bins) - ic_rate_sh(B, t, 0, ctx_set, go_rice, reg_bins);
  } else {
    inc = (int)B[O_GT1 + 21 * t + ctx_set][0];
  }
  gSh[blkpos] = inc;
  gSh[wh + blkpos] = dec;
  gSh[2 * wh + blkpos] = is_last ? 0 : (reg_bins < 4 ? 0 : (int)B[O_SIG + 12 * t + ctx_sig][1] - (int)B[O_SIG + 12 * t + ctx_sig][0]);
  gSh[3 * wh + blkpos] = (level_double - level * (1 << q_bits)) >> (q_bits - 8);
}

struct rdoq_decision { int level; int sig_code; double coded_cost, coded_sig; int ctx_sig, ctx_set, go_rice; };

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
  d.ctx_sig = ctx_sig; d.ctx_set = ctx_set; d.go_rice = go_rice;
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

// LDS bytes of one block: level int16[wh] (holds the input coefficient until the position's group is staged) + the Rice
// parameters, four per byte; block strides are odd in words so that the same position of the 16 blocks of a wave falls into
// 16 different banks
// bytes == true (32x32 blocks): the level array holds bytes -- levels clamped to 254 + parity, which keeps every context of the
// walk exact -- and no input coefficients: the exact levels go to the output as they are decided, the input is read from
// global memory a group ahead.  1284 instead of 2308 bytes per block: five workgroups per CU instead of three.
__host__ __device__ constexpr size_t rdoq_lds_per_block(int wh, bool bytes = false)
{
  const size_t raw = (size_t)wh * (bytes ? 1 : 2) + wh / 4;
  return ((raw >> 2) & 1) ? raw : raw + 4;
}

// Everything about a block that does not depend on decisions, in one fully parallel pass in front of the walk (one thread per
// four positions of a row, whole blocks per workgroup):
//   * the Rice parameter after each position: templateAbsSum(coef, 4, ...) over the INPUT block (rdo.c:846-871, 1697), two
//     bits per position, four positions per byte -> rice_out[block][wh / 4];
//   * the last significant position (rdo.c:1561-1592): the highest scan position whose rounded level is non-zero, -1 if
//     none -> last_out[block].
__global__ void __launch_bounds__(256)
rdoq_pre_kernel(const rdoq_params P, const int16_t *__restrict__ coef, uint8_t *__restrict__ rice_out, int *__restrict__ last_out)
{
  __shared__ uint8_t sInv[64];                                         // coefficient-group raster index -> index in scan order
  __shared__ int sLast[64];
  __shared__ __attribute__((aligned(8))) int16_t sC[1024];             // the workgroup's 1024 coefficients: every one is read six times
  const int width = P.width, height = P.height, l2w = P.l2w, wh = width * height, n = P.n;
  const int l2cgw = l2w - 2, cgw = 1 << l2cgw, cgh = height >> 2;
  const int mts = P.mts_idx;
  const int bpw = 1024 / wh, blk0 = blockIdx.x * bpw;
  if ((int)threadIdx.x < cgw * cgh) {
    // H.266 6.5.2 on the coefficient-group grid, every group for itself: the groups of the earlier anti-diagonals, then
    // those before it on its own (an anti-diagonal runs from its bottom-left end upwards: x ascending)
    const int cx = threadIdx.x & (cgw - 1), cy = threadIdx.x >> l2cgw, d = cx + cy;
    int idx = cx - max(0, d - (cgh - 1));
    for (int e = 0; e < d; ++e) idx += min(e, cgw - 1) - max(0, e - (cgh - 1)) + 1;
    sInv[threadIdx.x] = (uint8_t)idx;
  }
  if (threadIdx.x < 64) sLast[threadIdx.x] = -1;
  const int pos4 = threadIdx.x * 4, b = pos4 / wh, pos0 = pos4 - b * wh, tu = blk0 + b;
  if (tu < n) {                                                        // four coefficients per thread, one load where the pointer allows it
    const int16_t *g = coef + (size_t)blk0 * wh + pos4;
    if ((reinterpret_cast<uintptr_t>(g) & 7) == 0) *reinterpret_cast<uint2 *>(sC + pos4) = *reinterpret_cast<const uint2 *>(g);
    else { sC[pos4] = g[0]; sC[pos4 + 1] = g[1]; sC[pos4 + 2] = g[2]; sC[pos4 + 3] = g[3]; }
  }
  __syncthreads();
  if (tu < n) {
    const int16_t *c = sC + b * wh;
    const int q_bits = P.q_bits, q = P.q;
    const int cap = 0x7fffffff - (1 << (q_bits - 1));
    const int cg_num = P.lfnst_idx > 0 ? 1 : wh >> 4;
    const int max_group = P.lfnst_idx > 0 ? (((height == 4 && width == 4) || (height == 8 && width == 8)) ? 7 : 15) : 15;
    constexpr unsigned long long kInvDiag4 = 0xFDA6EB73C8419520ull;    // (y * 4 + x) inside a 4x4 group -> index in its up-right diagonal scan
    unsigned packed = 0;
    int last = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pos = pos0 + k;
      const uint32_t pos_y = (uint32_t)pos >> l2w, pos_x = (uint32_t)pos - (pos_y << l2w);
      const int16_t *c0p = c + pos;
      const bool hx1 = pos_x < (uint32_t)width - 1, hx2 = pos_x < (uint32_t)width - 2;
      const bool hy1 = pos_y < (uint32_t)height - 1, hy2 = pos_y < (uint32_t)height - 2;
      const bool zx1 = mts && pos_x + 1 >= 16, zx2 = mts && pos_x + 2 >= 16, zy1 = mts && pos_y + 1 >= 16, zy2 = mts && pos_y + 2 >= 16;
      const int r0 = c0p[hx1 ? 1 : 0], r1 = c0p[hx2 ? 2 : 0], r2 = c0p[(hx1 && hy1) ? width + 1 : 0], r3 = c0p[hy1 ? width : 0],
                r4 = c0p[hy2 ? 2 * width : 0];
      // coeff_t accumulator: wraps like the reference's
      const int16_t sum = (int16_t)(((hx1 && !zx1) ? abs(r0) : 0) + ((hx2 && !zx2) ? abs(r1) : 0) + ((hx1 && hy1 && !(zy1 || zx1)) ? abs(r2) : 0) +
                                    ((hy1 && !zy1) ? abs(r3) : 0) + ((hy2 && !zy2) ? abs(r4) : 0));
      packed |= (unsigned)go_rice_par(clampi((int)sum - 20, 0, 31)) << (2 * k);
      // is this position a candidate for the last significant one?
      const int g = (int)((pos_y >> 2) << l2cgw) + (int)(pos_x >> 2);
      const int cgs = sInv[g];
      const int s4 = (int)((kInvDiag4 >> (4 * ((pos_y & 3) * 4 + (pos_x & 3)))) & 15);
      const bool skipped = mts != 0 && ((g >> l2cgw) >= 4 || (g & (cgw - 1)) >= 4);
      const long long prod = (long long)abs((int)c0p[0]) * q;
      const int ld = (int)(prod < cap ? prod : cap);
      const bool sig = ((uint32_t)(ld + (1 << (q_bits - 1))) >> q_bits) > 0;
      if (sig && cgs < cg_num && !skipped && s4 <= max_group) last = max(last, cgs * 16 + s4);
    }
    rice_out[(size_t)tu * (wh >> 2) + (pos0 >> 2)] = (uint8_t)packed;
    // the lanes of one block inside a wave (wh / 4 of them, an aligned power of two) reduce first
    for (int o = min(32, wh >> 3); o; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if ((threadIdx.x & (min(64, wh >> 2) - 1)) == 0 && last >= 0) atomicMax(&sLast[b], last);
  }
  __syncthreads();
  if ((int)threadIdx.x < bpw && blk0 + (int)threadIdx.x < n) last_out[blk0 + threadIdx.x] = sLast[threadIdx.x];
}

// value of lane K of the caller's quad (DPP quad_perm broadcast; all four lanes of a block's quad are active together)
template <int K>
__device__ __forceinline__ double quad_bcast(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, K * 0x55, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, K * 0x55, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// LDS traffic of a workgroup (= one wave) is ordered by program order; what the phases need between them is only that the
// compiler keeps that order.  (__syncthreads would also wait for the global cost_coeff stores of the phase.)
#define WAVE_SYNC()                                             \
  do {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
    __builtin_amdgcn_wave_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
  } while (0)

// Four lanes per block (an anti-diagonal of a 4x4 group has at most four positions), TUS blocks per wave.
//
// Per coefficient group:
//   stage    every position (4 per lane): |coef| * q, its rounded level, cost0, and the context-free half of the two candidate
//            levels' costs -- distortion and the sign / Golomb-Rice part of the rate (the Rice parameter is context-free while
//            regular bins remain, see the header);
//   7 phases the positions of one anti-diagonal pick their context from the neighbours' decided levels, add the three
//            context-coded flag costs and the significance cost to the staged halves and choose -- ~1/4 of the instructions
//            of a full uvg_get_coded_level per position, which is what the wave's time goes into (SQ counters: the kernel
//            issues instructions > 50 % of its lifetime at one wave per SIMD);
//   replay   the 16 costs in scan order (bit-exact double sums), group decision.
// LDS per block (rdoq_lds_per_block): level int16[wh] (holds the input coefficient until the position's group is staged) +
// the Rice parameter after each position, two bits each.  What the walk only writes and the last-position search reads back
// once -- cost_coeff[], the code of each position's cost_sig[] entry, cost_coeffgroup_sig[] -- lives in the caller's
// workspace, written and re-read by the same lane: the number of resident workgroups is bound by their LDS (the waves wait
// on each other's latencies, not on the VALU), 32x32 blocks went from two to three workgroups per CU with it.
// SIGNHIDE: sign-data hiding compiled in (its bookkeeping costs registers the plain kernel needs for occupancy).
// SHAPE: log2 of the side of a square block (2..5) -- dimensions and plane type (CHROMA) become compile-time constants and
// every shape is its own kernel symbol in a profile -- or 0 for the generic kernel (rectangles; everything from the parameters).
template <int TUS, int SHAPE, int CHROMA, int SIGNHIDE>
// (4x4 blocks: three waves per SIMD asked for -- 168 registers, no spills -- because there the registers, not the LDS, bound
// the number of resident workgroups; the larger shapes are LDS-bound and spill when squeezed)
__global__ void __launch_bounds__(64, ((SHAPE == 2 || SHAPE == 3) && !SIGNHIDE) ? 3 : 1)
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
  __shared__ uint32_t sB[N_CTXS][2];
  // (16x16: sized by the shape -- the LDS allocator's granule is 1280 bytes, and 176 bytes less put an eighth workgroup on a CU)
  constexpr int NSIDE = SHAPE == 4 ? 16 : 32, NCGS = SHAPE == 4 ? 16 : 64;
  __shared__ int sLastXp[NSIDE], sLastYp[NSIDE];                       // get_rate_last (:645-658) per coordinate: prefix bits + suffix bits
  __shared__ uint8_t sScanCg[NCGS];
  // per group and position s4: D[3*s4 + {0: distortion of candidate 1 -> coded_cost, 1: candidate 2 -> coded_sig, 2: cost0}],
  // I[3*s4 + {0: rate half of candidate 1 -> level, 1: candidate 2, 2: level_double}]
  __shared__ double sStageD[TUS][49];
  __shared__ int sStageI[TUS][49];
  const int tid = threadIdx.x, grp = tid >> 2, j = tid & 3;
  const int l2w = SHAPE ? SHAPE : P.l2w, l2h_ = SHAPE ? SHAPE : P.l2h;
  const int width = 1 << l2w, height = 1 << l2h_, wh = width * height;
  const int n = P.n;
  constexpr bool BYTES = SHAPE == 5 && !SIGNHIDE;                      // byte level array (see rdoq_lds_per_block)
  constexpr bool SCAN_WS = !(SHAPE == 2 || SHAPE == 3);                // cost / significance workspace indexed by scan position
  const size_t per_tu = rdoq_lds_per_block(wh, BYTES);
  const int t = SHAPE ? CHROMA : (P.color ? 1 : 0);
  const int mts = P.mts_idx;

  // ---- per-workgroup tables ----
  {
    const uint8_t *st = reinterpret_cast<const uint8_t *>(&P.ctx);
    for (int i = tid; i < N_CTX; i += 64) {
      if (i >= O_LASTX && i < O_CBF_Y) continue;
      const int s = st[i], k = i < O_LASTX ? i : i - S_DROP;
      sB[k][0] = kEntropyBits[2 * s]; sB[k][1] = kEntropyBits[2 * s + 1];
    }
  }
  const int l2cgw = l2w - 2, cgw = 1 << l2cgw, cgh = height >> 2;
  if (tid == 0) {                                                      // H.266 6.5.2 on the coefficient-group grid
    int i = 0, x = 0, y = 0;
    while (i < cgw * cgh) {
      while (y >= 0) { if (x < cgw && y < cgh) sScanCg[i++] = (uint8_t)(y * cgw + x); y--; x++; }
      y = x; x = 0;
    }
  }
  __syncthreads();
  if (tid == 0) {                                                      // calc_last_bits, rdo.c:667-700
    auto prefix_ctx = [](int l2) { return l2 <= 2 ? 0 : l2 == 3 ? 3 : l2 == 4 ? 6 : 10; };   // {0,0,0,3,6,10,15,21}[log2 size]
    const int l2h = l2h_;
    const int ox = t ? 0 : prefix_ctx(l2w), oy = t ? 0 : prefix_ctx(l2h);
    const int sx = t ? clampi(width >> 3, 0, 2) : ((l2w + 1) >> 2), sy = t ? clampi(height >> 3, 0, 2) : ((l2h + 1) >> 2);
    const uint8_t *st = reinterpret_cast<const uint8_t *>(&P.ctx);
    int *sLastX = &sStageI[0][0], *sLastY = &sStageI[1][0];          // scratch: the staging arrays are idle until the walk
    int bits = 0, c;
    for (c = 0; c < group_idx(width - 1); ++c) {
      const int sm = st[O_LASTX + 20 * t + ox + (c >> sx)];
      sLastX[c] = bits + (int)kEntropyBits[2 * sm]; bits += (int)kEntropyBits[2 * sm + 1];
    }
    sLastX[c] = bits;
    bits = 0;
    for (c = 0; c < group_idx(height - 1); ++c) {
      const int sm = st[O_LASTY + 20 * t + oy + (c >> sy)];
      sLastY[c] = bits + (int)kEntropyBits[2 * sm]; bits += (int)kEntropyBits[2 * sm + 1];
    }
    sLastY[c] = bits;
    for (int p = 0; p < width; ++p) { const int cx = group_idx(p); sLastXp[p] = sLastX[cx] + (cx > 3 ? (int)(32768u * (uint32_t)((cx - 2) >> 1)) : 0); }
    for (int p = 0; p < height; ++p) { const int cy = group_idx(p); sLastYp[p] = sLastY[cy] + (cy > 3 ? (int)(32768u * (uint32_t)((cy - 2) >> 1)) : 0); }
  }
  __syncthreads();
  // ---- a workgroup takes batches of TUS blocks until none are left: the tables above are built once ----
  for (int tu0 = blockIdx.x * TUS; tu0 < n; tu0 += gridDim.x * TUS) {
  const int here = min(TUS, n - tu0);
  const bool live = grp < here;
  const int gq = live ? grp : 0;                                       // idle lanes alias block 0 for address arithmetic only
  const int tu = tu0 + gq;
  // workspace: cost_coeff[n][wh] | cost_coeffgroup_sig[n][wh / 16] | (sign hiding: sh_rates [n][4][wh]) | cost_sig code [n][wh]
  double *gCost = ws + (size_t)tu * wh;                                // cost_coeff[] of this block (re-read by the last-position search)
  double *gCgCost = ws + (size_t)n * wh + (size_t)tu * (wh >> 4);      // cost_coeffgroup_sig[]: written and re-read by lane 0 of the block
  int32_t *wsSh = reinterpret_cast<int32_t *>(ws + (size_t)n * wh + (size_t)n * (wh >> 4));
  int32_t *gSh = wsSh + (size_t)tu * 4 * wh;                           // sign hiding: the block's sh_rates
  // code of the significance-cost table entry of every walked position (cost_sig[] = sig_cost_of(code)); a position is
  // written and re-read by the same lane (the j + 4r mapping)
  uint8_t *wsSig = reinterpret_cast<uint8_t *>(SIGNHIDE ? wsSh + (size_t)n * 4 * wh : wsSh);
  uint8_t *gSig = wsSig + (size_t)tu * wh;
  // written by rdoq_pre_kernel: the Rice parameters (two bits per position) and the last significant position of every block
  const uint8_t *wsRice = wsSig + (size_t)n * wh;
  const int *wsLast = reinterpret_cast<const int *>(wsRice + (size_t)n * (wh >> 2));
  const int16_t *gCoef = coef + (size_t)tu * wh;
  int16_t *sLev = reinterpret_cast<int16_t *>(sDyn + gq * per_tu);
  uint8_t *sLevB = sDyn + gq * per_tu;                                 // BYTES: the same array as bytes
  uint8_t *sRice = sDyn + gq * per_tu + (BYTES ? wh : 2 * wh);         // Rice parameter after each position, four positions per byte
  int16_t *gOut = q_coef + (size_t)tu * wh;                            // BYTES: the exact levels, written and re-read by the same lane
  auto lev_get = [&](int pos) { return BYTES ? (int)sLevB[pos] : (int)sLev[pos]; };
  auto lev_set = [&](int pos, int v) {
    if (BYTES) sLevB[pos] = (uint8_t)(v < 254 ? v : 254 + (v & 1));
    else sLev[pos] = (int16_t)v;
  };
  auto rice_at = [&](int pos) { return (int)(sRice[pos >> 2] >> ((pos & 3) * 2)) & 3; };
  double *D = sStageD[gq];
  int *I = sStageI[gq];
  // ---- stage the coefficients into the level array (coalesced: the blocks of a workgroup are contiguous) ----
  const int l2wh = l2w + l2h_;
  if (!BYTES) {
    // two coefficients per lane and step (wh is a multiple of 16: pairs never straddle blocks; 4-byte aligned: the batch
    // pointer is 2-byte aligned by contract, so pair loads need an even element offset -- tu0 * wh is a multiple of 16)
    const uint32_t *src = reinterpret_cast<const uint32_t *>(coef + (size_t)tu0 * wh);
    const bool aligned = (reinterpret_cast<uintptr_t>(coef) & 3) == 0;
    if (aligned) {
      for (int e = tid; e < (here * wh) >> 1; e += 64) {
        const int b = (2 * e) >> l2wh, pos = 2 * e - (b << l2wh);
        *reinterpret_cast<uint32_t *>(sDyn + b * per_tu + 2 * pos) = src[e];
      }
    } else {
      for (int e = tid; e < here * wh; e += 64) {
        const int b = e >> l2wh, pos = e - (b << l2wh);
        reinterpret_cast<int16_t *>(sDyn + b * per_tu)[pos] = coef[(size_t)tu0 * wh + e];
      }
    }
  }
  __syncthreads();
  // the Rice parameter after each position.  BYTES: from rdoq_pre_kernel (the input block is not in LDS), wh / 16 words per block
  if (BYTES) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(wsRice + (size_t)tu0 * (wh >> 2));
    const int l2wpb = l2wh - 4;                                        // log2(words per block)
    for (int e = tid; e < here << l2wpb; e += 64) {
      const int b = e >> l2wpb, wq = e - (b << l2wpb);
      *reinterpret_cast<uint32_t *>(sDyn + b * per_tu + (BYTES ? wh : 2 * wh) + 4 * wq) = src[e];
    }
  } else {
    // templateAbsSum(coef, 4, ...) over the input block (rdo.c:846-871, 1697)
    if (live)
      for (int q4 = j; q4 < (wh >> 2); q4 += 4) {                        // a lane owns whole bytes of the packed array
        unsigned packed = 0;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
          const int pos = 4 * q4 + k;
          const uint32_t pos_y = (uint32_t)pos >> l2w, pos_x = (uint32_t)pos - (pos_y << l2w);
          const int16_t *c0p = sLev + pos;
          // straight-line: an absent neighbour reads the position itself and counts as zero; coeff_t accumulator: wraps like the reference's
          const bool hx1 = pos_x < (uint32_t)width - 1, hx2 = pos_x < (uint32_t)width - 2;
          const bool hy1 = pos_y < (uint32_t)height - 1, hy2 = pos_y < (uint32_t)height - 2;
          const bool zx1 = mts && pos_x + 1 >= 16, zx2 = mts && pos_x + 2 >= 16, zy1 = mts && pos_y + 1 >= 16, zy2 = mts && pos_y + 2 >= 16;
          const int r0 = c0p[hx1 ? 1 : 0], r1 = c0p[hx2 ? 2 : 0], r2 = c0p[(hx1 && hy1) ? width + 1 : 0], r3 = c0p[hy1 ? width : 0],
                    r4 = c0p[hy2 ? 2 * width : 0];
          const int16_t sum = (int16_t)(((hx1 && !zx1) ? abs(r0) : 0) + ((hx2 && !zx2) ? abs(r1) : 0) + ((hx1 && hy1 && !(zy1 || zx1)) ? abs(r2) : 0) +
                                        ((hy1 && !zy1) ? abs(r3) : 0) + ((hy2 && !zy2) ? abs(r4) : 0));
          packed |= (unsigned)go_rice_par(clampi((int)sum - 20, 0, 31)) << (2 * k);
        }
        sRice[q4] = (uint8_t)packed;
      }
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
  auto level_double_of = [&](int c) {
    const long long prod = (long long)abs(c) * q;
    return (int)(prod < cap ? prod : cap);
  };
  auto cost0_of = [&](int level_double) { const double err = (double)level_double; return err * err * error_scale; };   // cost_coeff0[]
  auto sig_cost_of = [&](int code) { return code ? lambda * (double)B[O_SIG + 12 * t + ((code - 1) >> 1)][(code - 1) & 1] : 0.0; };   // cost_sig[]
  auto cg_skipped = [&](int g) { return mts != 0 && ((g >> l2cgw) >= 4 || (g & (cgw - 1)) >= 4); };
  auto quad_or = [&](unsigned v) { v |= __shfl_xor(v, 1, 64); v |= __shfl_xor(v, 2, 64); return v; };
  auto quad_sum = [&](unsigned v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); return v; };
  auto neighbours = [&](int blkpos, uint32_t pos_x, uint32_t pos_y, int (&nb)[5], bool (&has)[5]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 5; ++k) { nb[k] = 0; has[k] = false; }
    if (pos_x < (uint32_t)width - 1) {
      has[0] = true; nb[0] = lev_get(blkpos + 1);
      if (pos_x < (uint32_t)width - 2) { has[1] = true; nb[1] = lev_get(blkpos + 2); }
      if (pos_y < (uint32_t)height - 1) { has[2] = true; nb[2] = lev_get(blkpos + width + 1); }
    }
    if (pos_y < (uint32_t)height - 1) {
      has[3] = true; nb[3] = lev_get(blkpos + width);
      if (pos_y < (uint32_t)height - 2) { has[4] = true; nb[4] = lev_get(blkpos + 2 * width); }
    }
    if (BYTES) {
      // a clamped neighbour (level >= 254): the template sums of rdoq_decide want the exact value (int16 wrap of the sum,
      // rdo.c:846-871) -- it was stored to the output by another lane of this wave: make that store visible, bypass L1
      // (a neighbour inside the group being walked is not in the output yet: its exact level is in the staging array)
      const int dxs[5] = {1, 2, 1, 0, 0}, dys[5] = {0, 0, 1, 1, 2};
      constexpr unsigned long long kInvDiag4 = 0xFDA6EB73C8419520ull;  // (y * 4 + x) inside a 4x4 group -> index in its scan
      bool any = false;
#pragma unroll
      for (int k = 0; k < 5; ++k) any = any || nb[k] >= 254;
      if (any) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (has[k] && nb[k] >= 254) {
            const uint32_t nx = pos_x + dxs[k], ny = pos_y + dys[k];
            if ((nx >> 2) == (pos_x >> 2) && (ny >> 2) == (pos_y >> 2)) {
              nb[k] = I[3 * (int)((kInvDiag4 >> (4 * ((ny & 3) * 4 + (nx & 3)))) & 15)];
            } else {
              const int p2 = blkpos + dys[k] * width + dxs[k];
              nb[k] = (int)(int16_t)__hip_atomic_load(reinterpret_cast<const uint16_t *>(gOut + p2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
      }
    }
  };

  // ---- the last significant position (rdo.c:1561-1592): highest scan position whose rounded level is non-zero ----
  int last_scanpos = -1;
  if (BYTES) last_scanpos = live ? wsLast[tu] : -1;                     // (rdoq_pre_kernel; the input block is not in LDS)
  else
  for (int cgs = cg_num - 1; cgs >= 0; --cgs) {                        // uniform trip count; the blocks differ only in predicates
    const int g = sScanCg[cgs];
    unsigned m = 0;
    if (live && last_scanpos < 0 && !cg_skipped(g))
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r;
        if (s4 <= max_group) {
          const int ld = level_double_of((int)sLev[blk_in(g, in_cg(s4))]);
          if (((uint32_t)(ld + (1 << (q_bits - 1))) >> q_bits) > 0) m |= 1u << s4;
        }
      }
    m = quad_or(m);
    if (last_scanpos < 0 && m) last_scanpos = cgs * 16 + (31 - __clz((int)m));
  }
  const int cg_last_scanpos = last_scanpos >> 4;                        // -1 >> 4 == -1
  // groups the walk never stages keep reading as zero levels: the MTS zero-out region, and everything of a block without
  // a significant position (its output is all zero)
  WAVE_SYNC();
  if (live && (mts != 0 || last_scanpos < 0 || P.lfnst_idx > 0))
    for (int cgs = 0; cgs < (wh >> 4); ++cgs) {
      const int g = sScanCg[cgs];
      if (last_scanpos < 0 || cg_skipped(g) || cgs >= cg_num)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int bp = blk_in(g, in_cg(j + 4 * r));
          lev_set(bp, 0);
          if (SIGNHIDE) { gSh[bp] = 0; gSh[wh + bp] = 0; gSh[2 * wh + bp] = 0; gSh[3 * wh + bp] = 0; }   // FILL(sh_rates, 0), :1493
        }
      else if (max_group < 15)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (j + 4 * r > max_group) lev_set(blk_in(g, in_cg(j + 4 * r)), 0);
    }
  WAVE_SYNC();

  // ---- the walk, coefficient group by coefficient group (rdo.c:1604-1773) ----
  unsigned long long sig_cg = 0;                                       // sig_coeffgroup_flag as a bit set over group raster positions
  double block_uncoded_cost = 0, base_cost = 0;
  uint32_t reg_bins = (uint32_t)(wh * 28) >> 4;
  int go_rice_state = 0;                                               // only tracked while a group is walked sequentially
  bool exhausted = false;                                              // reg_bins < 4: it never recovers (:1692-1697 stop updating it)
  int pf_in[4] = {0, 0, 0, 0};                                         // BYTES: the input coefficients of the next group to stage
  auto prefetch_in = [&](int cgs2) {
    const int g2 = sScanCg[cgs2];
#pragma unroll
    for (int r = 0; r < 4; ++r) pf_in[r] = (int)gCoef[blk_in(g2, in_cg(j + 4 * r))];
  };
  if (BYTES) prefetch_in(cg_num - 1);
#pragma unroll 1
  for (int cgs = cg_num - 1; cgs >= 0; --cgs) {
    const int g = sScanCg[cgs];
    const bool has_last = live && last_scanpos >= 0;
    const bool in_tail = has_last && cgs > cg_last_scanpos && !cg_skipped(g);   // beyond the last significant position: cost0 sums only (:1585)
    const bool in_walk = has_last && cgs <= cg_last_scanpos && !cg_skipped(g);
    // -- stage: rounded level into dest_coeff (:1620), cost0, the context-free halves of the candidates, bins this group can spend at most --
    unsigned spend = 0;
    if (in_tail || in_walk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r;
        if (s4 > max_group) continue;
        const int blkpos = blk_in(g, in_cg(s4));
        const int scanpos = cgs * 16 + s4;
        const int ld = level_double_of(BYTES ? pf_in[r] : (int)sLev[blkpos]);
        const int mx = (int)((uint32_t)(ld + (1 << (q_bits - 1))) >> q_bits);
        const bool walked = in_walk && scanpos <= last_scanpos;
        const double c0 = cost0_of(ld);
        // straight-line (see the phases): the candidates mx and mx - 1 are staged whether they exist or not -- a phase only takes
        // candidates >= 1 -- and a position beyond the last significant one gets cost_coeff0 / 0 (:1585: base_cost += cost_coeff0)
        // Rice parameter left by the previously visited position (scanpos + 1): reset after every 16th (:1692), else the
        // context-free value of that position; the last significant position starts with 0
        const int rice_next = rice_at(blk_in(g, in_cg((s4 + 1) & 15)));
        const int go_rice = (scanpos == last_scanpos || s4 == 15) ? 0 : rice_next;
        double dist[2]; int rate[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int a = mx - c;
          const double err = (double)(ld - (a * (1 << q_bits)));
          dist[c] = err * err * error_scale;
          // sign + the remainder of abs_level - 4 (rdo.c:520-547)
          const int thr = 5, max_log2 = 15;
          const int symbol = a - 4;
          int golomb = ((symbol >> go_rice) + 1 + go_rice) << 15;
          if (a >= 4 && symbol >= (thr << go_rice)) {                    // escape code: rare
            const uint32_t max_prefix = 32 - (thr + max_log2);
            uint32_t prefix = 0;
            const uint32_t suffix = (uint32_t)(symbol >> go_rice) - thr;
            while (prefix < max_prefix && (int)suffix > ((2 << prefix) - 2)) prefix++;
            const uint32_t suffix_len = prefix == max_prefix ? (uint32_t)(max_log2 - go_rice) : prefix + 1;
            golomb = (int)((thr + prefix + suffix_len + go_rice) << 15);
          }
          rate[c] = (1 << 15) + (a >= 4 ? golomb : 0);
        }
        D[3 * s4] = walked ? dist[0] : c0;
        D[3 * s4 + 1] = walked ? dist[1] : 0.0;
        D[3 * s4 + 2] = c0;
        I[3 * s4] = walked ? rate[0] : 0;                                // (a position that is not walked reads as level 0 in the replay)
        I[3 * s4 + 1] = rate[1];
        I[3 * s4 + 2] = ld;
        lev_set(blkpos, walked ? mx : 0);
        // (what the positions decided BEFORE the group's last one can spend at most, :1695-1696: scan position 0 of a group is the
        // last one decided, and the last significant position spends no significance bin)
        spend += (walked && s4 >= 1) ? (unsigned)(mx < 2 ? mx : 3) + (scanpos != last_scanpos ? 1u : 0u) : 0u;
      }
    if (BYTES && cgs > 0) prefetch_in(cgs - 1);
    spend = quad_sum(spend);
    // Only the group in which the regular-bin budget could fall below 4 is walked position by position: before it the
    // budget cannot run out inside a group; after it reg_bins is frozen below 4, the Rice parameter comes from the template of
    // decided levels (:1645-1648) instead of being carried along the scan, and the positions of an anti-diagonal are
    // independent of each other again.
    const bool slow = in_walk && !exhausted && reg_bins < 4 + spend;
    const bool plain = in_walk && !exhausted && !slow;
    const bool any_slow = __any(slow), any_exh = __any(in_walk && exhausted);
    WAVE_SYNC();
    // -- the positions of one anti-diagonal decide together (their neighbours lie on later anti-diagonals) --
    constexpr int PHASE_UNROLL = (SHAPE == 2 || SHAPE == 3) ? 1 : 7;                   // (4x4: one group, and the unrolled body would spill at three waves per SIMD)
    if (__any(plain))                                                  // (sparse planes: most groups lie beyond every block's last position)
#pragma unroll PHASE_UNROLL
    for (int dg = 6; dg >= 0; --dg) {
      const int cnt = dg <= 3 ? dg + 1 : 7 - dg;
      const int s4 = (dg <= 3 ? dg * (dg + 1) / 2 : 16 - (7 - dg) * (8 - dg) / 2) + j;   // scan index inside the group
      const int scanpos = cgs * 16 + s4;
      if (plain && j < cnt && s4 <= max_group && scanpos <= last_scanpos) {
        const int blkpos = blk_in(g, in_cg(s4));
        const uint32_t pos_y = (uint32_t)blkpos >> l2w, pos_x = (uint32_t)blkpos - (pos_y << l2w);
        // (the byte array holds the clamped level: the rounded level again from the staged |coef| * q)
        const int mx = BYTES ? (int)((uint32_t)(I[3 * s4 + 2] + (1 << (q_bits - 1))) >> q_bits) : (int)sLev[blkpos];
        const bool is_last = scanpos == last_scanpos;
        // Straight-line code (selects instead of branches: with 16 blocks in the wave every branch is taken by some lane anyway,
        // and each divergent branch costs an exec-mask save / restore and a jump).
        // context_get_sig_ctx_idx_abs + the ctx_set line, as rdoq_decide: an absent neighbour reads the position itself and counts as 0
        const bool hx1 = pos_x < (uint32_t)width - 1, hx2 = pos_x < (uint32_t)width - 2;
        const bool hy1 = pos_y < (uint32_t)height - 1, hy2 = pos_y < (uint32_t)height - 2;
        const bool zx1 = mts && pos_x + 1 >= 16, zx2 = mts && pos_x + 2 >= 16, zy1 = mts && pos_y + 1 >= 16;
        const int r0 = lev_get(blkpos + (hx1 ? 1 : 0)), r1 = lev_get(blkpos + (hx2 ? 2 : 0)), r2 = lev_get(blkpos + ((hx1 && hy1) ? width + 1 : 0)),
                  r3 = lev_get(blkpos + (hy1 ? width : 0)), r4 = lev_get(blkpos + (hy2 ? 2 * width : 0));
        // (zero-out tests of the "below" terms use pos_x, as the reference does, rdo.c:1425)
        const int nbv[5] = {(hx1 && !zx1) ? abs(r0) : 0, (hx2 && !zx2) ? abs(r1) : 0, (hx1 && hy1 && !(zy1 || zx1)) ? abs(r2) : 0,
                            (hy1 && !zx1) ? abs(r3) : 0, (hy2 && !zx2) ? abs(r4) : 0};
        int num_pos = 0, sum_abs = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) { sum_abs += min(4 + (nbv[k] & 1), nbv[k]); num_pos += nbv[k] ? 1 : 0; }
        const int diag = (int)(pos_x + pos_y);
        int ctx_sig = min((sum_abs + 1) >> 1, 3) + (diag < 2 ? 4 : 0) + ((t == 0 && diag < 5) ? 4 : 0);
        int ctx_set = (min(sum_abs - num_pos, 4) + 1) + (!diag ? ((t == 0) ? 15 : 5) : (t == 0) ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
        ctx_sig = is_last ? 0 : ctx_sig;                               // the last significant position is the first one visited: no context
        ctx_set = is_last ? 0 : ctx_set;
        const double c0 = D[3 * s4 + 2];
        const uint32_t sig0 = B[O_SIG + 12 * t + ctx_sig][0], sig1 = B[O_SIG + 12 * t + ctx_sig][1];
        const uint32_t par0 = B[O_PAR + 21 * t + ctx_set][0], par1 = B[O_PAR + 21 * t + ctx_set][1];
        const uint32_t g10 = B[O_GT1 + 21 * t + ctx_set][0], g11 = B[O_GT1 + 21 * t + ctx_set][1];
        const uint32_t g20 = B[O_GT2 + 21 * t + ctx_set][0], g21 = B[O_GT2 + 21 * t + ctx_set][1];
        // uvg_get_coded_level, rdo.c:612-640: "not significant" where allowed, then the candidates mx and mx - 1 (those >= 1)
        const bool zero_ok = !is_last && mx < 3;
        const double cs0 = lambda * (double)sig0;
        double cc = zero_ok ? c0 + cs0 : 1.7e+308, cs = zero_ok ? cs0 : 0.0;
        int level = 0, sig_code = zero_ok ? 1 + 2 * ctx_sig : 0;
        const double cur_cost_sig = is_last ? 0.0 : lambda * (double)sig1;
        const int code_sig = is_last ? 0 : 2 + 2 * ctx_sig;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int a = mx - c;                                          // (a < 1: the staged halves are stale, the result is not taken)
          // the context-coded flags of uvg_get_ic_rate (:549-571): gt1; for levels >= 2 parity and gt2
          const int flags = a >= 2 ? (int)g11 + (int)((a & 1) ? par1 : par0) + (int)(a >= 4 ? g21 : g20) : (int)g10;
          const int rate = I[3 * s4 + c] + flags;
          double cur = D[3 * s4 + c] + lambda * (double)rate;
          cur += cur_cost_sig;
          const bool take = a >= 1 && cur < cc;
          level = take ? a : level; cc = take ? cur : cc; cs = take ? cur_cost_sig : cs; sig_code = take ? code_sig : sig_code;
        }
        if (SIGNHIDE) {
          const int go_rice = (is_last || s4 == 15) ? 0 : rice_at(blk_in(g, in_cg(s4 + 1)));
          sh_record(gSh, wh, blkpos, B, t, is_last, level, I[3 * s4 + 2], q_bits, ctx_sig, ctx_set, go_rice, 4);
        }
        lev_set(blkpos, level);
        D[3 * s4] = cc; D[3 * s4 + 1] = cs;
        I[3 * s4] = level;
        I[3 * s4 + 1] = sig_code;                                        // (the candidates' rate halves have been consumed)
      }
      WAVE_SYNC();
    }
    // -- the same with the budget spent: full uvg_get_coded_level with reg_bins < 4 (bypass-coded levels, Rice parameter
    //    from the decided neighbours) --
    if (any_exh)
      for (int dg = 6; dg >= 0; --dg) {
        const int cnt = dg <= 3 ? dg + 1 : 7 - dg;
        const int s4 = (dg <= 3 ? dg * (dg + 1) / 2 : 16 - (7 - dg) * (8 - dg) / 2) + j;
        const int scanpos = cgs * 16 + s4;
        if (in_walk && exhausted && j < cnt && s4 <= max_group && scanpos <= last_scanpos) {
          const int blkpos = blk_in(g, in_cg(s4));
          const uint32_t pos_y = (uint32_t)blkpos >> l2w, pos_x = (uint32_t)blkpos - (pos_y << l2w);
          const int ld = I[3 * s4 + 2];
          const uint32_t mx = (uint32_t)(ld + (1 << (q_bits - 1))) >> q_bits;
          int nb[5]; bool has[5];
          neighbours(blkpos, pos_x, pos_y, nb, has);
          const rdoq_decision d = rdoq_decide(P, B, t, scanpos == last_scanpos, ld, mx, D[3 * s4 + 2], nb, has, pos_x, pos_y, 0, reg_bins);
          if (SIGNHIDE) sh_record(gSh, wh, blkpos, B, t, scanpos == last_scanpos, d.level, ld, q_bits, d.ctx_sig, d.ctx_set, d.go_rice, reg_bins);
          lev_set(blkpos, d.level);
          D[3 * s4] = d.coded_cost; D[3 * s4 + 1] = d.coded_sig;
          I[3 * s4] = d.level;
          I[3 * s4 + 1] = d.sig_code;
        }
        WAVE_SYNC();
      }
    // -- the group that may run out of bins: the lanes of the block walk its positions in order (identical results in all four) --
    if (any_slow) {
      go_rice_state = 0;                                                 // every group starts from 0 (:1692: reset after scan position 16k)
      for (int s2 = max_group; s2 >= 0; --s2) {
        const int sc2 = cgs * 16 + s2;
        const bool act2 = slow && sc2 <= last_scanpos;
        rdoq_decision d2;
        d2.level = 0; d2.sig_code = 0; d2.coded_cost = 0; d2.coded_sig = 0;
        const int b2 = blk_in(g, in_cg(s2));
        if (act2) {
          const uint32_t py = (uint32_t)b2 >> l2w, px = (uint32_t)b2 - (py << l2w);
          const int ld2 = I[3 * s2 + 2];
          const uint32_t mx2 = (uint32_t)(ld2 + (1 << (q_bits - 1))) >> q_bits;
          int nb[5]; bool has[5];
          neighbours(b2, px, py, nb, has);
          const bool last2 = sc2 == last_scanpos;
          d2 = rdoq_decide(P, B, t, last2, ld2, mx2, D[3 * s2 + 2], nb, has, px, py, go_rice_state, reg_bins);
          if (SIGNHIDE && j == 0) sh_record(gSh, wh, b2, B, t, last2, d2.level, ld2, q_bits, d2.ctx_sig, d2.ctx_set, d2.go_rice, reg_bins);
          // context set update (rdo.c:1691-1699), tracked here because the budget is nearly spent
          if ((sc2 % 16 == 0) && sc2 > 0) go_rice_state = 0;
          else if (reg_bins >= 4) {
            reg_bins -= (uint32_t)((d2.level < 2 ? d2.level : 3) + (last2 ? 0 : 1));
            go_rice_state = rice_at(b2);
          }
        }
        WAVE_SYNC();
        if (act2 && j == 0) {
          lev_set(b2, d2.level);
          D[3 * s2] = d2.coded_cost; D[3 * s2 + 1] = d2.coded_sig;
          I[3 * s2] = d2.level;
          I[3 * s2 + 1] = d2.sig_code;
        }
        WAVE_SYNC();
      }
    }
    // cost_coeff[] and the cost_sig[] code of the group's walked positions
    if (in_walk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r;
        if (s4 <= max_group && cgs * 16 + s4 <= last_scanpos) {
          const int bp = blk_in(g, in_cg(s4));
          // (the workspace arrays are private to the kernel: indexed by SCAN position, so that the 16 positions of a group are 128
          //  / 16 contiguous bytes -- with the block's own raster order every store was a partial 32-byte sector)
          //  (4x4 / 8x8 blocks keep the raster index: their few groups gain little and the extra index costs the 168-register fit)
          const int wi = SCAN_WS ? cgs * 16 + s4 : bp;
          gCost[wi] = D[3 * s4];
          gSig[wi] = (uint8_t)I[3 * s4 + 1];
          if (BYTES) gOut[bp] = (int16_t)I[3 * s4];
        }
      }
    // -- replay the group's costs in scan order (bit-exact double sums).  The five running sums are independent chains:
    //    lane 0 of the block carries base_cost (+ cost_coeff), lane 1 block_uncoded_cost (+ cost_coeff0), lane 2 the group's
    //    sig_cost (+ cost_sig) and uncoded_dist (+ cost_coeff0 of non-zero levels), lane 3 coded_level_and_dist
    //    (+ cost_coeff - cost_sig of non-zero levels); x + 0.0 == x exactly, so "skip" is "add zero".  Then the sums are
    //    exchanged inside the quad and every lane takes the group decision. --
    unsigned m = 0;                                                      // non-zero levels of the group
    double acc_a = 0.0, acc_b = 0.0;
    if (in_tail || in_walk) {
      unsigned dec = 0;                                                  // regular bins the group spends
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r;
        if (s4 > max_group) continue;
        const int lv = I[3 * s4];
        if (lv) m |= 1u << s4;
        const int sc = cgs * 16 + s4;
        if (in_walk && sc <= last_scanpos && !(s4 == 0 && cgs > 0)) dec += (unsigned)(lv < 2 ? lv : 3) + (sc == last_scanpos ? 0u : 1u);
      }
      m = quad_or(m); dec = quad_sum(dec);
      // (:1692-1697) inside a group that is not walked sequentially the budget stays >= 4 (reg_bins >= 4 + spend >= 4 + dec)
      if (in_walk && !slow && reg_bins >= 4) reg_bins -= dec;
      const int o1 = j == 1 ? 2 : j == 2 ? 1 : 0, o2 = j == 2 ? 2 : 1;
      acc_a = j == 0 ? base_cost : j == 1 ? block_uncoded_cost : 0.0;
      const double *Dp = D + o1, *Dq = D + o2;
      constexpr int PW = 8;                                              // eight positions' operands in flight at a time
#pragma unroll
      for (int part = 16 / PW - 1; part >= 0; --part) {
        double pv[PW], qv[PW];
#pragma unroll
        for (int u = PW - 1; u >= 0; --u) { pv[u] = Dp[3 * (PW * part + u)]; qv[u] = Dq[3 * (PW * part + u)]; }   // (past max_group: stale, unused)
#pragma unroll
        for (int u = PW - 1; u >= 0; --u) {
          const int s2 = PW * part + u;
          if (s2 > max_group) continue;
          const bool nz = (m >> s2) & 1;
          const double dv = pv[u] - qv[u];
          acc_a += j == 3 ? (nz ? dv : 0.0) : pv[u];
          acc_b += (j == 2 && nz) ? qv[u] : 0.0;
        }
      }
      base_cost = quad_bcast<0>(acc_a);
      block_uncoded_cost = quad_bcast<1>(acc_a);
    }
    if (in_walk) {
      double rd_sig = quad_bcast<2>(acc_a), rd_uncoded = quad_bcast<2>(acc_b), rd_coded = quad_bcast<3>(acc_a);
      const double rd_sig0 = D[1];
      const int nnz_before_pos0 = __popc(m & ~1u);
      const bool any_level = m != 0;
      if (reg_bins < 4) exhausted = true;
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
      if (j == 0) gCgCost[cgs] = cg_cost;                                // cost_coeffgroup_sig[cgs]
      if (zero_it)                                                       // reset the group (:1752-1762): cost_coeff = cost_coeff0, cost_sig = 0
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s4 = j + 4 * r;
          const int blkpos = blk_in(g, in_cg(s4));
          if (s4 <= max_group && lev_get(blkpos)) { lev_set(blkpos, 0); const int wi = SCAN_WS ? cgs * 16 + s4 : blkpos; gCost[wi] = D[3 * s4 + 2]; gSig[wi] = 0; if (BYTES) gOut[blkpos] = 0; }
        }
    } else if (has_last && j == 0) {
      gCgCost[cgs] = 0;                                                  // groups skipped by the MTS zero-out keep a zero flag cost
    }
    WAVE_SYNC();
  }

  // ---- last position (rdo.c:1775-1829): one short sequential pass, identical in the lanes of a block ----
  int best_last_idx_p1 = 0;
  if (live && last_scanpos >= 0) {
    double best_cost;
    if (P.block_type != 1 && !P.color) {
      best_cost = block_uncoded_cost + lambda * (double)B[S_ROOT][0];
      base_cost += lambda * (double)B[S_ROOT][1];
    } else {
      const int m = P.color == 0 ? S_CBF_Y : P.color == 1 ? S_CBF_CB : S_CBF_CR + (P.cbf_u ? 1 : 0);
      best_cost = block_uncoded_cost + lambda * (double)B[m][0];
      base_cost += lambda * (double)B[m][1];
    }
    bool found_last = false;
    // Per group the four lanes fetch what the walk over its positions needs -- cost_coeff (workspace), cost_coeff0 (from the
    // input coefficient), cost_sig (from the code kept in the meta byte) and the bits of the last-position syntax -- one
    // group ahead of the sequential pass, which is then a handful of double operations per position.
    double pf_cost[4], pf_cg = 0.0; int pf_coef[4], pf_sig[4];
    auto prefetch = [&](int cgs) {
      const int g = sScanCg[cgs];
      if (j == 0) pf_cg = gCgCost[cgs];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r;
        const int b2 = blk_in(g, in_cg(s4 <= max_group ? s4 : 0));
        const int sp = SCAN_WS ? cgs * 16 + (s4 <= max_group ? s4 : 0) : b2;
        pf_cost[r] = gCost[sp]; pf_coef[r] = (int)gCoef[b2]; pf_sig[r] = (int)gSig[sp];
      }
    };
    prefetch(cg_last_scanpos);
#pragma unroll 1
    for (int cgs = cg_last_scanpos; cgs >= 0 && !found_last; cgs--) {
      const int g = sScanCg[cgs];
      base_cost -= quad_bcast<0>(pf_cg);
      const bool coded = (sig_cg >> g) & 1;
      unsigned mnz = 0, mgt1 = 0;
      // straight-line: per position what the pass subtracts from base_cost (cost_coeff of a level, cost_sig of a zero), what it
      // adds back (cost_coeff0 of a level), cost_sig, and the bits of "last position here"; nothing for positions past the
      // last significant one
      if (coded)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s4 = j + 4 * r;
          const bool valid = s4 <= max_group && cgs * 16 + s4 <= last_scanpos;
          const int b2 = blk_in(g, in_cg(s4 <= max_group ? s4 : 0));
          const int lv = lev_get(b2);
          const bool nz = valid && lv != 0;
          const double cs = sig_cost_of(valid ? pf_sig[r] : 0);
          const double c0 = cost0_of(level_double_of(pf_coef[r]));
          D[3 * s4] = nz ? pf_cost[r] : cs;
          D[3 * s4 + 1] = cs;
          D[3 * s4 + 2] = nz ? c0 : 0.0;
          const uint32_t py = (uint32_t)b2 >> l2w, px = (uint32_t)b2 - (py << l2w);
          I[3 * s4 + 1] = sLastXp[px] + sLastYp[py];                     // get_rate_last without the final lambda *
          mnz |= (nz ? 1u : 0u) << s4;
          mgt1 |= ((nz && lv > 1) ? 1u : 0u) << s4;
        }
      if (cgs > 0) prefetch(cgs - 1);
      WAVE_SYNC();
      if (coded) {
        mnz = quad_or(mnz); mgt1 = quad_or(mgt1);
#pragma unroll
        for (int part = 3; part >= 0; --part) {
          double sub[4], add[4], csv[4]; int ui[4];
#pragma unroll
          for (int u = 3; u >= 0; --u) {
            const int s2 = 4 * part + u;
            sub[u] = D[3 * s2]; csv[u] = D[3 * s2 + 1]; add[u] = D[3 * s2 + 2]; ui[u] = I[3 * s2 + 1];
          }
#pragma unroll
          for (int u = 3; u >= 0; --u) {
            const int s2 = 4 * part + u, sc2 = cgs * 16 + s2;
            const bool nz = (mnz >> s2) & 1;
            const double total = base_cost + lambda * (double)ui[u] - csv[u];
            const bool upd = nz && !found_last && total < best_cost;
            best_last_idx_p1 = upd ? sc2 + 1 : best_last_idx_p1;
            best_cost = upd ? total : best_cost;
            found_last = found_last || ((mgt1 >> s2) & 1);               // (the reference stops here; base_cost is not used any more)
            base_cost -= sub[u];
            base_cost += add[u];
          }
        }
      }
      WAVE_SYNC();
    }
  }
  WAVE_SYNC();
  // ---- signs, clean-up, outputs (rdo.c:1831-1858): parallel over the positions ----
  uint32_t my_abs = 0;
  if (live) {
    const bool reduce = mts && !(width < 32 && height < 32);
    for (int cgs = 0; cgs < (wh >> 4); ++cgs)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s4 = j + 4 * r, scanpos = cgs * 16 + s4;
        const int g2 = sScanCg[cgs];
        const int b = blk_in(g2, in_cg(s4));
        int level;
        if (BYTES) {                                                     // the exact level is in the output already where the walk decided one
          const bool decided = last_scanpos >= 0 && scanpos < best_last_idx_p1 && cgs < cg_num && s4 <= max_group && !cg_skipped(g2);
          level = decided ? (int)gOut[b] : 0;
        } else {
          level = sLev[b];
          if (last_scanpos < 0 || scanpos >= best_last_idx_p1) level = 0;
        }
        if (level && reduce) { const int bx = b & (width - 1), by = b >> l2w; if (bx >= 16 || by >= 16) level = 0; }
        my_abs += (uint32_t)level;
        const int signed_level = (level != 0 && gCoef[b] < 0) ? -level : level;
        if (BYTES) gOut[b] = (int16_t)signed_level;
        else sLev[b] = (int16_t)signed_level;
      }
  }
  my_abs = quad_sum(my_abs);
  if (live && j == 0) {
    if (abs_sum_out) abs_sum_out[tu] = my_abs;
    if (has_coeffs) has_coeffs[tu] = my_abs ? 1 : 0;
  }
  // ---- sign-data hiding (uvg_rdoq_sign_hiding, rdo.c:700-845): per coefficient group, groups do not interact: the four lanes
  //      of a block take every fourth group ----
  if (SIGNHIDE) {
    __threadfence_block();                                             // the sh_rates records of the other lanes
    WAVE_SYNC();
    if (live && my_abs >= 2) {
      const long long rd_factor = P.rd_factor;
      const int last_cg = (best_last_idx_p1 - 1) >> 4;
      for (int cgk = j; cgk <= last_cg; cgk += 4) {
        const int g = sScanCg[cgk];
        int last_nz = -1, first_nz = 16;
        for (int k = 15; k >= 0; --k) if (sLev[blk_in(g, in_cg(k))]) { last_nz = k; break; }
        for (int k = 0; k <= last_nz; ++k) if (sLev[blk_in(g, in_cg(k))]) { first_nz = k; break; }
        if (last_nz - first_nz < 4) continue;                            // SBH_THRESHOLD
        const int signbit = sLev[blk_in(g, in_cg(first_nz))] <= 0;
        unsigned sum = 0;
        for (int k = first_nz; k <= last_nz; ++k) sum += (unsigned)(int)sLev[blk_in(g, in_cg(k))];
        if (signbit == (int)(sum & 1)) continue;
        long long best_cost = 0x7fffffffffffffffll;
        int best_pos = 0, best_change = 0;
        for (int k = (cgk == last_cg ? last_nz : 15); k >= 0; --k) {
          const int pos = blk_in(g, in_cg(k));
          const long long qbits = rd_factor * (long long)gSh[3 * wh + pos];
          const int a = abs((int)sLev[pos]);
          long long cost;
          int change;
          if (a != 0) {
            long long inc_bits = gSh[pos], dec_bits = gSh[wh + pos];
            if (a == 1) dec_bits -= gSh[2 * wh + pos];
            if (cgk == last_cg && last_nz == k && a == 1) dec_bits -= 4 * 32768;
            inc_bits = -qbits + inc_bits;
            dec_bits = qbits + dec_bits;
            if (inc_bits < dec_bits) { change = 1; cost = inc_bits; }
            else {
              change = -1; cost = dec_bits;
              if (k == first_nz && a == 1) cost = 0x7fffffffffffffffll;
            }
          } else {
            const int bits = 32768 + gSh[pos] + gSh[2 * wh + pos];
            cost = -(qbits < 0 ? -qbits : qbits) + bits;
            change = 1;
            if (k < first_nz && ((gCoef[pos] >= 0) ? 0 : 1) != signbit) cost = 0x7fffffffffffffffll;
          }
          if (cost < best_cost) { best_cost = cost; best_pos = pos; best_change = change; }
        }
        const int qv = sLev[best_pos];
        if (qv == 32767 || qv == -32768) best_change = -1;
        sLev[best_pos] = (int16_t)(gCoef[best_pos] >= 0 ? qv + best_change : qv - best_change);
      }
    }
  }
  __syncthreads();
  if (BYTES) {
  } else if ((reinterpret_cast<uintptr_t>(q_coef) & 3) == 0) {
    uint32_t *dst = reinterpret_cast<uint32_t *>(q_coef + (size_t)tu0 * wh);
    for (int e = tid; e < (here * wh) >> 1; e += 64) {
      const int b = (2 * e) >> l2wh, pos = 2 * e - (b << l2wh);
      dst[e] = *reinterpret_cast<const uint32_t *>(sDyn + b * per_tu + 2 * pos);
    }
  } else {
    for (int e = tid; e < here * wh; e += 64) {
      const int b = e >> l2wh, pos = e - (b << l2wh);
      q_coef[(size_t)tu0 * wh + e] = reinterpret_cast<const int16_t *>(sDyn + b * per_tu)[pos];
    }
  }
  __syncthreads();
  }   // batches
}

}  // namespace

extern "C" size_t uvghip_rdoq_workspace_bytes(int width, int height, int n)
{
  if (width <= 0 || height <= 0 || n <= 0) return 0;
  // cost_coeff[] + cost_coeffgroup_sig[] + the cost_sig[] codes of every block
  // + the Rice parameters (2 bits per position) and the last significant position of every block (rdoq_pre_kernel)
  return (size_t)width * height * (size_t)n * (sizeof(double) + 1) + (size_t)(width * height / 16) * (size_t)n * sizeof(double) +
         (size_t)(width * height / 4) * (size_t)n + (size_t)n * sizeof(int);
}

extern "C" size_t uvghip_rdoq_signhide_workspace_bytes(int width, int height, int n)
{
  if (width <= 0 || height <= 0 || n <= 0) return 0;
  return uvghip_rdoq_workspace_bytes(width, height, n) + (size_t)width * height * (size_t)n * 4 * sizeof(int32_t);   // + sh_rates of every block
}

static int rdoq_launch(int signhide, int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n, int color,
                       int block_type, int cbf_u, int lfnst_idx, int mts_idx, int qp_scaled, double lambda,
                       const uvghip_rdoq_ctx_t *ctx_host, void *workspace, size_t workspace_bytes, uint32_t *abs_sum_out,
                       uint8_t *has_coeffs, void *stream);

extern "C" int uvghip_rdoq_batch(int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n, int color,
                                 int block_type, int cbf_u, int lfnst_idx, int mts_idx, int qp_scaled, double lambda,
                                 const uvghip_rdoq_ctx_t *ctx_host, void *workspace, size_t workspace_bytes,
                                 uint32_t *abs_sum_out, uint8_t *has_coeffs, void *stream)
{
  return rdoq_launch(0, bitdepth, coef, q_coef, width, height, n, color, block_type, cbf_u, lfnst_idx, mts_idx, qp_scaled, lambda, ctx_host,
                     workspace, workspace_bytes, abs_sum_out, has_coeffs, stream);
}

extern "C" int uvghip_rdoq_signhide_batch(int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n, int color,
                                          int block_type, int cbf_u, int lfnst_idx, int mts_idx, int qp_scaled, double lambda,
                                          const uvghip_rdoq_ctx_t *ctx_host, void *workspace, size_t workspace_bytes,
                                          uint32_t *abs_sum_out, uint8_t *has_coeffs, void *stream)
{
  return rdoq_launch(1, bitdepth, coef, q_coef, width, height, n, color, block_type, cbf_u, lfnst_idx, mts_idx, qp_scaled, lambda, ctx_host,
                     workspace, workspace_bytes, abs_sum_out, has_coeffs, stream);
}

static int rdoq_launch(int signhide, int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n, int color,
                       int block_type, int cbf_u, int lfnst_idx, int mts_idx, int qp_scaled, double lambda,
                       const uvghip_rdoq_ctx_t *ctx_host, void *workspace, size_t workspace_bytes, uint32_t *abs_sum_out,
                       uint8_t *has_coeffs, void *stream)
{
  UVGHIP_REQUIRE_READY();
  auto pow2 = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32; };
  if ((bitdepth != 8 && bitdepth != 10) || !pow2(width) || !pow2(height) || color < 0 || color > 2 || !ctx_host || !coef || !q_coef ||
      qp_scaled < 0 || lfnst_idx < 0 || lfnst_idx > 2 || mts_idx < 0 || !(lambda >= 0))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  if (!workspace || workspace_bytes < (signhide ? uvghip_rdoq_signhide_workspace_bytes(width, height, n) : uvghip_rdoq_workspace_bytes(width, height, n)))
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_rdoq_batch: workspace");
  if (signhide && !(lambda > 0)) return uvghip_set_error(hipErrorInvalidValue, "uvghip_rdoq_signhide_batch: lambda");
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
  P.signhide = signhide;
  P.rd_factor = 0;
  if (signhide) {
    // rdo.c:721-726.  The reference forms inv_quant^2 * 2^(2 (qp / 6)) in int (it wraps for qp_scaled >= 54) and divides in double.
    static const int iqs[2][6] = {{40, 45, 51, 57, 64, 72}, {57, 64, 72, 80, 90, 102}};
    const int inv_quant = iqs[sqrt2][qp_scaled % 6];
    const int32_t prod = (int32_t)((uint32_t)(inv_quant * inv_quant) * (1u << (2 * (qp_scaled / 6))));
    P.rd_factor = (long long)(prod / lambda / 16 / (1 << (2 * (bitdepth - 8))) + 0.5);
  }
  const int wh = width * height;
  // blocks per wave (four lanes each): 16.  LDS per block: levels (int16) + meta (byte) per position, odd word stride
  const int tus = 16;
  const size_t lds = (size_t)tus * rdoq_lds_per_block(wh, width == 32 && height == 32 && !signhide);
  double *w = static_cast<double *>(workspace);
  hipStream_t st = uvghip_stream(stream);
  // one workgroup per batch of 16 blocks (measured: capping the grid and looping batches inside a workgroup to share the
  // table set-up is slower -- the hardware dispatcher balances the uneven block times better); the in-kernel batch loop only
  // matters beyond 2^20 batches
  const int batches = (n + tus - 1) / tus;
  const int grid = batches < (1 << 20) ? batches : (1 << 20);
  if (width == 32 && height == 32 && !signhide) {
    // byte level array (rdoq_lds_per_block): the decision-free part first (Rice parameters, last significant position), into the
    // tail of the workspace
    uint8_t *ws_sig = reinterpret_cast<uint8_t *>(w + (size_t)n * wh + (size_t)n * (wh >> 4)) + (signhide ? (size_t)n * 4 * wh * sizeof(int32_t) : 0);
    uint8_t *ws_rice = ws_sig + (size_t)n * wh;
    int *ws_last = reinterpret_cast<int *>(ws_rice + (size_t)n * (wh >> 2));
    const long long chunks = ((long long)n * wh + 1023) / 1024;
    if (chunks > 0x7fffffffLL) return uvghip_set_error(hipErrorInvalidValue, "uvghip_rdoq_batch: n");
    rdoq_pre_kernel<<<(int)chunks, 256, 0, st>>>(P, coef, ws_rice, ws_last);
    { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return uvghip_set_error(e__, "uvghip_rdoq_batch: rdoq_pre_kernel"); }
  }
#define RDOQ_LAUNCH(SH, CH)                                                                                              \
  do {                                                                                                                   \
    if (signhide) rdoq_kernel<16, SH, CH, 1><<<grid, 64, lds, st>>>(P, coef, q_coef, w, abs_sum_out, has_coeffs);        \
    else rdoq_kernel<16, SH, CH, 0><<<grid, 64, lds, st>>>(P, coef, q_coef, w, abs_sum_out, has_coeffs);                 \
  } while (0)
  const int shape = width == height ? P.l2w : 0;
  const bool ch = color != 0;
  switch (shape) {
    case 2: if (ch) RDOQ_LAUNCH(2, 1); else RDOQ_LAUNCH(2, 0); break;
    case 3: if (ch) RDOQ_LAUNCH(3, 1); else RDOQ_LAUNCH(3, 0); break;
    case 4: if (ch) RDOQ_LAUNCH(4, 1); else RDOQ_LAUNCH(4, 0); break;
    case 5: if (ch) RDOQ_LAUNCH(5, 1); else RDOQ_LAUNCH(5, 0); break;
    default: RDOQ_LAUNCH(0, 0); break;
  }
#undef RDOQ_LAUNCH
  UVGHIP_CHECK_LAUNCH();
}
