/*
 * oracle/orc_rdoq.c -- restatement of rate-distortion optimised quantisation:
 *   uvg_rdoq                 src/rdo.c:1449-1870   (VTM-13 RDOQ: regular residual coding, no transform skip)
 *   uvg_get_coded_level      src/rdo.c:597-640
 *   uvg_get_ic_rate          src/rdo.c:465-581
 *   get_rate_last / calc_last_bits   src/rdo.c:651-700
 *   context_get_sig_ctx_idx_abs / templateAbsSum   src/rdo.c:1400-1438, 846-871
 *   uvg_context_get_sig_coeff_group  src/context.c:647-660
 * on plain arguments: what the reference reads from encoder_state_t is passed in -- qp_scaled
 * (uvg_get_scaled_qp, transform.c:150), lambda (state->lambda / c_lambda) and a snapshot of the CABAC context
 * models it prices bins with (state->cabac.ctx, cabac.h:60-131) reduced to CTX_STATE(ctx) = (state[0]+state[1])>>8
 * per model (cabac.h:175-176), which is all CTX_ENTROPY_BITS (rdo.h:106) looks at.
 * Configuration subset: scaling lists off, sign-data hiding off (signhide = 0 in the target presets), diagonal scan
 * (the only scan uvg_get_scan_order_table implements, tables.c:2649).
 *
 * The arithmetic is double precision exactly as the reference writes it (products and sums in its order, no
 * contraction); tables are computed from their definitions:
 *   uvg_entropy_bits[2s+v] = round(-log2(p_v) * 2^15), p_1 = (2s+1)/512           (rdo.c:75-140, VTM BinFracBits)
 *   g_auiGoRiceParsCoeff[s] = (s >= 7) + (s >= 14) + (s >= 28)                     (rdo.c:67-71)
 *   g_group_idx / g_min_in_group: the last-position prefix groups                  (H.266 9.3.3.x)
 *   diagonal scans: H.266 6.5.2 (up-right diagonal) over 4x4 coefficient groups    (tables.c g_scan_order)
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

/* CTX_STATE of every context model uvg_rdoq prices bins with; [0] luma, [1] chroma where split */
typedef struct orc_rdoq_ctx {
  uint8_t sig_group[2][2];   /* sig_coeff_group_model[0..1] / [2..3] */
  uint8_t sig[2][12];        /* cu_sig_model_luma[0][12] / cu_sig_model_chroma[0][8] */
  uint8_t par[2][21];        /* cu_parity_flag_model_luma[21] / _chroma[11] */
  uint8_t gt1[2][21];        /* cu_gtx_flag_model_*[1][..] */
  uint8_t gt2[2][21];        /* cu_gtx_flag_model_*[0][..] */
  uint8_t last_x[2][20];     /* cu_ctx_last_x_luma[20] / _chroma[3] */
  uint8_t last_y[2][20];
  uint8_t cbf_luma[4], cbf_cb[2], cbf_cr[3];
  uint8_t root_cbf;
} orc_rdoq_ctx;

static uint32_t entropy_bits(int state, int val)
{
  const double p1 = (2 * state + 1) / 512.0;
  return (uint32_t)floor(-log2(val ? p1 : 1.0 - p1) * 32768.0 + 0.5);
}
#define BITS(st, val) entropy_bits((st), (val))

static int group_idx(int pos)
{
  if (pos < 4) return pos;
  int l = 0;
  while ((pos >> (l + 1)) != 0) ++l;
  return 2 * l + ((pos >> (l - 1)) & 1);
}

/* H.266 6.5.2: up-right diagonal scan of a bw x bh array -> raster positions */
static void diag_scan(int bw, int bh, int *out)
{
  int i = 0, x = 0, y = 0, stop = 0;
  while (!stop) {
    while (y >= 0) {
      if (x < bw && y < bh) { out[i++] = y * bw + x; }
      y--; x++;
    }
    y = x; x = 0;
    if (i >= bw * bh) stop = 1;
  }
}

/* coefficient scan in 4x4 groups (SCAN_GROUP_4X4) and the group scan (SCAN_GROUP_UNGROUPED on the group grid) */
ORC_EXPORT void ORC_FN(rdoq_scans)(int width, int height, uint32_t *scan, uint32_t *scan_cg)
{
  int cgw = width >> 2, cgh = height >> 2, cg[64], in[16];
  diag_scan(cgw, cgh, cg);
  diag_scan(4, 4, in);
  for (int g = 0; g < cgw * cgh; ++g) {
    scan_cg[g] = (uint32_t)cg[g];
    const int gx = cg[g] % cgw, gy = cg[g] / cgw;
    for (int k = 0; k < 16; ++k) scan[g * 16 + k] = (uint32_t)((gy * 4 + in[k] / 4) * width + gx * 4 + in[k] % 4);
  }
}

static uint32_t sig_ctx_idx_abs(const int16_t *coeff, uint32_t pos_x, uint32_t pos_y, uint32_t width, uint32_t height, int color,
                                int32_t *temp_diag, int32_t *temp_sum, int mts)
{
  const int16_t *data = coeff + pos_x + pos_y * width;
  const int diag = (int)(pos_x + pos_y);
  int num_pos = 0, sum_abs = 0;
#define UPDATE(x) { int a = abs(x); sum_abs += (4 + (a & 1)) < a ? (4 + (a & 1)) : a; num_pos += (a ? 1 : 0); }
  if (pos_x < width - 1) {
    UPDATE(mts && pos_x + 1 >= 16 ? 0 : data[1]);
    if (pos_x < width - 2) UPDATE(mts && pos_x + 2 >= 16 ? 0 : data[2]);
    if (pos_y < height - 1) UPDATE(mts && (pos_y + 1 >= 16 || pos_x + 1 >= 16) ? 0 : data[width + 1]);
  }
  if (pos_y < height - 1) {
    UPDATE(mts && pos_x + 1 >= 16 ? 0 : data[width]);             /* sic: the reference tests pos_x here (rdo.c:1425) */
    if (pos_y < height - 2) UPDATE(mts && pos_x + 2 >= 16 ? 0 : data[width << 1]);
  }
#undef UPDATE
  int ctx_ofs = ((sum_abs + 1) >> 1 < 3 ? (sum_abs + 1) >> 1 : 3) + (diag < 2 ? 4 : 0);
  if (color == 0) ctx_ofs += diag < 5 ? 4 : 0;
  *temp_diag = diag;
  *temp_sum = sum_abs - num_pos;
  return (uint32_t)ctx_ofs;
}

static unsigned template_abs_sum(const int16_t *coeff, int base_level, uint32_t pos_x, uint32_t pos_y, uint32_t width, uint32_t height, int mts)
{
  const int16_t *p = coeff + pos_x + pos_y * width;
  int16_t sum = 0;                                   /* coeff_t accumulator, as in the reference (rdo.c:849) */
  if (pos_x < width - 1) {
    sum = (int16_t)(sum + (mts && pos_x + 1 >= 16 ? 0 : abs(p[1])));
    if (pos_x < width - 2) sum = (int16_t)(sum + (mts && pos_x + 2 >= 16 ? 0 : abs(p[2])));
    if (pos_y < height - 1) sum = (int16_t)(sum + (mts && (pos_y + 1 >= 16 || pos_x + 1 >= 16) ? 0 : abs(p[width + 1])));
  }
  if (pos_y < height - 1) {
    sum = (int16_t)(sum + (mts && pos_y + 1 >= 16 ? 0 : abs(p[width])));
    if (pos_y < height - 2) sum = (int16_t)(sum + (mts && pos_y + 2 >= 16 ? 0 : abs(p[width << 1])));
  }
  int v = sum - 5 * base_level;
  v = v < 31 ? v : 31;
  return (unsigned)(v > 0 ? v : 0);
}

static int go_rice_par(unsigned s) { return (s >= 7) + (s >= 14) + (s >= 28); }

/* rdo.c:465-581, use_limited_prefix_length as a parameter */
static int32_t ic_rate(const orc_rdoq_ctx *c, int t, uint32_t abs_level, int ctx_gt1, int ctx_gt2, int ctx_par, int go_rice,
                       uint32_t reg_bins, int limited)
{
  int32_t rate = 1 << 15;
  const uint32_t go_rice_zero = 1u << go_rice;
  const int max_log2 = 15, thr = 5;
  if (reg_bins < 4) {
    uint32_t symbol = (abs_level == 0 ? go_rice_zero : abs_level <= go_rice_zero ? abs_level - 1 : abs_level);
    uint32_t length;
    if (symbol < ((uint32_t)thr << go_rice)) {
      length = symbol >> go_rice;
      rate += (int32_t)((length + 1 + go_rice) << 15);
    } else if (limited) {
      const uint32_t max_prefix = 32 - (thr + max_log2);
      uint32_t prefix = 0, suffix = (symbol >> go_rice) - thr;
      while (prefix < max_prefix && (int32_t)suffix > ((2 << prefix) - 2)) prefix++;
      const uint32_t suffix_len = prefix == max_prefix ? (uint32_t)(max_log2 - go_rice) : prefix + 1;
      rate += (int32_t)((thr + prefix + suffix_len + go_rice) << 15);
    } else {
      length = (uint32_t)go_rice;
      symbol = symbol - ((uint32_t)thr << go_rice);
      while ((int32_t)symbol >= (1 << length)) symbol -= (1u << (length++));
      rate += (int32_t)((thr + length + 1 - go_rice + length) << 15);
    }
    return rate;
  }
  if (abs_level >= 4) {
    int32_t symbol = (int32_t)abs_level - 4, length;
    if (symbol < (thr << go_rice)) {
      length = symbol >> go_rice;
      rate += (length + 1 + go_rice) << 15;
    } else if (limited) {
      const uint32_t max_prefix = 32 - (thr + max_log2);
      uint32_t prefix = 0, suffix = (uint32_t)(symbol >> go_rice) - thr;
      while (prefix < max_prefix && (int32_t)suffix > ((2 << prefix) - 2)) prefix++;
      const uint32_t suffix_len = prefix == max_prefix ? (uint32_t)(max_log2 - go_rice) : prefix + 1;
      rate += (int32_t)((thr + prefix + suffix_len + go_rice) << 15);
    } else {
      length = go_rice;
      symbol = symbol - (thr << go_rice);
      while (symbol >= (1 << length)) symbol -= (1 << (length++));
      rate += (thr + length + 1 - go_rice + length) << 15;
    }
    rate += (int32_t)BITS(c->par[t][ctx_par], (abs_level - 2) & 1);
    rate += (int32_t)BITS(c->gt1[t][ctx_gt1], 1);
    rate += (int32_t)BITS(c->gt2[t][ctx_gt2], 1);
  } else if (abs_level == 1) {
    rate += (int32_t)BITS(c->gt1[t][ctx_gt1], 0);
  } else if (abs_level == 2) {
    rate += (int32_t)BITS(c->par[t][ctx_par], 0);
    rate += (int32_t)BITS(c->gt1[t][ctx_gt1], 1);
    rate += (int32_t)BITS(c->gt2[t][ctx_gt2], 0);
  } else if (abs_level == 3) {
    rate += (int32_t)BITS(c->par[t][ctx_par], 1);
    rate += (int32_t)BITS(c->gt1[t][ctx_gt1], 1);
    rate += (int32_t)BITS(c->gt2[t][ctx_gt2], 0);
  } else {
    rate = 0;
  }
  return rate;
}

/* rdo.c:597-640 */
static uint32_t coded_level(const orc_rdoq_ctx *c, int t, double lambda, double *coded_cost, const double *coded_cost0, double *coded_cost_sig,
                            int32_t level_double, uint32_t max_abs_level, int ctx_sig, int ctx_gt1, int ctx_gt2, int ctx_par,
                            int go_rice, uint32_t reg_bins, int32_t q_bits, double error_scale, int last)
{
  double cur_cost_sig = 0;
  uint32_t best = 0;
  if (!last && max_abs_level < 3) {
    *coded_cost_sig = lambda * BITS(c->sig[t][ctx_sig], 0);
    *coded_cost = *coded_cost0 + *coded_cost_sig;
    if (max_abs_level == 0) return best;
  } else {
    *coded_cost = 1.7e+308;                        /* MAX_DOUBLE, global.h */
  }
  if (!last) cur_cost_sig = lambda * BITS(c->sig[t][ctx_sig], 1);
  const int32_t min_abs = max_abs_level > 1 ? (int32_t)max_abs_level - 1 : 1;
  for (int32_t a = (int32_t)max_abs_level; a >= min_abs; a--) {
    const double err = (double)(level_double - (a * (1 << q_bits)));
    double cur = err * err * error_scale + lambda * ic_rate(c, t, (uint32_t)a, ctx_gt1, ctx_gt2, ctx_par, go_rice, reg_bins, 1);
    cur += cur_cost_sig;
    if (cur < *coded_cost) { best = (uint32_t)a; *coded_cost = cur; *coded_cost_sig = cur_cost_sig; }
  }
  return best;
}

static void calc_last_bits(const orc_rdoq_ctx *c, int width, int height, int t, int32_t *lx, int32_t *ly)
{
  static const int prefix_ctx[8] = {0, 0, 0, 3, 6, 10, 15, 21};
  int32_t bits_x = 0, bits_y = 0, ctx;
  const int l2w = orc_log2i(width), l2h = orc_log2i(height);
  const int ox = t ? 0 : prefix_ctx[l2w], oy = t ? 0 : prefix_ctx[l2h];
  const int sx = t ? orc_clip3(0, 2, width >> 3) : ((l2w + 1) >> 2), sy = t ? orc_clip3(0, 2, height >> 3) : ((l2h + 1) >> 2);
  for (ctx = 0; ctx < group_idx(width - 1); ctx++) {
    const int o = ox + (ctx >> sx);
    lx[ctx] = bits_x + (int32_t)BITS(c->last_x[t][o], 0);
    bits_x += (int32_t)BITS(c->last_x[t][o], 1);
  }
  lx[ctx] = bits_x;
  for (ctx = 0; ctx < group_idx(height - 1); ctx++) {
    const int o = oy + (ctx >> sy);
    ly[ctx] = bits_y + (int32_t)BITS(c->last_y[t][o], 0);
    bits_y += (int32_t)BITS(c->last_y[t][o], 1);
  }
  ly[ctx] = bits_y;
}

static double rate_last(double lambda, uint32_t pos_x, uint32_t pos_y, const int32_t *lx, const int32_t *ly)
{
  const uint32_t cx = (uint32_t)group_idx((int)pos_x), cy = (uint32_t)group_idx((int)pos_y);
  double cost = lx[cx] + ly[cy];
  if (cx > 3) cost += 32768 * ((cx - 2) >> 1);
  if (cy > 3) cost += 32768 * ((cy - 2) >> 1);
  return lambda * cost;
}

static uint32_t sig_coeff_group_ctx(const uint32_t *flags, uint32_t px, uint32_t py, int w, int h)
{
  uint32_t right = 0, lower = 0;
  const uint32_t pos = py * (uint32_t)w + px;
  if (px + 1 < (uint32_t)w) right = flags[pos + 1];
  if (py + 1 < (uint32_t)h) lower = flags[pos + (uint32_t)w];
  return right || lower;
}

/*
 * uvg_rdoq.  color 0/1/2; block_type: cu_type_t (1 intra); cbf_u: cbf_is_set(cbf, COLOR_U) (only read for color 2).
 * Returns the sum of the absolute levels kept (abs_sum, rdo.c:1836).
 */
/* what RDOQ records per position for sign-data hiding (struct sh_rates_t, rdo.c:214-223) */
typedef struct { int32_t inc[1024], dec[1024], sig_coeff_inc[1024], quant_delta[1024]; } orc_sh_rates;

/* uvg_rdoq_sign_hiding (rdo.c:700-845).  last_pos = best_last_idx_p1; need_sqrt_adjust: odd log2 size sum. */
static void rdoq_sign_hiding(int qp_scaled, const uint32_t *scan, const orc_sh_rates *sh, int32_t last_pos, const int16_t *coeffs,
                             int16_t *quant_coeffs, double lambda, int need_sqrt_adjust)
{
  static const int16_t inv_quant_scales[2][6] = {{40, 45, 51, 57, 64, 72}, {57, 64, 72, 80, 90, 102}};
  const int inv_quant = inv_quant_scales[need_sqrt_adjust][qp_scaled % 6];
  /* the reference forms this product in int (it wraps for qp_scaled / 6 >= 9) and divides in double */
  const int32_t prod = (int32_t)((uint32_t)(inv_quant * inv_quant) * (1u << (2 * (qp_scaled / 6))));
  const int64_t rd_factor = (int64_t)(prod / lambda / 16 / (1 << (2 * (ORC_BIT_DEPTH - 8))) + 0.5);
  const int last_cg = (last_pos - 1) >> 4;
  for (int32_t cg_scan = last_cg; cg_scan >= 0; cg_scan--) {
    const int32_t cg0 = cg_scan << 4;
    int32_t last_nz = -1, first_nz = 16;
    for (int32_t i = 15; i >= 0; --i) if (quant_coeffs[scan[i + cg0]]) { last_nz = i; break; }
    for (int32_t i = 0; i <= last_nz; i++) if (quant_coeffs[scan[i + cg0]]) { first_nz = i; break; }
    if (last_nz - first_nz < 4) continue;                                      /* SBH_THRESHOLD */
    const int32_t signbit = quant_coeffs[scan[cg0 + first_nz]] <= 0;
    unsigned sum = 0;
    for (int32_t i = first_nz; i <= last_nz; i++) sum += (unsigned)quant_coeffs[scan[i + cg0]];
    if (signbit == (int32_t)(sum & 1)) continue;
    int64_t best_cost = INT64_MAX;
    int best_pos = 0, best_change = 0;
    const int last_coeff_scan = cg_scan == last_cg ? last_nz : 15;
    for (int cs = last_coeff_scan; cs >= 0; --cs) {
      const int pos = (int)scan[cs + cg0];
      int64_t cost;
      int change;
      const int64_t qbits = rd_factor * sh->quant_delta[pos];
      const int a = abs((int)quant_coeffs[pos]);
      if (a != 0) {
        int64_t inc_bits = sh->inc[pos], dec_bits = sh->dec[pos];
        if (a == 1) dec_bits -= sh->sig_coeff_inc[pos];
        if (cg_scan == last_cg && last_nz == cs && a == 1) dec_bits -= 4 * 32768;
        inc_bits = -qbits + inc_bits;
        dec_bits = qbits + dec_bits;
        if (inc_bits < dec_bits) { change = 1; cost = inc_bits; }
        else {
          change = -1; cost = dec_bits;
          if (cs == first_nz && a == 1) cost = INT64_MAX;
        }
      } else {
        const int bits = 32768 + sh->inc[pos] + sh->sig_coeff_inc[pos];
        cost = -llabs(qbits) + bits;
        change = 1;
        if (cs < first_nz && ((coeffs[pos] >= 0) ? 0 : 1) != signbit) cost = INT64_MAX;
      }
      if (cost < best_cost) { best_cost = cost; best_pos = pos; best_change = change; }
    }
    if (quant_coeffs[best_pos] == 32767 || quant_coeffs[best_pos] == -32768) best_change = -1;
    if (coeffs[best_pos] >= 0) quant_coeffs[best_pos] = (int16_t)(quant_coeffs[best_pos] + best_change);
    else quant_coeffs[best_pos] = (int16_t)(quant_coeffs[best_pos] - best_change);
  }
}

/* signhide: cfg.signhide_enable (rdo.c:1660-1687 record the rates, :1865-1867 apply the hiding) */
ORC_EXPORT int ORC_FN(rdoq_sh)(const int16_t *coef, int16_t *dest_coeff, int width, int height, int color, int block_type, int cbf_u,
                               int lfnst_idx, int mts_idx, int qp_scaled, double lambda, const void *ctx_snapshot, int signhide)
{
  const orc_rdoq_ctx *c = (const orc_rdoq_ctx *)ctx_snapshot;
  static orc_sh_rates sh;
#pragma omp threadprivate(sh)
  if (signhide) memset(&sh, 0, sizeof sh);
  const int t = color ? 1 : 0;
  const uint32_t l2w = (uint32_t)orc_log2i(width), l2h = (uint32_t)orc_log2i(height);
  const int sqrt2 = ((l2w + l2h) % 2 == 1);
  const int32_t transform_shift = 15 - ORC_BIT_DEPTH - (int32_t)((l2w + l2h) >> 1);
  uint16_t go_rice_param = 0;
  uint32_t reg_bins = (uint32_t)(width * height * 28) >> 4;
  const int32_t q_bits = 14 + qp_scaled / 6 + transform_shift - sqrt2;
  static const int16_t quant_scales[2][6] = {{26214, 23302, 20560, 18396, 16384, 14564}, {18396, 16384, 14564, 13107, 11651, 10280}};
  const int32_t q = quant_scales[sqrt2][qp_scaled % 6];
  double block_uncoded_cost = 0;
  double cost_coeff[32 * 32], cost_sig[32 * 32], cost_coeff0[32 * 32];
  memset(dest_coeff, 0, sizeof(int16_t) * (size_t)(width * height));
  const uint32_t cg_width = (uint32_t)(width < 32 ? width : 32) >> 2, cg_height = (uint32_t)(height < 32 ? height : 32) >> 2;
  uint32_t scan[1024], scan_cg[64];
  ORC_FN(rdoq_scans)(width, height, scan, scan_cg);
  const uint32_t cg_size = 16;
  const uint32_t num_blk_side = (uint32_t)(width >> 2) > 1 ? (uint32_t)(width >> 2) : 1;
  double cost_coeffgroup_sig[64];
  uint32_t sig_coeffgroup_flag[64];
  uint16_t ctx_set = 0;
  double base_cost = 0;
  int32_t temp_diag = -1, temp_sum = -1;
  int32_t cg_last_scanpos = -1, last_scanpos = -1;
  const uint32_t cg_num = lfnst_idx > 0 ? 1 : (uint32_t)(width * height) >> 4;
  const double d_trans_shift = (double)transform_shift + (sqrt2 ? -0.5 : 0.0);
  double scale = 32768;
  scale = scale * pow(2.0, -2.0 * d_trans_shift);
  const double error_scale = scale / q / q;
  for (uint32_t i = 0; i < cg_num; ++i) sig_coeffgroup_flag[i] = 0;

  struct { double coded_level_and_dist, uncoded_dist, sig_cost, sig_cost_0; int32_t nnz_before_pos0; } rd;
  const int max_lfnst_pos = ((height == 4 && width == 4) || (height == 8 && width == 8)) ? 7 : 15;
  int32_t cg_scanpos;
  const uint32_t max_group = lfnst_idx > 0 ? (uint32_t)max_lfnst_pos : cg_size - 1;
  for (cg_scanpos = (int32_t)cg_num - 1; cg_scanpos >= 0; cg_scanpos--) {
    const uint32_t cg_blkpos = scan_cg[cg_scanpos];
    const uint32_t cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
    if (mts_idx != 0 && (cg_pos_y >= 4 || cg_pos_x >= 4)) continue;
    for (int32_t sp = (int32_t)max_group; sp >= 0; sp--) {
      const int32_t scanpos = cg_scanpos * (int32_t)cg_size + sp;
      const uint32_t blkpos = scan[scanpos];
      int32_t level_double = coef[blkpos];
      { const int64_t prod = (int64_t)abs(level_double) * q; const int32_t cap = 0x7fffffff - (1 << (q_bits - 1));
        level_double = (int32_t)(prod < cap ? prod : cap); }
      const uint32_t max_abs_level = (uint32_t)(level_double + (1 << (q_bits - 1))) >> q_bits;
      const double err = (double)level_double;
      cost_coeff0[scanpos] = err * err * error_scale;
      dest_coeff[blkpos] = (int16_t)max_abs_level;
      if (max_abs_level > 0) { last_scanpos = scanpos; cg_last_scanpos = cg_scanpos; break; }
      block_uncoded_cost += cost_coeff0[scanpos];
      base_cost += cost_coeff0[scanpos];
    }
    if (last_scanpos != -1) break;
  }
  if (last_scanpos == -1) return 0;
  for (; cg_scanpos >= 0; cg_scanpos--) cost_coeffgroup_sig[cg_scanpos] = 0;

  int32_t last_x_bits[32], last_y_bits[32];
  for (int32_t cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const uint32_t cg_blkpos = scan_cg[cgs];
    const uint32_t cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
    memset(&rd, 0, sizeof rd);
    if (mts_idx != 0 && (cg_pos_y >= 4 || cg_pos_x >= 4)) continue;
    for (int32_t sp = (int32_t)max_group; sp >= 0; sp--) {
      const int32_t scanpos = cgs * (int32_t)cg_size + sp;
      if (scanpos > last_scanpos) continue;
      const uint32_t blkpos = scan[scanpos];
      int32_t level_double = coef[blkpos];
      { const int64_t prod = (int64_t)abs(level_double) * q; const int32_t cap = 0x7fffffff - (1 << (q_bits - 1));
        level_double = (int32_t)(prod < cap ? prod : cap); }
      const uint32_t max_abs_level = (uint32_t)(level_double + (1 << (q_bits - 1))) >> q_bits;
      dest_coeff[blkpos] = (int16_t)max_abs_level;
      const double err = (double)level_double;
      cost_coeff0[scanpos] = err * err * error_scale;
      block_uncoded_cost += cost_coeff0[scanpos];
      {
        const uint32_t pos_y = blkpos >> l2w, pos_x = blkpos - (pos_y << l2w);
        int32_t level;
        uint16_t ctx_sig = 0;
        if (scanpos != last_scanpos)
          ctx_sig = (uint16_t)sig_ctx_idx_abs(dest_coeff, pos_x, pos_y, (uint32_t)width, (uint32_t)height, color, &temp_diag, &temp_sum, mts_idx);
        if (temp_diag != -1)
          ctx_set = (uint16_t)(((temp_sum < 4 ? temp_sum : 4) + 1) +
                               (!temp_diag ? ((color == 0) ? 15 : 5) : (color == 0) ? (temp_diag < 3 ? 10 : (temp_diag < 10 ? 5 : 0)) : 0));
        else ctx_set = 0;
        if (reg_bins < 4) go_rice_param = (uint16_t)go_rice_par(template_abs_sum(dest_coeff, 0, pos_x, pos_y, (uint32_t)width, (uint32_t)height, mts_idx));
        if (scanpos == last_scanpos)
          level = (int32_t)coded_level(c, t, lambda, &cost_coeff[scanpos], &cost_coeff0[scanpos], &cost_sig[scanpos], level_double, max_abs_level,
                                       0, ctx_set, ctx_set, ctx_set, go_rice_param, reg_bins, q_bits, error_scale, 1);
        else
          level = (int32_t)coded_level(c, t, lambda, &cost_coeff[scanpos], &cost_coeff0[scanpos], &cost_sig[scanpos], level_double, max_abs_level,
                                       ctx_sig, ctx_set, ctx_set, ctx_set, go_rice_param, reg_bins, q_bits, error_scale, 0);
        if (signhide) {
          if (scanpos != last_scanpos)
            sh.sig_coeff_inc[blkpos] = reg_bins < 4 ? 0 : (int32_t)BITS(c->sig[t][ctx_sig], 1) - (int32_t)BITS(c->sig[t][ctx_sig], 0);
          sh.quant_delta[blkpos] = (level_double - level * (1 << q_bits)) >> (q_bits - 8);
          if (level > 0) {
            const int32_t now = ic_rate(c, t, (uint32_t)level, ctx_set, ctx_set, ctx_set, go_rice_param, reg_bins, 0);
            sh.inc[blkpos] = ic_rate(c, t, (uint32_t)level + 1, ctx_set, ctx_set, ctx_set, go_rice_param, reg_bins, 0) - now;
            sh.dec[blkpos] = ic_rate(c, t, (uint32_t)level - 1, ctx_set, ctx_set, ctx_set, go_rice_param, reg_bins, 0) - now;
          } else if (reg_bins < 4) {
            const int32_t now = ic_rate(c, t, 0, ctx_set, ctx_set, ctx_set, go_rice_param, reg_bins, 0);
            sh.inc[blkpos] = ic_rate(c, t, 1, ctx_set, ctx_set, ctx_set, go_rice_param, reg_bins, 0) - now;
          } else {
            sh.inc[blkpos] = (int32_t)BITS(c->gt1[t][ctx_set], 0);
          }
        }
        dest_coeff[blkpos] = (int16_t)level;
        base_cost += cost_coeff[scanpos];
        if ((scanpos % 16 == 0) && scanpos > 0) go_rice_param = 0;
        else if (reg_bins >= 4) {
          reg_bins -= (uint32_t)((level < 2 ? level : 3) + (scanpos != last_scanpos));
          go_rice_param = (uint16_t)go_rice_par(template_abs_sum(coef, 4, pos_x, pos_y, (uint32_t)width, (uint32_t)height, mts_idx));   /* sic: coef, not dest_coeff (rdo.c:1697) */
        }
      }
      rd.sig_cost += cost_sig[scanpos];
      if (sp == 0) rd.sig_cost_0 = cost_sig[scanpos];
      if (dest_coeff[blkpos]) {
        sig_coeffgroup_flag[cg_blkpos] = 1;
        rd.coded_level_and_dist += cost_coeff[scanpos] - cost_sig[scanpos];
        rd.uncoded_dist += cost_coeff0[scanpos];
        if (sp != 0) rd.nnz_before_pos0++;
      }
    }
    if (cgs) {
      if (sig_coeffgroup_flag[cg_blkpos] == 0) {
        const uint32_t cs = sig_coeff_group_ctx(sig_coeffgroup_flag, cg_pos_x, cg_pos_y, (int)cg_width, (int)cg_height);
        cost_coeffgroup_sig[cgs] = lambda * BITS(c->sig_group[t][cs], 0);
        base_cost += cost_coeffgroup_sig[cgs] - rd.sig_cost;
      } else if (cgs < cg_last_scanpos) {
        if (rd.nnz_before_pos0 == 0) { base_cost -= rd.sig_cost_0; rd.sig_cost -= rd.sig_cost_0; }
        double cost_zero_cg = base_cost;
        const uint32_t cs = sig_coeff_group_ctx(sig_coeffgroup_flag, cg_pos_x, cg_pos_y, (int)cg_width, (int)cg_height);
        cost_coeffgroup_sig[cgs] = lambda * BITS(c->sig_group[t][cs], 1);
        base_cost += cost_coeffgroup_sig[cgs];
        cost_zero_cg += lambda * BITS(c->sig_group[t][cs], 0);
        cost_zero_cg += rd.uncoded_dist;
        cost_zero_cg -= rd.coded_level_and_dist;
        cost_zero_cg -= rd.sig_cost;
        if (cost_zero_cg < base_cost) {
          sig_coeffgroup_flag[cg_blkpos] = 0;
          base_cost = cost_zero_cg;
          cost_coeffgroup_sig[cgs] = lambda * BITS(c->sig_group[t][cs], 0);
          for (int32_t sp = (int32_t)max_group; sp >= 0; sp--) {
            const int32_t scanpos = cgs * (int32_t)cg_size + sp;
            const uint32_t blkpos = scan[scanpos];
            if (dest_coeff[blkpos]) { dest_coeff[blkpos] = 0; cost_coeff[scanpos] = cost_coeff0[scanpos]; cost_sig[scanpos] = 0; }
          }
        }
      }
    } else {
      sig_coeffgroup_flag[cg_blkpos] = 1;
    }
  }

  double best_cost = 0;
  int found_last = 0;
  int32_t best_last_idx_p1 = 0;
  if (block_type != 1 && !color) {
    best_cost = block_uncoded_cost + lambda * BITS(c->root_cbf, 0);
    base_cost += lambda * BITS(c->root_cbf, 1);
  } else {
    const uint8_t *cbf_model = color == 0 ? c->cbf_luma : color == 1 ? c->cbf_cb : c->cbf_cr;
    const int ctx_cbf = color != 2 ? 0 : (cbf_u ? 1 : 0);
    best_cost = block_uncoded_cost + lambda * BITS(cbf_model[ctx_cbf], 0);
    base_cost += lambda * BITS(cbf_model[ctx_cbf], 1);
  }
  calc_last_bits(c, width, height, t, last_x_bits, last_y_bits);
  for (int32_t cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const uint32_t cg_blkpos = scan_cg[cgs];
    base_cost -= cost_coeffgroup_sig[cgs];
    if (sig_coeffgroup_flag[cg_blkpos]) {
      for (int32_t sp = (int32_t)max_group; sp >= 0; sp--) {
        const int32_t scanpos = cgs * (int32_t)cg_size + sp;
        if (scanpos > last_scanpos) continue;
        const uint32_t blkpos = scan[scanpos];
        if (dest_coeff[blkpos]) {
          const uint32_t pos_y = blkpos >> l2w, pos_x = blkpos - (pos_y << l2w);
          const double cost_last = rate_last(lambda, pos_x, pos_y, last_x_bits, last_y_bits);
          const double total = base_cost + cost_last - cost_sig[scanpos];
          if (total < best_cost) { best_last_idx_p1 = scanpos + 1; best_cost = total; }
          if (dest_coeff[blkpos] > 1) { found_last = 1; break; }
          base_cost -= cost_coeff[scanpos];
          base_cost += cost_coeff0[scanpos];
        } else {
          base_cost -= cost_sig[scanpos];
        }
      }
      if (found_last) break;
    }
  }
  uint32_t abs_sum = 0;
  if (!mts_idx || (width < 32 && height < 32)) {
    for (int32_t scanpos = 0; scanpos < best_last_idx_p1; scanpos++) {
      const int32_t b = (int32_t)scan[scanpos], level = dest_coeff[b];
      abs_sum += (uint32_t)level;
      dest_coeff[b] = (int16_t)((coef[b] < 0) ? -level : level);
    }
  } else {
    for (int32_t scanpos = 0; scanpos < best_last_idx_p1; scanpos++) {
      const int32_t b = (int32_t)scan[scanpos];
      const int32_t bx = b & (width - 1), by = b >> l2w;
      const int32_t level = bx >= 16 || by >= 16 ? 0 : dest_coeff[b];
      abs_sum += (uint32_t)level;
      dest_coeff[b] = (int16_t)((level != 0 && coef[b] < 0) ? -level : level);
    }
  }
  for (int32_t scanpos = best_last_idx_p1; scanpos <= last_scanpos; scanpos++) dest_coeff[scan[scanpos]] = 0;
  if (signhide && abs_sum >= 2) rdoq_sign_hiding(qp_scaled, scan, &sh, best_last_idx_p1, coef, dest_coeff, lambda, sqrt2);
  return (int)abs_sum;
}

ORC_EXPORT int ORC_FN(rdoq)(const int16_t *coef, int16_t *dest_coeff, int width, int height, int color, int block_type, int cbf_u,
                            int lfnst_idx, int mts_idx, int qp_scaled, double lambda, const void *ctx_snapshot)
{
  return ORC_FN(rdoq_sh)(coef, dest_coeff, width, height, color, block_type, cbf_u, lfnst_idx, mts_idx, qp_scaled, lambda, ctx_snapshot, 0);
}
