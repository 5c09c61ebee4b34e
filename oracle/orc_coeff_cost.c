/*
 * oracle/orc_coeff_cost.c -- restatement of the CABAC bit-cost estimate of a block's coefficients:
 *   uvg_get_coeff_cost -> get_coeff_cabac_cost      src/rdo.c:297-356, 393-454  (the branch that is not uvg_fast_coeff_cost)
 *   uvg_encode_coeff_nxn_generic in count mode      src/strategies/generic/encode_coding_tree-generic.c:53-323
 *   uvg_encode_last_significant_xy                  src/encode_coding_tree.c:415-470
 *   uvg_cabac_write_coeff_remain (bit count)        src/cabac.c:318-354
 *   CABAC_FBITS_UPDATE / CTX_UPDATE / CTX_STATE     src/cabac.h:166-196
 *   uvg_context_get_sig_ctx_idx_abs, uvg_abs_sum, uvg_context_get_sig_coeff_group   src/context.c:647-727, 846-877
 * The reference takes a COPY of the search CABAC (rdo.c:322-328: only_count = 1, update = 1), runs the real coefficient
 * coder on it and sums, per context-coded bin, the fractional bits of the model's current state
 * (uvg_f_entropy_bits[(CTX_STATE << 1) ^ bin], a float table = uvg_entropy_bits / 2^15) while the models adapt bin by bin
 * (CTX_UPDATE: the VVC two-rate estimator); bypass bins count 1 bit each.  The arithmetic-coder registers do not enter
 * the count.  Here the copy is the caller's snapshot of the 244 models (same index space as the RDOQ snapshot, orc_rdoq.c)
 * with their two 16-bit states and the rate byte.
 * Configuration subset: dependent quantisation off (quant_state stays 0), sign-data hiding off, diagonal scan, no
 * transform skip (uvg_encode_ts_residual is a different coder).
 * Also returned: the LFNST / MTS signalling constraints the coder records in cur_cu as a side effect (:113-121, :303-317).
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"
#include <math.h>
#include <string.h>

enum { CC_SIGGRP = 0, CC_SIG = 4, CC_PAR = 28, CC_GT1 = 70, CC_GT2 = 112, CC_LASTX = 154, CC_LASTY = 194, CC_N = 244 };

typedef struct orc_cabac_models {
  uint16_t state0[CC_N], state1[CC_N];   /* cabac_ctx_t.state[0], state[1] */
  uint8_t rate[CC_N];                    /* cabac_ctx_t.rate */
} orc_cabac_models;

void ORC_FN(rdoq_scans)(int width, int height, uint32_t *scan, uint32_t *scan_cg);

static float f_entropy_bits(int state, int val)      /* rdo.c:143: uvg_entropy_bits scaled by 2^-15, exactly representable */
{
  const double p1 = (2 * state + 1) / 512.0;
  return (float)(floor(-log2(val ? p1 : 1.0 - p1) * 32768.0 + 0.5) / 32768.0);
}

/* the table as an array, for the pinning test against uvg_f_entropy_bits */
ORC_EXPORT void ORC_FN(f_entropy_table)(float *out512)
{
  for (int i = 0; i < 512; ++i) out512[i] = f_entropy_bits(i >> 1, i & 1);
}

/* CABAC_FBITS_UPDATE with only_count = 1, update = 1 */
/* Count mode of the arithmetic coder (uvg_cabac_encode_bin, cabac.c:76-109): when switched on, every context-coded bin also
 * moves the coder's range and counts the renormalisation shifts it causes -- the bits the real coder consumes for it.  Bypass
 * bins cost one bit each whatever the range is, so a walk's exact size is shifts + (its bit estimate - regular_fbits).
 * Shared with orc_search.c (the bins of the coding tree outside the coefficients). */
orc_cabac_sim ORC_FN(cabac_sim) = {0, 510, 0, 0.0, 0, 0xff, 23, 0, NULL, 0, 0};
static void sim_put_byte(orc_cabac_sim *c, uint32_t byte)
{
  if (c->out_len == c->out_cap) { c->out_cap = c->out_cap ? 2 * c->out_cap : 4096; c->out = (uint8_t *)realloc(c->out, c->out_cap); }
  c->out[c->out_len++] = (uint8_t)byte;
}
static void sim_write(orc_cabac_sim *c)                                     /* uvg_cabac_write (cabac.c:114-144) */
{
  const uint32_t lead_byte = c->low >> (24 - c->bits_left);
  c->bits_left += 8;
  c->low &= 0xffffffffu >> c->bits_left;
  if (lead_byte == 0xff) { c->num_buffered_bytes++; return; }
  if (c->num_buffered_bytes > 0) {
    const uint32_t carry = lead_byte >> 8;
    uint32_t byte = c->buffered_byte + carry;
    c->buffered_byte = lead_byte & 0xff;
    sim_put_byte(c, byte);
    byte = (0xff + carry) & 0xff;
    while (c->num_buffered_bytes > 1) { sim_put_byte(c, byte); c->num_buffered_bytes--; }
  } else {
    c->num_buffered_bytes = 1;
    c->buffered_byte = lead_byte;
  }
}
void ORC_FN(cabac_sim_bin)(int state, int bin)                            /* uvg_cabac_encode_bin (cabac.c:76-109) */
{
  static const uint8_t renorm[32] = {6, 5, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};   /* uvg_g_auc_renorm_table */
  orc_cabac_sim *c = &ORC_FN(cabac_sim);
  const uint32_t q = (state & 0x80) ? (uint32_t)(state ^ 0xff) : (uint32_t)state;
  const uint32_t lps = (uint8_t)((((q >> 2) * (c->range >> 5)) >> 1) + 4);        /* CTX_LPS (cabac.h:182) */
  c->range -= lps;
  if ((bin ? 1 : 0) != (state >> 7)) {                                               /* CTX_MPS */
    const int nb = renorm[lps >> 3];
    c->low = (c->low + c->range) << nb;
    c->range = lps << nb;
    c->shifts += (uint64_t)nb;
    c->bits_left -= nb;
    if (c->on == 2 && c->bits_left < 12) sim_write(c);
  } else if (c->range < 256) {
    c->low <<= 1;
    c->range <<= 1;
    c->shifts += 1;
    c->bits_left--;
    if (c->on == 2 && c->bits_left < 12) sim_write(c);
  }
}
void ORC_FN(cabac_sim_ep)(uint32_t bin)                                     /* uvg_cabac_encode_bin_ep (cabac.c:235-246) */
{
  orc_cabac_sim *c = &ORC_FN(cabac_sim);
  if (c->on != 2) return;
  c->low <<= 1;
  if (bin) c->low += c->range;
  c->bits_left--;
  if (c->bits_left < 12) sim_write(c);
}
void ORC_FN(cabac_sim_eps)(uint32_t bin_values, int num_bins)               /* uvg_cabac_encode_bins_ep / _aligned_bins_ep (cabac.c:249-311) */
{
  orc_cabac_sim *c = &ORC_FN(cabac_sim);
  if (c->on != 2) return;
  if (c->range == 256) {
    uint32_t rem = (uint32_t)num_bins;
    while (rem > 0) {
      const uint32_t n = rem < 8 ? rem : 8, mask = (1u << n) - 1;
      const uint32_t nb = (bin_values >> (rem - n)) & mask;
      c->low = (c->low << n) + (nb << 8);
      rem -= n;
      c->bits_left -= (int32_t)n;
      if (c->bits_left < 12) sim_write(c);
    }
    return;
  }
  while (num_bins > 8) {
    num_bins -= 8;
    const uint32_t pattern = bin_values >> num_bins;
    c->low <<= 8;
    c->low += c->range * pattern;
    bin_values -= pattern << num_bins;
    c->bits_left -= 8;
    if (c->bits_left < 12) sim_write(c);
  }
  c->low <<= num_bins;
  c->low += c->range * bin_values;
  c->bits_left -= num_bins;
  if (c->bits_left < 12) sim_write(c);
}
void ORC_FN(cabac_sim_start)(void)                                           /* uvg_cabac_start (cabac.c:63-72) */
{
  orc_cabac_sim *c = &ORC_FN(cabac_sim);
  c->low = 0; c->range = 510; c->bits_left = 23; c->num_buffered_bytes = 0; c->buffered_byte = 0xff; c->out_len = 0;
}
/* the end of a substream (encoderstate.c:925-938): end_of_sub_stream_one_bit = uvg_cabac_encode_bin_trm(1) (cabac.c:179-199),
 * uvg_cabac_finish (:149-174), a one bit and the zero bits up to the byte boundary */
void ORC_FN(cabac_sim_row_end)(void)
{
  orc_cabac_sim *c = &ORC_FN(cabac_sim);
  c->range -= 2;
  c->low += c->range;
  c->low <<= 7;
  c->range = 2 << 7;
  c->bits_left -= 7;
  if (c->bits_left < 12) sim_write(c);
  if (c->low >> (32 - c->bits_left)) {
    sim_put_byte(c, c->buffered_byte + 1);
    while (c->num_buffered_bytes > 1) { sim_put_byte(c, 0); c->num_buffered_bytes--; }
    c->low -= 1u << (32 - c->bits_left);
  } else {
    if (c->num_buffered_bytes > 0) sim_put_byte(c, c->buffered_byte);
    while (c->num_buffered_bytes > 1) { sim_put_byte(c, 0xff); c->num_buffered_bytes--; }
  }
  /* uvg_bitstream_put(stream, low >> 8, 24 - bits_left), then the one bit, then zeros to the boundary */
  uint32_t acc = 0;
  int nacc = 0;
  const int nb = 24 - c->bits_left;
  const uint32_t v = c->low >> 8;
  for (int i = nb - 1; i >= 0; --i) { acc = (acc << 1) | ((v >> i) & 1u); if (++nacc == 8) { sim_put_byte(c, acc); acc = 0; nacc = 0; } }
  acc = (acc << 1) | 1u; if (++nacc == 8) { sim_put_byte(c, acc); acc = 0; nacc = 0; }
  if (nacc) sim_put_byte(c, acc << (8 - nacc));
}

/* uvg_cabac_write_coeff_remain (cabac.c:318-354): the bins, not their count */
static void sim_coeff_remain(uint32_t remainder, uint32_t rice, unsigned cutoff)
{
  if (ORC_FN(cabac_sim).on != 2) return;
  const unsigned threshold = cutoff << rice;
  if (remainder < threshold) {
    const uint32_t length = (remainder >> rice) + 1;
    ORC_FN(cabac_sim_eps)((1u << length) - 2, (int)length);
    ORC_FN(cabac_sim_eps)(remainder & ((1u << rice) - 1), (int)rice);
  } else {
    const unsigned max_prefix = 32 - cutoff - 15;
    unsigned prefix_length = 0, suffix_length;
    const unsigned code_value = (remainder >> rice) - cutoff;
    if ((int32_t)code_value >= ((1 << max_prefix) - 1)) { prefix_length = max_prefix; suffix_length = 15; }
    else { while ((int32_t)code_value > ((2 << prefix_length) - 2)) prefix_length++; suffix_length = prefix_length + rice + 1; }
    const unsigned total_prefix = prefix_length + cutoff;
    const unsigned prefix = (1u << total_prefix) - 1;
    const unsigned suffix = ((code_value - ((1u << prefix_length) - 1)) << rice) | (remainder & ((1u << rice) - 1));
    ORC_FN(cabac_sim_eps)(prefix, (int)total_prefix);
    ORC_FN(cabac_sim_eps)(suffix, (int)suffix_length);
  }
}

static void code_bin(orc_cabac_models *m, int ctx, int bin, double *bits)
{
  const int st = (m->state0[ctx] + m->state1[ctx]) >> 8;                         /* CTX_STATE */
  *bits += f_entropy_bits(st, bin);
  if (ORC_FN(cabac_sim).on) { ORC_FN(cabac_sim).regular_fbits += f_entropy_bits(st, bin); ORC_FN(cabac_sim_bin)(st, bin); }
  const int rate0 = m->rate[ctx] >> 4, rate1 = m->rate[ctx] & 15;                /* CTX_UPDATE */
  const unsigned mask0 = (~(~0u << 10)) << 5, mask1 = (~(~0u << 14)) << 1;       /* CTX_MASK_0, CTX_MASK_1 */
  m->state0[ctx] = (uint16_t)(m->state0[ctx] - ((m->state0[ctx] >> rate0) & mask0));
  m->state1[ctx] = (uint16_t)(m->state1[ctx] - ((m->state1[ctx] >> rate1) & mask1));
  if (bin) {
    m->state0[ctx] = (uint16_t)(m->state0[ctx] + ((0x7fffu >> rate0) & mask0));
    m->state1[ctx] = (uint16_t)(m->state1[ctx] + ((0x7fffu >> rate1) & mask1));
  }
}

static int group_idx_cc(int pos)
{
  if (pos < 4) return pos;
  int l = 0;
  while ((pos >> (l + 1)) != 0) ++l;
  return 2 * l + ((pos >> (l - 1)) & 1);
}

static int coeff_remain_bits(uint32_t remainder, uint32_t rice, unsigned cutoff)   /* cabac.c:318-354 */
{
  const unsigned threshold = cutoff << rice;
  if (remainder < threshold) return (int)((remainder >> rice) + 1 + rice);
  const unsigned max_prefix = 32 - cutoff - 15;
  unsigned prefix = 0, suffix_len;
  const unsigned code_value = (remainder >> rice) - cutoff;
  if ((int32_t)code_value >= ((1 << max_prefix) - 1)) {
    prefix = max_prefix;
    suffix_len = 15;
  } else {
    while ((int32_t)code_value > ((2 << prefix) - 2)) prefix++;
    suffix_len = prefix + rice + 1;
  }
  return (int)(prefix + cutoff + suffix_len);
}

static int sig_ctx_abs(const int16_t *coeff, uint32_t px, uint32_t py, uint32_t w, uint32_t h, int color, int *diag_out, int *sum_out)
{
  const int16_t *d = coeff + px + py * w;
  const int diag = (int)(px + py);
  int num_pos = 0, sum_abs = 0;
#define UPD(x) { const int a = abs((int)(x)); sum_abs += (4 + (a & 1)) < a ? (4 + (a & 1)) : a; num_pos += a ? 1 : 0; }
  if (px < w - 1) {
    UPD(d[1]);
    if (px < w - 2) UPD(d[2]);
    if (py < h - 1) UPD(d[w + 1]);
  }
  if (py < h - 1) {
    UPD(d[w]);
    if (py < h - 2) UPD(d[w << 1]);
  }
#undef UPD
  int ofs = (((sum_abs + 1) >> 1) < 3 ? ((sum_abs + 1) >> 1) : 3) + (diag < 2 ? 4 : 0);
  if (color == 0) ofs += diag < 5 ? 4 : 0;
  *diag_out = diag;
  *sum_out = sum_abs - num_pos;
  return ofs;
}

static int abs_sum(const int16_t *coeff, uint32_t px, uint32_t py, uint32_t w, uint32_t h, int baselevel)
{
  const int16_t *d = coeff + px + py * w;
  int sum = 0;
  if (px < w - 1) {
    sum += abs((int)d[1]);
    if (px < w - 2) sum += abs((int)d[2]);
    if (py < h - 1) sum += abs((int)d[w + 1]);
  }
  if (py < h - 1) {
    sum += abs((int)d[w]);
    if (py < h - 2) sum += abs((int)d[w << 1]);
  }
  sum -= 5 * baselevel;
  return sum > 31 ? 31 : (sum < 0 ? 0 : sum);
}

static int go_rice(int s) { return (s >= 7) + (s >= 14) + (s >= 28); }               /* g_go_rice_pars, tables.h:44 */

/* flags_out: bit 0 scan_pos_last > max_lfnst_pos (violates_lfnst_constrained_*), bit 1 scan_pos_last >= 1
 * (lfnst_last_scan_pos), bit 2 mts_last_scan_pos, bit 3 violates_mts_coeff_constraint (luma only for 2, 3).
 * models_out (may be NULL): the adapted copy, as cabac_copy stands at the end. */
ORC_EXPORT double ORC_FN(coeff_cost)(const int16_t *coeff, int width, int height, int color, const orc_cabac_models *models_in,
                                    uint32_t *flags_out, orc_cabac_models *models_out)
{
  uint32_t flags = 0;
  if (flags_out) *flags_out = 0;
  int found = 0;
  for (int i = 0; i < width * height; ++i) if (coeff[i]) { found = 1; break; }
  if (!found) { if (models_out) *models_out = *models_in; return 0.0; }             /* rdo.c:312-320 */
  orc_cabac_models m = *models_in;
  static uint32_t scan[32 * 32], scan_cg[64];
#pragma omp threadprivate(scan, scan_cg)
  ORC_FN(rdoq_scans)(width, height, scan, scan_cg);
  const int t = color ? 1 : 0;
  uint32_t sig_cg[64];
  memset(sig_cg, 0, sizeof sig_cg);
  int scan_pos_last = -1;
  for (int i = 0; i < width * height; ++i)
    if (coeff[scan[i]]) { scan_pos_last = i; sig_cg[scan_cg[i >> 4]] = 1; }
  const int scan_cg_last = scan_pos_last >> 4;
  const int pos_last = (int)scan[scan_pos_last];
  const int last_y = pos_last / width, last_x = pos_last - last_y * width;
  {
    const unsigned max_lfnst_pos = ((height == 4 && width == 4) || (height == 8 && width == 8)) ? 7 : 15;
    if ((unsigned)scan_pos_last > max_lfnst_pos) flags |= 1;
    if (scan_pos_last >= 1) flags |= 2;
  }
  double bits_out = 0;
  /* ---- uvg_encode_last_significant_xy ---- */
  {
    static const int prefix_ctx[8] = {0, 0, 0, 3, 6, 10, 15, 21};
    int ix = 0, iy = 0;
    while ((width >> (ix + 1)) != 0) ++ix;
    while ((height >> (iy + 1)) != 0) ++iy;
    const int off_x = t ? 0 : prefix_ctx[ix], off_y = t ? 0 : prefix_ctx[iy];
    const int sh_x = t ? ((width >> 3) < 0 ? 0 : ((width >> 3) > 2 ? 2 : (width >> 3))) : (ix + 1) >> 2;
    const int sh_y = t ? ((height >> 3) < 0 ? 0 : ((height >> 3) > 2 ? 2 : (height >> 3))) : (iy + 1) >> 2;
    const int gx = group_idx_cc(last_x), gy = group_idx_cc(last_y);
    double bits = 0;
    int k = 0;
    for (; k < gx; k++) code_bin(&m, CC_LASTX + 20 * t + off_x + (k >> sh_x), 1, &bits);
    if (gx < group_idx_cc((width < 32 ? width : 32) - 1)) code_bin(&m, CC_LASTX + 20 * t + off_x + (k >> sh_x), 0, &bits);
    k = 0;
    for (; k < gy; k++) code_bin(&m, CC_LASTY + 20 * t + off_y + (k >> sh_y), 1, &bits);
    if (gy < group_idx_cc((height < 32 ? height : 32) - 1)) code_bin(&m, CC_LASTY + 20 * t + off_y + (k >> sh_y), 0, &bits);
    static const int min_in_group[14] = {0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96};      /* g_min_in_group */
    if (gx > 3) { bits += (gx - 2) / 2; ORC_FN(cabac_sim_eps)((uint32_t)(last_x - min_in_group[gx]), (gx - 2) / 2); }
    if (gy > 3) { bits += (gy - 2) / 2; ORC_FN(cabac_sim_eps)((uint32_t)(last_y - min_in_group[gy]), (gy - 2) / 2); }
    bits_out += bits;
  }
  /* ---- the coefficient groups ---- */
  double bits = 0;
  uint8_t ctx_offset[16];
  int temp_diag = -1, temp_sum = -1;
  int32_t reg_bins = (width * height * 28) >> 4;
  const uint32_t cg_width = (uint32_t)(width < 32 ? width : 32) >> 2, cg_height = (uint32_t)(height < 32 ? height : 32) >> 2;
  for (int i = scan_cg_last; i >= 0; i--) {
    const int cg_blk_pos = (int)scan_cg[i];
    const int cg_pos_y = cg_blk_pos / (int)cg_width, cg_pos_x = cg_blk_pos - cg_pos_y * (int)cg_width;
    if (i == scan_cg_last || i == 0) {
      sig_cg[cg_blk_pos] = 1;
    } else {
      uint32_t right = 0, lower = 0;
      if ((uint32_t)cg_pos_x + 1 < cg_width) right = sig_cg[cg_blk_pos + 1];
      if ((uint32_t)cg_pos_y + 1 < cg_height) lower = sig_cg[cg_blk_pos + (int)cg_width];
      code_bin(&m, CC_SIGGRP + 2 * t + ((right || lower) ? 1 : 0), sig_cg[cg_blk_pos] != 0, &bits);
    }
    if (sig_cg[cg_blk_pos]) {
      const int min_sub_pos = i << 4;
      const int first_sig_pos = (i == scan_cg_last) ? scan_pos_last : (min_sub_pos + 15);
      int next_sig_pos = first_sig_pos;
      const int infer_sig_pos = (next_sig_pos != scan_pos_last) ? ((i != 0) ? min_sub_pos : -1) : next_sig_pos;
      int num_non_zero = 0;
      uint32_t coeff_signs = 0;                /* one bit per non-zero level in coding order, first coded = most significant */
      /* first pass: context-coded flags while regular bins remain */
      for (next_sig_pos = first_sig_pos; next_sig_pos >= min_sub_pos && reg_bins >= 4; next_sig_pos--) {
        const uint32_t blk_pos = scan[next_sig_pos];
        const uint32_t pos_y = blk_pos / (uint32_t)width, pos_x = blk_pos - pos_y * (uint32_t)width;
        const int sig = coeff[blk_pos] != 0;
        if (num_non_zero || next_sig_pos != infer_sig_pos) {
          const int ctx_sig = sig_ctx_abs(coeff, pos_x, pos_y, (uint32_t)width, (uint32_t)height, color, &temp_diag, &temp_sum);
          code_bin(&m, CC_SIG + 12 * t + (t ? (ctx_sig < 7 ? ctx_sig : 7) : ctx_sig), sig, &bits);
          reg_bins--;
        } else if (next_sig_pos != scan_pos_last) {
          (void)sig_ctx_abs(coeff, pos_x, pos_y, (uint32_t)width, (uint32_t)height, color, &temp_diag, &temp_sum);
        }
        if (sig) {
          uint8_t *offset = &ctx_offset[next_sig_pos - min_sub_pos];
          num_non_zero++;
          coeff_signs = (coeff_signs << 1) | (coeff[blk_pos] < 0);
          *offset = 0;
          if (temp_diag != -1) {
            *offset = (uint8_t)((temp_sum < 4 ? temp_sum : 4) + 1);
            *offset = (uint8_t)(*offset + (!temp_diag ? (color == 0 ? 15 : 5) : color == 0 ? (temp_diag < 3 ? 10 : (temp_diag < 10 ? 5 : 0)) : 0));
          }
          int rem = abs((int)coeff[blk_pos]) - 1;
          const int gt1 = rem ? 1 : 0;
          code_bin(&m, CC_GT1 + 21 * t + *offset, gt1, &bits);
          reg_bins--;
          if (gt1) {
            rem -= 1;
            code_bin(&m, CC_PAR + 21 * t + *offset, rem & 1, &bits);
            rem >>= 1;
            reg_bins--;
            code_bin(&m, CC_GT2 + 21 * t + *offset, rem ? 1 : 0, &bits);
            reg_bins--;
          }
        }
      }
      /* second pass: Golomb-Rice remainders of the context-coded positions */
      for (int scan_pos = first_sig_pos; scan_pos > next_sig_pos; scan_pos--) {
        const uint32_t blk_pos = scan[scan_pos];
        const uint32_t pos_y = blk_pos / (uint32_t)width, pos_x = blk_pos - pos_y * (uint32_t)width;
        const int rice = go_rice(abs_sum(coeff, pos_x, pos_y, (uint32_t)width, (uint32_t)height, 4));
        const uint32_t a = (uint32_t)abs((int)coeff[blk_pos]);
        if (a >= 4) { bits += coeff_remain_bits((a - 4) >> 1, (uint32_t)rice, 5); sim_coeff_remain((a - 4) >> 1, (uint32_t)rice, 5); }
      }
      /* bypass-coded positions (regular bins spent) */
      for (int scan_pos = next_sig_pos; scan_pos >= min_sub_pos; scan_pos--) {
        const uint32_t blk_pos = scan[scan_pos];
        const uint32_t pos_y = blk_pos / (uint32_t)width, pos_x = blk_pos - pos_y * (uint32_t)width;
        const uint32_t a = (uint32_t)abs((int)coeff[blk_pos]);
        const int rice = go_rice(abs_sum(coeff, pos_x, pos_y, (uint32_t)width, (uint32_t)height, 0));
        const uint32_t pos0 = 1u << rice;                                       /* quant_state < 2 */
        const uint32_t remainder = a == 0 ? pos0 : (a <= pos0 ? a - 1 : a);
        bits += coeff_remain_bits(remainder, (uint32_t)rice, 5);
        sim_coeff_remain(remainder, (uint32_t)rice, 5);
        if (a) { num_non_zero++; coeff_signs = (coeff_signs << 1) | (coeff[blk_pos] < 0); }
      }
      if (color == 0 && first_sig_pos > 0) flags |= 4;                          /* mts_last_scan_pos (tr_idx != MTS_SKIP assumed) */
      bits += num_non_zero;                                                     /* coeff_signs, bypass */
      ORC_FN(cabac_sim_eps)(coeff_signs, num_non_zero);
    }
    if (color == 0 && (cg_pos_y > 3 || cg_pos_x > 3) && sig_cg[cg_blk_pos] != 0) flags |= 8;
  }
  bits_out += bits;
  if (flags_out) *flags_out = flags;
  if (models_out) *models_out = m;
  return bits_out;
}
