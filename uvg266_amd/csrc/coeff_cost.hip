// CABAC bit cost of coefficient blocks on gfx950: uvg_get_coeff_cost on its CABAC branch = get_coeff_cabac_cost
// (src/rdo.c:297-356) -> uvg_encode_coeff_nxn in count mode (src/strategies/generic/encode_coding_tree-generic.c:53-323,
// uvg_encode_last_significant_xy src/encode_coding_tree.c:415-470) for n blocks of one shape.
//
// The reference runs its real coefficient coder on a COPY of the search CABAC with only_count = 1, update = 1: every
// context-coded bin adds the fractional bits of its model's current state and then adapts the model (CTX_UPDATE, the VVC
// two-rate estimator), bypass bins add one bit.  Inside a block this is a strictly sequential recurrence over the bins
// (every bin's price depends on the adaptation by the bins before it); across blocks nothing is shared -- every block starts
// from the caller's snapshot -- so the mapping is ONE LANE = ONE BLOCK: its private copy of the models of its plane type lives
// in LDS as [model][lane] words (conflict-free), its coefficients in LDS with an odd word stride, scans and rates in
// shared tables.  The double-precision sum follows the reference's order (fractional bits are multiples of 2^-15: exact).
// Same configuration subset as the RDOQ kernel: no dependent quantisation, no sign-data hiding, diagonal scan, no transform
// skip (uvg_encode_ts_residual is a different coder).
#include "uvghip_common.h"
#include "vvc_rdoq_tables.h"

namespace {

enum : int { C_SIGGRP = 0, C_SIG = 4, C_PAR = 28, C_GT1 = 70, C_GT2 = 112, C_LASTX = 154, C_LASTY = 194, C_N = 244 };
// local index space of one plane type: sig group 2 | sig 12 | par 21 | gt1 21 | gt2 21 | last x 20 | last y 20
enum : int { L_SIGGRP = 0, L_SIG = 2, L_PAR = 14, L_GT1 = 35, L_GT2 = 56, L_LASTX = 77, L_LASTY = 97, L_N = 117 };

struct cc_params {
  int width, height, l2w, l2h, n, color;
  uvghip_cabac_models_t models;
};

__device__ __forceinline__ int cc_group_idx(int pos)
{
  if (pos < 4) return pos;
  const int l = 31 - __clz(pos);
  return 2 * l + ((pos >> (l - 1)) & 1);
}

__device__ __forceinline__ int cc_remain_bits(uint32_t remainder, uint32_t rice)          // cabac.c:318-354, cutoff 5
{
  const uint32_t cutoff = 5;
  if (remainder < (cutoff << rice)) return (int)((remainder >> rice) + 1 + rice);
  const uint32_t max_prefix = 32 - cutoff - 15;
  uint32_t prefix = 0, suffix_len;
  const uint32_t code_value = (remainder >> rice) - cutoff;
  if ((int)code_value >= ((1 << max_prefix) - 1)) { prefix = max_prefix; suffix_len = 15; }
  else { while ((int)code_value > ((2 << prefix) - 2)) prefix++; suffix_len = prefix + rice + 1; }
  return (int)(prefix + cutoff + suffix_len);
}

// LANES blocks per workgroup (one lane each); the workgroup is one wave
template <int LANES>
__global__ void __launch_bounds__(64)
coeff_cost_kernel(const cc_params Pk, const int16_t *__restrict__ coeff, double *__restrict__ bits_out,
                  uint8_t *__restrict__ flags_out)
{
  const cc_params *Pg = &Pk;                       // (kernel-argument segment: indexed loads, never copied to registers)
  extern __shared__ __attribute__((aligned(16))) unsigned char sDyn[];
  __shared__ uint32_t sModel[L_N][LANES];        // state0 | state1 << 16 of the lane's copy
  __shared__ uint8_t sRate[L_N];
  __shared__ uint16_t sScan[1024];
  __shared__ uint8_t sScanCg[64];
  const int tid = threadIdx.x;
  const int width = Pg->width, height = Pg->height, l2w = Pg->l2w, wh = width * height, n = Pg->n;
  const int color = Pg->color, t = color ? 1 : 0;
  const int l2cgw = l2w - 2, cgw = 1 << l2cgw, cgh = height >> 2, ncg = cgw * cgh;
  // ---- shared tables: coefficient-group scan (H.266 6.5.2), position scan in 4x4 groups, rates ----
  if (tid == 0) {
    int i = 0, x = 0, y = 0;
    while (i < ncg) {
      while (y >= 0) { if (x < cgw && y < cgh) sScanCg[i++] = (uint8_t)(y * cgw + x); y--; x++; }
      y = x; x = 0;
    }
  }
  auto glob = [&](int l) {                          // local model index -> index in uvghip_cabac_models_t
    if (l < L_SIG) return C_SIGGRP + 2 * t + l;
    if (l < L_PAR) return C_SIG + 12 * t + (l - L_SIG);
    if (l < L_GT1) return C_PAR + 21 * t + (l - L_PAR);
    if (l < L_GT2) return C_GT1 + 21 * t + (l - L_GT1);
    if (l < L_LASTX) return C_GT2 + 21 * t + (l - L_GT2);
    if (l < L_LASTY) return C_LASTX + 20 * t + (l - L_LASTX);
    return C_LASTY + 20 * t + (l - L_LASTY);
  };
  for (int l = tid; l < L_N; l += 64) sRate[l] = Pg->models.rate[glob(l)];
  for (int e = tid; e < L_N * LANES; e += 64) {
    const int l = e / LANES, g = glob(l);
    sModel[l][e - l * LANES] = (uint32_t)Pg->models.state0[g] | ((uint32_t)Pg->models.state1[g] << 16);
  }
  __syncthreads();
  constexpr unsigned long long kDiag4 = 0xFBE7AD369C258140ull;       // in-group diagonal order as (y * 4 + x) nibbles
  for (int e = tid; e < wh; e += 64) {
    const int g = sScanCg[e >> 4], k = (int)((kDiag4 >> (4 * (e & 15))) & 15);
    sScan[e] = (uint16_t)(((((g >> l2cgw) << 2) + (k >> 2)) << l2w) + ((g & (cgw - 1)) << 2) + (k & 3));
  }
  // ---- this workgroup's blocks into LDS (coalesced), odd word stride per block ----
  const int blk0 = blockIdx.x * LANES;
  const int here = min(LANES, n - blk0);
  const int stride = wh + 2;                                          // int16 elements: (wh / 2 + 1) words, odd
  int16_t *sCoef = reinterpret_cast<int16_t *>(sDyn);
  for (int e = tid; e < here * wh; e += 64) {
    const int b = e / wh, pos = e - b * wh;
    sCoef[b * stride + pos] = coeff[(size_t)blk0 * wh + e];
  }
  __syncthreads();
  if (tid >= here) return;
  const int16_t *C = sCoef + tid * stride;
  const int lane = tid;
  double total = 0.0;
  uint32_t flags = 0;
  // CABAC_FBITS_UPDATE with only_count = 1, update = 1 (cabac.h:166-196)
  auto code_bin = [&](int l, int bin, double &bits) {
    const uint32_t m = sModel[l][lane];
    uint32_t s0 = m & 0xffffu, s1 = m >> 16;
    const int st = (int)((s0 + s1) >> 8);
    bits += (double)kEntropyBits[2 * st + bin] * (1.0 / 32768.0);    // uvg_f_entropy_bits: exact (multiples of 2^-15)
    const int rate = sRate[l], rate0 = rate >> 4, rate1 = rate & 15;
    const uint32_t mask0 = 0x7fe0u, mask1 = 0x7ffeu;                  // CTX_MASK_0, CTX_MASK_1
    s0 = (s0 - ((s0 >> rate0) & mask0)) & 0xffffu;
    s1 = (s1 - ((s1 >> rate1) & mask1)) & 0xffffu;
    if (bin) { s0 = (s0 + ((0x7fffu >> rate0) & mask0)) & 0xffffu; s1 = (s1 + ((0x7fffu >> rate1) & mask1)) & 0xffffu; }
    sModel[l][lane] = s0 | (s1 << 16);
  };
  // ---- which groups hold coefficients, last significant position ----
  unsigned long long sig_cg = 0;
  int scan_pos_last = -1;
  for (int i = 0; i < wh; ++i)
    if (C[sScan[i]]) { scan_pos_last = i; sig_cg |= 1ull << sScanCg[i >> 4]; }
  if (scan_pos_last >= 0) {                                           // an empty block costs 0 bits (rdo.c:312-320)
    const int scan_cg_last = scan_pos_last >> 4;
    const int pos_last = sScan[scan_pos_last];
    const int last_y = pos_last >> l2w, last_x = pos_last - (last_y << l2w);
    {
      const int max_lfnst_pos = ((height == 4 && width == 4) || (height == 8 && width == 8)) ? 7 : 15;
      if (scan_pos_last > max_lfnst_pos) flags |= 1;
      if (scan_pos_last >= 1) flags |= 2;
    }
    // ---- uvg_encode_last_significant_xy ----
    {
      auto prefix_ctx = [](int l2) { return l2 <= 2 ? 0 : l2 == 3 ? 3 : l2 == 4 ? 6 : l2 == 5 ? 10 : 15; };
      const int l2h = Pg->l2h;
      const int off_x = t ? 0 : prefix_ctx(l2w), off_y = t ? 0 : prefix_ctx(l2h);
      const int sh_x = t ? clampi(width >> 3, 0, 2) : ((l2w + 1) >> 2), sh_y = t ? clampi(height >> 3, 0, 2) : ((l2h + 1) >> 2);
      const int gx = cc_group_idx(last_x), gy = cc_group_idx(last_y);
      double bits = 0.0;
      int k = 0;
      for (; k < gx; k++) code_bin(L_LASTX + off_x + (k >> sh_x), 1, bits);
      if (gx < cc_group_idx(min(32, width) - 1)) code_bin(L_LASTX + off_x + (k >> sh_x), 0, bits);
      k = 0;
      for (; k < gy; k++) code_bin(L_LASTY + off_y + (k >> sh_y), 1, bits);
      if (gy < cc_group_idx(min(32, height) - 1)) code_bin(L_LASTY + off_y + (k >> sh_y), 0, bits);
      if (gx > 3) bits += (double)((gx - 2) / 2);
      if (gy > 3) bits += (double)((gy - 2) / 2);
      total += bits;
    }
    // ---- residual_coding_subblock for every group from the last one down ----
    auto nb_terms = [&](int blk_pos, uint32_t px, uint32_t py, int &sum_abs, int &num_pos, int &sum_full) {
      const int16_t *d = C + blk_pos;
      sum_abs = 0; num_pos = 0; sum_full = 0;
      auto upd = [&](int v) { const int a = abs(v); sum_abs += min(4 + (a & 1), a); num_pos += a ? 1 : 0; sum_full += a; };
      if (px < (uint32_t)width - 1) {
        upd(d[1]);
        if (px < (uint32_t)width - 2) upd(d[2]);
        if (py < (uint32_t)height - 1) upd(d[width + 1]);
      }
      if (py < (uint32_t)height - 1) {
        upd(d[width]);
        if (py < (uint32_t)height - 2) upd(d[2 * width]);
      }
    };
    auto rice_of = [&](int sum_full, int baselevel) { const int s = clampi(sum_full - 5 * baselevel, 0, 31); return (s >= 7) + (s >= 14) + (s >= 28); };
    double bits = 0.0;
    int temp_diag = -1, temp_sum = -1;
    int reg_bins = (wh * 28) >> 4;
    const int cg_width = min(width, 32) >> 2, cg_height = min(height, 32) >> 2;
    for (int i = scan_cg_last; i >= 0; i--) {
      const int cg_blk_pos = sScanCg[i];
      const int cg_pos_y = cg_blk_pos >> l2cgw, cg_pos_x = cg_blk_pos & (cgw - 1);
      if (i == scan_cg_last || i == 0) {
        sig_cg |= 1ull << cg_blk_pos;
      } else {
        unsigned right = 0, lower = 0;
        if (cg_pos_x + 1 < cg_width) right = (unsigned)(sig_cg >> (cg_blk_pos + 1)) & 1;
        if (cg_pos_y + 1 < cg_height) lower = (unsigned)(sig_cg >> (cg_blk_pos + cg_width)) & 1;
        code_bin(L_SIGGRP + ((right || lower) ? 1 : 0), (int)((sig_cg >> cg_blk_pos) & 1), bits);
      }
      if ((sig_cg >> cg_blk_pos) & 1) {
        const int min_sub_pos = i << 4;
        const int first_sig_pos = (i == scan_cg_last) ? scan_pos_last : (min_sub_pos + 15);
        int next_sig_pos = first_sig_pos;
        const int infer_sig_pos = (next_sig_pos != scan_pos_last) ? ((i != 0) ? min_sub_pos : -1) : next_sig_pos;
        int num_non_zero = 0;
        // first pass: context-coded flags while regular bins remain (:196-262)
        for (next_sig_pos = first_sig_pos; next_sig_pos >= min_sub_pos && reg_bins >= 4; next_sig_pos--) {
          const int blk_pos = sScan[next_sig_pos];
          const uint32_t pos_y = (uint32_t)blk_pos >> l2w, pos_x = (uint32_t)blk_pos - (pos_y << l2w);
          const int v = C[blk_pos];
          const bool coded_sig = num_non_zero || next_sig_pos != infer_sig_pos;
          if (coded_sig || next_sig_pos != scan_pos_last) {
            int sum_abs, num_pos, sum_full;
            nb_terms(blk_pos, pos_x, pos_y, sum_abs, num_pos, sum_full);
            const int diag = (int)(pos_x + pos_y);
            int ctx_sig = min((sum_abs + 1) >> 1, 3) + (diag < 2 ? 4 : 0);
            if (color == 0) ctx_sig += diag < 5 ? 4 : 0;
            temp_diag = diag; temp_sum = sum_abs - num_pos;
            if (coded_sig) {
              code_bin(L_SIG + (t ? min(ctx_sig, 7) : ctx_sig), v != 0, bits);
              reg_bins--;
            }
          }
          if (v != 0) {
            num_non_zero++;
            int offset = 0;                                           // ctxOffsetAbs (:218-226)
            if (temp_diag != -1) {
              offset = min(temp_sum, 4) + 1;
              offset += !temp_diag ? (color == 0 ? 15 : 5) : (color == 0 ? (temp_diag < 3 ? 10 : (temp_diag < 10 ? 5 : 0)) : 0);
            }
            int rem = abs(v) - 1;
            const int gt1 = rem ? 1 : 0;
            code_bin(L_GT1 + offset, gt1, bits);
            reg_bins--;
            if (gt1) {
              rem -= 1;
              code_bin(L_PAR + offset, rem & 1, bits);
              rem >>= 1;
              reg_bins--;
              code_bin(L_GT2 + offset, rem ? 1 : 0, bits);
              reg_bins--;
            }
          }
        }
        // second pass: Golomb-Rice remainders of the context-coded positions (:268-281)
        for (int scan_pos = first_sig_pos; scan_pos > next_sig_pos; scan_pos--) {
          const int blk_pos = sScan[scan_pos];
          const uint32_t a = (uint32_t)abs((int)C[blk_pos]);
          if (a >= 4) {
            const uint32_t pos_y = (uint32_t)blk_pos >> l2w, pos_x = (uint32_t)blk_pos - (pos_y << l2w);
            int sum_abs, num_pos, sum_full;
            nb_terms(blk_pos, pos_x, pos_y, sum_abs, num_pos, sum_full);
            bits += (double)cc_remain_bits((a - 4) >> 1, (uint32_t)rice_of(sum_full, 4));
          }
        }
        // bypass-coded positions once the regular bins are spent (:286-305)
        for (int scan_pos = next_sig_pos; scan_pos >= min_sub_pos; scan_pos--) {
          const int blk_pos = sScan[scan_pos];
          const uint32_t pos_y = (uint32_t)blk_pos >> l2w, pos_x = (uint32_t)blk_pos - (pos_y << l2w);
          const uint32_t a = (uint32_t)abs((int)C[blk_pos]);
          int sum_abs, num_pos, sum_full;
          nb_terms(blk_pos, pos_x, pos_y, sum_abs, num_pos, sum_full);
          const uint32_t rice = (uint32_t)rice_of(sum_full, 0);
          const uint32_t pos0 = 1u << rice;                           // quant_state < 2
          const uint32_t remainder = a == 0 ? pos0 : (a <= pos0 ? a - 1 : a);
          bits += (double)cc_remain_bits(remainder, rice);
          if (a) num_non_zero++;
        }
        if (color == 0 && first_sig_pos > 0) flags |= 4;
        bits += (double)num_non_zero;                                 // signs: bypass
      }
      if (color == 0 && (cg_pos_y > 3 || cg_pos_x > 3) && ((sig_cg >> cg_blk_pos) & 1)) flags |= 8;
    }
    total += bits;
  }
  bits_out[blk0 + tid] = total;
  if (flags_out) flags_out[blk0 + tid] = (uint8_t)flags;
}

}  // namespace

extern "C" int uvghip_coeff_cost_batch(const int16_t *coeff, int width, int height, int n, int color,
                                       const uvghip_cabac_models_t *models_host, double *bits_out, uint8_t *flags_out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  auto pow2 = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32; };
  if (!pow2(width) || !pow2(height) || color < 0 || color > 2 || !coeff || !models_host || !bits_out)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  cc_params P;
  P.width = width; P.height = height; P.n = n; P.color = color;
  P.l2w = 31 - __builtin_clz(width); P.l2h = 31 - __builtin_clz(height);
  P.models = *models_host;
  hipStream_t st = uvghip_stream(stream);
  const int wh = width * height;
  const int lanes = wh <= 256 ? 64 : 16;                               // blocks per workgroup: coefficient staging must fit in LDS
  const size_t lds = (size_t)lanes * (wh + 2) * sizeof(int16_t);
  const int grid = (n + lanes - 1) / lanes;
  if (lanes == 64) coeff_cost_kernel<64><<<grid, 64, lds, st>>>(P, coeff, bits_out, flags_out);
  else coeff_cost_kernel<16><<<grid, 64, lds, st>>>(P, coeff, bits_out, flags_out);
  UVGHIP_CHECK_LAUNCH();
}
