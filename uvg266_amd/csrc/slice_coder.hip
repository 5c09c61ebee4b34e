// uvghip_encode_slice_rows: the arithmetic coder on the device -- the slice data of all-intra pictures from what the CTU search
// and the SAO decision left in device memory.
// replaces: encoder_state_worker_encode_lcu_bitstream for every CTU (src/encoderstate.c:862-939): encode_sao (:523-608),
// uvg_encode_coding_tree (src/encode_coding_tree.c:1365-1727: uvg_write_split_flag :1240-1363, the intra luma / chroma mode syntax
// :902-1238, cbf flags and encode_transform_coeff :628-900), uvg_encode_coeff_nxn (strategies/generic/encode_coding_tree-generic.c:53-323)
// with uvg_encode_last_significant_xy (encode_coding_tree.c:415-470), and the arithmetic coder itself (src/cabac.c:63-413) down to
// the end of the substream (end_of_sub_stream_one_bit, uvg_cabac_finish, alignment) and the emulation prevention of
// uvg_bitstream_put_byte (src/bitstream.c:215-226).
//
// A WPP row is one substream: its coder starts fresh (uvg_cabac_start) and its context models start from the models after the
// first CTU of the row above -- which the search already returned per CTU (the third model set) and uvghip_sao_decide_pictures
// returned for the two SAO models.  So rows are independent: one workgroup (one wave) per (picture, row).  Inside a row the coder is
// a chain bin after bin; lane 0 walks it, the other lanes only help to stage a transform block's levels into LDS.
#include "uvghip_common.h"
#include "ctu_core.h"
#include "inter_cand_dev.h"

namespace {

using namespace ctu;

// the 18 models of the inter syntax behind the others (the order of tools/refcheck/ctu_dump.c's snapshot_inter and of the oracle)
enum { NM_INTER = 18, MI = NMODELS + 2, MI_SKIP = MI + 0, MI_PRED_MODE = MI + 3, MI_MERGE_FLAG = MI + 5, MI_MERGE_IDX = MI + 6, MI_INTER_DIR = MI + 7,
       MI_REF_PIC = MI + 13, MI_MVD = MI + 15, MI_MVP_IDX = MI + 17, M_ROOT_CBF_ = 243 };

// the 18 models of the CTU-level ALF syntax behind those (alf_ctb_flag [9], the APS / fixed set switch, chroma alternatives [2], CC-ALF control [6])
enum { NM_ALF = 18, MA = NMODELS + 2 + NM_INTER };
// one picture's ALF decisions as the coder needs them (uvghip_slice_alf_t with device pointers)
struct alf_dev { int32_t alf_type, enabled[3], n_luma_aps, cc_enabled[2], cc_count[2], n_alts; const uint8_t *flags; const int16_t *set_idx; };

// H.266 6.5.2: the up-right diagonal scans of the four square shapes (4x4 groups in diagonal order, the 16 positions of a group
// likewise), at scan_base(log2 size): the same for every row of every picture -- a table of the code object, not 2.7 KB of every
// workgroup's LDS (with them a row's image was 12.7 KB: three rows to the 40 KB a search workgroup leaves behind, now four)
struct scan_table { uint16_t v[1360]; };
constexpr scan_table make_scan_table()
{
  scan_table t = {};
  int in[16] = {};
  int q = 0;
  for (int d = 0; d < 7; ++d) for (int x = 0; x <= d; ++x) { const int y = d - x; if (x < 4 && y < 4) in[q++] = y * 4 + x; }
  for (int l2 = 2; l2 <= 5; ++l2) {
    const int n = 1 << l2, cgw = n >> 2;
    const int base = l2 == 5 ? 0 : l2 == 4 ? 1024 : l2 == 3 ? 1280 : 1344;          // ctu::scan_base
    int g = 0;
    for (int d = 0; d < 2 * cgw - 1; ++d)
      for (int x = 0; x <= d; ++x) {
        const int y = d - x;
        if (x >= cgw || y >= cgw) continue;
        for (int k = 0; k < 16; ++k) t.v[base + g * 16 + k] = (uint16_t)((y * 4 + (in[k] >> 2)) * n + x * 4 + (in[k] & 3));
        ++g;
      }
  }
  return t;
}
__device__ const scan_table k_scan = make_scan_table();

struct row_state {
  uint32_t models[NMODELS + 2 + NM_INTER + NM_ALF];   // state0 | state1 << 16; [NMODELS] sao_merge_flag, [NMODELS + 1] sao_type_idx, the inter syntax, the ALF syntax
  uint8_t rate[NMODELS + 2 + NM_INTER + NM_ALF];
  int16_t lv[1024];                    // the levels of the transform block being coded, raster
  struct cui { uint8_t type, log2w, cbf, mode, mode_c, skipped, pad[2]; } cu[17 * 17];      // the CTU's side information + the row / column before it
  // what stage() finds out about the staged block with all lanes, so that the coding lane does not walk 1024 positions for it: the last
  // significant scan position (-1: none), the groups that hold a level by scan index / by raster position (the last group counted in)
  int32_t ci_last;
  unsigned long long ci_cg, ci_r;
  // ... and per raster position (of the scan positions up to the last one): what the coding lane needs of the position's context
  // template -- the five levels right of and below it -- so that it reads one word where it would read five levels and do the
  // arithmetic of context_get_sig_ctx_idx_abs / uvg_abs_sum per coefficient: sig context (4 bits, chroma's cap applied) | context
  // offset of the gt1 / parity / gt2 flags << 4 (5 bits) | Rice parameter of a remainder behind the context-coded flags << 9 |
  // Rice parameter of a bypass-coded position << 11
  // ... by SCAN position, with the position's level in the low half: the coding lane reads one word per position, in the order it
  // visits them (the next one is in flight while the bins of this one are coded), instead of scan -> level, scan -> template
  uint32_t ci_pos[1024];
  int32_t sao_l[34];                   // the CTU's SAO decision (fetched by the wave: a row that codes BEHIND the launch that decides it reads past the caches)
};
// P / B slices: the CTU's motion for the AMVP predictors (the table uvg_inter_get_mv_cand_cua reads of the picture's cu array), the row's
// history table, the picture's reference lists
struct row_state_pb {
  icand::unit tab[17 * 17 + 1];
  uint8_t flags[17 * 17][8];           // uvghip_inter4_t of the CTU's units (border included)
  int32_t hmvp[41];
  int32_t pred[4], ref_idx[2];         // (indexed dynamically: not in registers)
  icand::amvp_ws ws;
  icand::frame_ctx f;
};
struct lds_tab { icand::unit *p; __device__ icand::unit &at(int i) { return p[i]; } };
struct glb_col {
  const int32_t *p;
  int stride, gw;                     // stride > 0: p is a per-4x4 table of `stride` units per row, read at the even positions
  __device__ icand::col_unit at(int i) const
  {
    const int gy = stride > 0 ? i / gw : 0, gx = i - gy * gw;
    const int32_t *o = stride > 0 ? p + ((size_t)(gy * 2) * stride + gx * 2) * 8 : p + (size_t)i * 8;
    icand::col_unit c;
    c.type = o[0]; c.mv[0][0] = o[1]; c.mv[0][1] = o[2]; c.mv[1][0] = o[3]; c.mv[1][1] = o[4]; c.dir = o[5]; c.poc[0] = o[6]; c.poc[1] = o[7];
    return c;
  }
};

struct coder {                          // cabac_data_t (cabac.h:56-66) + the substream's output
  uint32_t low, range, buffered;
  int bits_left, nbuf;
  uint8_t *out;
  int n, cap, zeros;
};

// a row coded beside the launch that finishes its pictures (wait_flags: that launch's per-CTU "final" flags): one lane waits for the CTU
__device__ inline void wait_final(const int32_t *flag)
{
  int naps = 1;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
    if (naps < 16) naps <<= 1;
  }
}
__device__ __forceinline__ void put_byte(coder &c, uint32_t b)          // uvg_bitstream_put_byte: emulation prevention
{
  b &= 0xff;
  if (c.zeros == 2 && b < 4) { if (c.n < c.cap) c.out[c.n] = 3; c.n++; c.zeros = 0; }
  c.zeros = b == 0 ? c.zeros + 1 : 0;
  if (c.n < c.cap) c.out[c.n] = (uint8_t)b;
  c.n++;
}
__device__ __forceinline__ void cwrite(coder &c)                          // uvg_cabac_write
{
  const uint32_t lead = c.low >> (24 - c.bits_left);
  c.bits_left += 8;
  c.low &= 0xffffffffu >> c.bits_left;
  if (lead == 0xff) { c.nbuf++; return; }
  if (c.nbuf > 0) {
    const uint32_t carry = lead >> 8;
    put_byte(c, c.buffered + carry);
    c.buffered = lead & 0xff;
    const uint32_t fill = (0xff + carry) & 0xff;
    while (c.nbuf > 1) { put_byte(c, fill); c.nbuf--; }
  } else { c.nbuf = 1; c.buffered = lead; }
}
__device__ __forceinline__ void enc_bin(coder &c, row_state *R, int idx, int bin)      // uvg_cabac_encode_bin + CTX_UPDATE
{
#if defined(SLC_EXP_NO_BINS)
  c.low += idx + bin; return;      // (timing experiment)
#endif
  const uint32_t st = R->models[idx];
  uint32_t s0 = st & 0xffffu, s1 = st >> 16;
  const uint32_t state = (s0 + s1) >> 8;
  const uint32_t q = (state & 0x80) ? (state ^ 0xff) : state;
  const uint32_t lps = ((((q >> 2) * (c.range >> 5)) >> 1) + 4) & 0xff;
  c.range -= lps;
  if ((uint32_t)(bin ? 1 : 0) != (state >> 7)) {
    const int nb = __clz((int)lps) - 23;            // uvg_g_auc_renorm_table[lps >> 3]: shifts that bring lps (>= 4) to 256..511
    c.low = (c.low + c.range) << nb;
    c.range = lps << nb;
    c.bits_left -= nb;
    if (c.bits_left < 12) cwrite(c);
  } else if (c.range < 256) {
    c.low <<= 1; c.range <<= 1; c.bits_left--;
    if (c.bits_left < 12) cwrite(c);
  }
  const int r0 = R->rate[idx] >> 4, r1 = R->rate[idx] & 15;
  s0 -= (s0 >> r0) & 0x7fe0u;
  s1 -= (s1 >> r1) & 0x7ffeu;
  if (bin) { s0 += (0x7fffu >> r0) & 0x7fe0u; s1 += (0x7fffu >> r1) & 0x7ffeu; }
  R->models[idx] = (s0 & 0xffffu) | (s1 << 16);
}
__device__ inline void ctx_update(row_state *R, int idx, int bin)          // CTX_UPDATE alone: a bin another row's coder sent through this model
{
  const uint32_t st = R->models[idx];
  uint32_t s0 = st & 0xffffu, s1 = st >> 16;
  const int r0 = R->rate[idx] >> 4, r1 = R->rate[idx] & 15;
  s0 -= (s0 >> r0) & 0x7fe0u;
  s1 -= (s1 >> r1) & 0x7ffeu;
  if (bin) { s0 += (0x7fffu >> r0) & 0x7fe0u; s1 += (0x7fffu >> r1) & 0x7ffeu; }
  R->models[idx] = (s0 & 0xffffu) | (s1 << 16);
}
__device__ __forceinline__ void enc_ep(coder &c, int bin)                 // uvg_cabac_encode_bin_ep
{
  c.low <<= 1;
  if (bin) c.low += c.range;
  c.bits_left--;
  if (c.bits_left < 12) cwrite(c);
}
__device__ __forceinline__ void enc_eps(coder &c, uint32_t v, int n)      // uvg_cabac_encode_bins_ep / _aligned_bins_ep
{
  if (c.range == 256) {
    int rem = n;
    while (rem > 0) {
      const int k = rem < 8 ? rem : 8;
      const uint32_t nb = (v >> (rem - k)) & ((1u << k) - 1);
      c.low = (c.low << k) + (nb << 8);
      rem -= k;
      c.bits_left -= k;
      if (c.bits_left < 12) cwrite(c);
    }
    return;
  }
  while (n > 8) {
    n -= 8;
    const uint32_t pattern = v >> n;
    c.low <<= 8;
    c.low += c.range * pattern;
    v -= pattern << n;
    c.bits_left -= 8;
    if (c.bits_left < 12) cwrite(c);
  }
  c.low <<= n;
  c.low += c.range * v;
  c.bits_left -= n;
  if (c.bits_left < 12) cwrite(c);
}
__device__ __forceinline__ void enc_remain(coder &c, uint32_t remainder, uint32_t rice)   // uvg_cabac_write_coeff_remain, cutoff 5
{
  const unsigned cutoff = 5, threshold = cutoff << rice;
  if (remainder < threshold) {
    const uint32_t length = (remainder >> rice) + 1;
    enc_eps(c, (1u << length) - 2, (int)length);
    enc_eps(c, remainder & ((1u << rice) - 1), (int)rice);
  } else {
    const unsigned max_prefix = 32 - cutoff - 15;
    unsigned prefix_length = 0, suffix_length;
    const unsigned code_value = (remainder >> rice) - cutoff;
    if ((int32_t)code_value >= ((1 << max_prefix) - 1)) { prefix_length = max_prefix; suffix_length = 15; }
    else { while ((int32_t)code_value > ((2 << prefix_length) - 2)) prefix_length++; suffix_length = prefix_length + rice + 1; }
    const unsigned total_prefix = prefix_length + cutoff;
    enc_eps(c, (1u << total_prefix) - 1, (int)total_prefix);
    enc_eps(c, ((code_value - ((1u << prefix_length) - 1)) << rice) | (remainder & ((1u << rice) - 1)), (int)suffix_length);
  }
}

__device__ inline void code_mvd(coder &c, row_state *R, int mvd_hor, int mvd_ver)
{
  const uint32_t ah = (uint32_t)(mvd_hor < 0 ? -mvd_hor : mvd_hor), av = (uint32_t)(mvd_ver < 0 ? -mvd_ver : mvd_ver);
  enc_bin(c, R, MI_MVD + 0, mvd_hor != 0);
  enc_bin(c, R, MI_MVD + 0, mvd_ver != 0);
  if (ah) enc_bin(c, R, MI_MVD + 1, ah > 1);
  if (av) enc_bin(c, R, MI_MVD + 1, av > 1);
  for (int k = 0; k < 2; ++k) {
    const uint32_t a = k ? av : ah;
    const int v = k ? mvd_ver : mvd_hor;
    if (!a) continue;
    if (a > 1) {                                  // uvg_cabac_write_ep_ex_golomb(a - 2, 1) (cabac.c:320-354)
      uint32_t symbol = a - 2, count = 1, bins = 0;
      int num_bins = 0;
      while (symbol >= (1u << count)) { bins = 2 * bins + 1; ++num_bins; symbol -= 1u << count; ++count; }
      bins = 2 * bins; ++num_bins;
      bins = (bins << count) | symbol;
      num_bins += (int)count;
      enc_eps(c, bins, num_bins);
    }
    enc_ep(c, v > 0 ? 0 : 1);
  }
}

// uvg_encode_coeff_nxn on the levels staged in R->lv (n x n, raster); lane 0
__device__ __forceinline__ void code_coeffs(coder &c, row_state *R, int n, int color)
{
#if defined(SLC_EXP_NO_COEFF)
  return;      // (timing experiment: tools/dev/coder_time.py)
#endif
  const int l2 = ilog2_dev(n), nn = n * n, cgw = n >> 2, t = color ? 1 : 0;
  const uint16_t *scan = k_scan.v + scan_base(l2);
  const int last = R->ci_last;                         // (stage() looked at the whole block with all lanes)
  if (last < 0) return;
  const unsigned long long sig_cg = R->ci_cg, sig_r = R->ci_r;      // per group, by scan index / by raster position: has a level
  (void)nn;
  const int cg_last = last >> 4;
  {   // uvg_encode_last_significant_xy
    const int pos_last = scan[last], last_y = pos_last >> l2, last_x = pos_last - (last_y << l2);
    const int off = t ? 0 : (l2 == 2 ? 0 : l2 == 3 ? 3 : l2 == 4 ? 6 : 10);          // prefix_ctx[log2 size]
    const int sh = t ? clampi(n >> 3, 0, 2) : ((l2 + 1) >> 2);
    const int bx = M_LASTX + 20 * t + off, by = M_LASTY + 20 * t + off;
    const int gx = group_idx(last_x), gy = group_idx(last_y), gmax = group_idx(n - 1);
    int k = 0;
    for (; k < gx; k++) enc_bin(c, R, bx + (k >> sh), 1);
    if (gx < gmax) enc_bin(c, R, bx + (k >> sh), 0);
    k = 0;
    for (; k < gy; k++) enc_bin(c, R, by + (k >> sh), 1);
    if (gy < gmax) enc_bin(c, R, by + (k >> sh), 0);
    // g_min_in_group[g] for g > 3: 4, 6, 8, 12, 16, 24 = (2 + (g & 1)) << ((g >> 1) - 1)
    if (gx > 3) enc_eps(c, (uint32_t)(last_x - ((2 + (gx & 1)) << ((gx >> 1) - 1))), (gx - 2) / 2);
    if (gy > 3) enc_eps(c, (uint32_t)(last_y - ((2 + (gy & 1)) << ((gy >> 1) - 1))), (gy - 2) / 2);
  }
  int reg_bins = (nn * 28) >> 4;
  // the groups' positions in the block: group g of the scan sits at the 4x4 whose first coefficient is scan[g * 16]
  for (int g = cg_last; g >= 0; --g) {
    const int first = scan[g * 16];
    const int cx = (first & (n - 1)) >> 2, cy = (first >> l2) >> 2;
    int sig = (int)((sig_cg >> g) & 1);
    if (g == cg_last || g == 0) sig = 1;
    else {
      const int right = cx + 1 < cgw ? (int)((sig_r >> (cy * cgw + cx + 1)) & 1) : 0;
      const int lower = cy + 1 < cgw ? (int)((sig_r >> ((cy + 1) * cgw + cx)) & 1) : 0;
      enc_bin(c, R, M_SIGGRP + 2 * t + ((right || lower) ? 1 : 0), sig);
    }
    if (!sig) continue;
    const int min_sub = g << 4;
    const int first_sig = (g == cg_last) ? last : (min_sub + 15);
    const int infer_sig = (first_sig != last) ? ((g != 0) ? min_sub : -1) : first_sig;
    int num_nz = 0, next_sig;
    uint32_t signs = 0;
    uint32_t ahead = R->ci_pos[first_sig];
    for (next_sig = first_sig; next_sig >= min_sub && reg_bins >= 4; next_sig--) {
      const uint32_t w = ahead;                            // (stage(): the position's level and context template, all lanes)
      if (next_sig > 0) ahead = R->ci_pos[next_sig - 1];
      const int level = (int)(int16_t)(w & 0xffffu), rec = (int)(w >> 16);
      const int s = level != 0;
      if (num_nz || next_sig != infer_sig) {
        enc_bin(c, R, M_SIG + 12 * t + (rec & 15), s);
        reg_bins--;
      }
      if (s) {
        num_nz++;
        signs = (signs << 1) | (level < 0);
        // (the block's last position is coded before any template was looked at: offset 0, encode_coding_tree-generic.c:203-215)
        const int ofs = next_sig == last ? 0 : (rec >> 4) & 31;
        int rem = iabs_(level) - 1;
        const int gt1 = rem ? 1 : 0;
        enc_bin(c, R, M_GT1 + 21 * t + ofs, gt1);
        reg_bins--;
        if (gt1) {
          rem -= 1;
          enc_bin(c, R, M_PAR + 21 * t + ofs, rem & 1);
          rem >>= 1;
          reg_bins--;
          enc_bin(c, R, M_GT2 + 21 * t + ofs, rem ? 1 : 0);
          reg_bins--;
        }
      }
    }
    for (int sp = first_sig; sp > next_sig; sp--) {               // Golomb-Rice remainders of the context-coded positions
      const uint32_t w = R->ci_pos[sp];
      const uint32_t a = (uint32_t)iabs_((int)(int16_t)(w & 0xffffu));
      if (a >= 4) enc_remain(c, (a - 4) >> 1, (w >> (16 + 9)) & 3);
    }
    for (int sp = next_sig; sp >= min_sub; sp--) {                 // positions coded in bypass once the regular bins are spent
      const uint32_t w = R->ci_pos[sp];
      const int level = (int)(int16_t)(w & 0xffffu);
      const uint32_t a = (uint32_t)iabs_(level);
      const uint32_t rice = (w >> (16 + 11)) & 3, pos0 = 1u << rice;
      enc_remain(c, a == 0 ? pos0 : (a <= pos0 ? a - 1 : a), rice);
      if (a) { num_nz++; signs = (signs << 1) | (level < 0); }
    }
    enc_eps(c, signs, num_nz);
  }
}

struct pic_dev { const uvghip_scu_t *cu; const int16_t *coeff; const uint32_t *models; int cu_stride, pad; };
// a P / B picture's reference lists and side tables (uvghip_slice_pb_t with device pointers), behind the pic_dev table in the workspace
struct pb_dev {
  int32_t slice_type, poc, n_refs, ref_pocs[16], l_size[2], l[2][16], tmvp, max_merge, merge_level, frame_qp;
  const int32_t *col;
  const uvghip_inter4_t *inter4;
  const uint32_t *models_inter;
  int32_t col_stride, pad;
};

// uvg_hmvp_add_mv (src/inter.c:1831-1905) on the row's table: [0] size, then five units, most recent first
__device__ inline void hmvp_push(int32_t *hm, const icand::unit &u)
{
  icand::unit *lut = reinterpret_cast<icand::unit *>(hm + 1);
  const int size = hm[0];
  int dup = -1;
  for (int i = 0; i < size; ++i) if (icand::same_motion(u, &lut[i])) { dup = i; break; }
  if (dup != 0) {
    int end = dup == -1 ? 5 : dup;
    if (end > 4) end = 4;
    if (end == 0 && size == 1) end = 1;
    for (int i = end - 1; i >= 0; --i) lut[i + 1] = lut[i];
  }
  lut[0] = u;
  if (dup == -1 && hm[0] < 5) hm[0]++;
}

__device__ __forceinline__ void enc_eps(coder &c, uint32_t v, int n);
// uvg_encode_mvd (src/encode_coding_tree.c:1865-1910): greater-than-0 / -1 flags, the remainder as Exp-Golomb of order 1, the sign
__device__ inline void code_mvd(coder &c, row_state *R, int mvd_hor, int mvd_ver);

// side information of the 4x4 unit at picture position (x, y), from the CTU's LDS copy (x0, y0: the CTU's origin; one unit of
// border to the left and above)
__device__ __forceinline__ const row_state::cui &cu_of(const row_state *R, int x0, int y0, int x, int y)
{
  return R->cu[(((y - y0) >> 2) + 1) * 17 + ((x - x0) >> 2) + 1];
}

__device__ __forceinline__ void stage(row_state *R, const int16_t *src, int stride, int n, int color)        // all lanes: n x n levels -> R->lv
{
  __syncthreads();
  const int l2 = ilog2_dev(n), nn = n * n, cgw = n >> 2;
  for (int e = threadIdx.x; e < nn; e += blockDim.x) R->lv[e] = src[(e >> l2) * stride + (e & (n - 1))];
  __syncthreads();
  // (one wave per row: blockDim.x == 64)
  const uint16_t *scan = k_scan.v + scan_base(l2);
  int last = -1;
  for (int sp = threadIdx.x; sp < nn; sp += 64) if (R->lv[scan[sp]]) last = sp;
  for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(last, o, 64); last = v > last ? v : last; }
  unsigned long long cg = 0, rr = 0;
  if (last >= 0) {
    const int g = threadIdx.x;
    bool mine = false;
    unsigned long long rb = 0;
    if (g <= (last >> 4)) {
      int any = 0;
      for (int k = 0; k < 16; ++k) any |= R->lv[scan[g * 16 + k]] != 0;
      if (any || g == (last >> 4)) {                     // (the last group counts as significant for its neighbours)
        const int f = scan[g * 16];
        mine = true;
        rb = 1ull << (((f >> l2) >> 2) * cgw + ((f & (n - 1)) >> 2));
      }
    }
    cg = __ballot(mine);
    unsigned lo = (unsigned)rb, hi = (unsigned)(rb >> 32);
    for (int o = 32; o >= 1; o >>= 1) { lo |= (unsigned)__shfl_xor((int)lo, o, 64); hi |= (unsigned)__shfl_xor((int)hi, o, 64); }
    rr = (unsigned long long)hi << 32 | lo;
  }
  if (threadIdx.x == 0) { R->ci_last = last; R->ci_cg = cg; R->ci_r = rr; }
  // the context template of every position the coder will visit (context_get_sig_ctx_idx_abs rdo.c:1400-1438, uvg_abs_sum context.c:846-877)
  for (int sp = threadIdx.x; sp <= last; sp += 64) {
    const int blk = scan[sp], py = blk >> l2, px = blk - (py << l2);
    const int16_t *d = R->lv + blk;
    int num_pos = 0, sum_cap = 0, sum = 0;
#define SLC_UPD(v) { const int a = iabs_((int)(v)); sum += a; sum_cap += (4 + (a & 1)) < a ? (4 + (a & 1)) : a; num_pos += a ? 1 : 0; }
    if (px < n - 1) {
      SLC_UPD(d[1]);
      if (px < n - 2) SLC_UPD(d[2]);
      if (py < n - 1) SLC_UPD(d[n + 1]);
    }
    if (py < n - 1) {
      SLC_UPD(d[n]);
      if (py < n - 2) SLC_UPD(d[n << 1]);
    }
#undef SLC_UPD
    const int diag = px + py, tsum = sum_cap - num_pos;
    int ctx_sig = (((sum_cap + 1) >> 1) < 3 ? ((sum_cap + 1) >> 1) : 3) + (diag < 2 ? 4 : 0);
    if (color == 0) ctx_sig += diag < 5 ? 4 : 0;
    else if (ctx_sig > 7) ctx_sig = 7;
    const int ofs = ((tsum < 4 ? tsum : 4) + 1) + (!diag ? (color == 0 ? 15 : 5) : color == 0 ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
    const int s4 = sum - 20 < 0 ? 0 : (sum - 20 < 31 ? sum - 20 : 31), s0 = sum < 31 ? sum : 31;
    R->ci_pos[sp] = (uint32_t)(uint16_t)R->lv[blk] | (uint32_t)(ctx_sig | ofs << 4 | go_rice_par((unsigned)s4) << 9 | go_rice_par((unsigned)s0) << 11) << 16;
  }
  __syncthreads();
}

__device__ __forceinline__ void code_split_flag(coder &c, row_state *R, int x0, int y0, int W, int H, int x, int y, int s, int split)
{
  if (!(W >= x + s && H >= y + s) || s <= 4) return;                // implicit split, or nothing to split
  int model = 0;
  if (x > 0 && (1 << cu_of(R, x0, y0, x - 4, y).log2w) < s) model++;
  if (y > 0 && (1 << cu_of(R, x0, y0, x, y - 4).log2w) < s) model++;
  enc_bin(c, R, M_SPLIT + model, split);
}

__device__ __forceinline__ void code_luma_mode(coder &c, row_state *R, int x0, int y0, int x, int y, int n, int mode)
{
  // uvg_intra_get_dir_luma_predictor (intra.c:88-188), MIP off, on the two neighbours' modes (0 = not intra / not there)
  int left_dir = 0, above_dir = 0;
  if (x > 0) { const row_state::cui &q = cu_of(R, x0, y0, x - 4, y + n - 4); if (q.type == 1) left_dir = q.mode; }
  if ((y & 63) > 0 && y > 0) { const row_state::cui &q = cu_of(R, x0, y0, x + n - 4, y - 4); if (q.type == 1) above_dir = q.mode; }
  int p0 = 0, p1 = 1, p2 = 50, p3 = 18, p4 = 46, p5 = 54;
  const int offset = 61, mod = 64;
  if (left_dir == above_dir) {
    if (left_dir > 1) {
      p1 = left_dir; p2 = ((left_dir + offset) % mod) + 2; p3 = ((left_dir - 1) % mod) + 2;
      p4 = ((left_dir + offset - 1) % mod) + 2; p5 = (left_dir % mod) + 2;
    }
  } else if (left_dir > 1 && above_dir > 1) {
    p1 = left_dir; p2 = above_dir;
    const int mx = p1 > p2 ? p1 : p2, mn = p1 > p2 ? p2 : p1, diff = mx - mn;
    if (diff == 1) { p3 = ((mn + offset) % mod) + 2; p4 = ((mx - 1) % mod) + 2; p5 = ((mn + offset - 1) % mod) + 2; }
    else if (diff >= 62) { p3 = ((mn - 1) % mod) + 2; p4 = ((mx + offset) % mod) + 2; p5 = (mn % mod) + 2; }
    else if (diff == 2) { p3 = ((mn - 1) % mod) + 2; p4 = ((mn + offset) % mod) + 2; p5 = ((mx - 1) % mod) + 2; }
    else { p3 = ((mn + offset) % mod) + 2; p4 = ((mn - 1) % mod) + 2; p5 = ((mx + offset) % mod) + 2; }
  } else if (left_dir + above_dir >= 2) {
    p1 = left_dir < above_dir ? above_dir : left_dir;
    p2 = ((p1 + offset) % mod) + 2; p3 = ((p1 - 1) % mod) + 2; p4 = ((p1 + offset - 1) % mod) + 2; p5 = (p1 % mod) + 2;
  }
  const int preds[6] = {p0, p1, p2, p3, p4, p5};
  int mpm = -1;
#pragma unroll
  for (int i = 5; i >= 0; --i) if (preds[i] == mode) mpm = i;
  enc_bin(c, R, M_MPM, mpm != -1);
  if (mpm != -1) {
    enc_bin(c, R, M_PLANAR + 1, mpm > 0);
    if (mpm > 0) enc_ep(c, mpm > 1);
    if (mpm > 1) enc_ep(c, mpm > 2);
    if (mpm > 2) enc_ep(c, mpm > 3);
    if (mpm > 3) enc_ep(c, mpm > 4);
  } else {
    int tmp = mode;
#pragma unroll
    for (int i = 0; i < 6; ++i) tmp -= preds[i] < mode;
    if (tmp < 3) enc_eps(c, (uint32_t)tmp, 5); else enc_eps(c, (uint32_t)tmp + 3, 6);      // truncated binary, 61 symbols
  }
}

__device__ __forceinline__ void code_chroma_mode(coder &c, row_state *R, int chroma_mode, int luma_dir)
{
  const int derived = chroma_mode == luma_dir;
  enc_bin(c, R, M_CHROMA_PRED, derived ? 0 : 1);
  if (!derived) {
    // the list {planar, vertical, horizontal, DC} with the entry equal to the luma mode standing for 66
    const int m0 = luma_dir == 0 ? 66 : 0, m1 = luma_dir == 50 ? 66 : 50, m2 = luma_dir == 18 ? 66 : 18;
    const int idx = chroma_mode == m0 ? 0 : chroma_mode == m1 ? 1 : chroma_mode == m2 ? 2 : 3;
    enc_eps(c, (uint32_t)idx, 2);
  }
}

__device__ __forceinline__ void code_sao_color(coder &c, row_state *R, const int32_t *info, int color, int max_off)
{
  const int type = info[0], off = color == 2 ? 5 : 0;
  if (color != 2) {
    enc_bin(c, R, NMODELS + 1, type != 0);
    if (type == 1) enc_ep(c, 0); else if (type == 2) enc_ep(c, 1);
  }
  if (type == 0) return;
  for (int i = 1; i <= 4; ++i) {
    int symbol = info[7 + i + off] < 0 ? -info[7 + i + off] : info[7 + i + off];
    const int code_last = max_off > symbol;
    enc_ep(c, symbol ? 1 : 0);
    if (!symbol) continue;
    while (--symbol) enc_ep(c, 1);
    if (code_last) enc_ep(c, 0);
  }
  if (type == 1) {
    for (int i = 1; i <= 4; ++i) if (info[7 + i + off] != 0) enc_ep(c, info[7 + i + off] < 0 ? 1 : 0);
    enc_eps(c, (uint32_t)info[5 + (color == 2 ? 1 : 0)], 5);
  } else if (color != 2) {
    enc_eps(c, (uint32_t)info[1], 2);
  }
}

__device__ inline int z_to_xy(int z) { return (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4) | ((z >> 3) & 8); }

// uvg_encode_alf_bits (src/alf.c:1365-1413) of CTU k, between its SAO syntax and its coding tree (encoderstate.c:880).  CODE = false: only the
// models move -- what the first CTU of every row above did to them before this row's coder started (the WPP hand-over, encoderstate.c:966-975).
template <bool CODE>
__device__ inline void alf_ctu(coder &c, row_state *R, const alf_dev &A, int k, int wc, int n)
{
  auto bin = [&](int idx, int b) { if (CODE) enc_bin(c, R, idx, b); else ctx_update(R, idx, b); };
  const int left = k % wc ? k - 1 : -1, above = k >= wc ? k - wc : -1;
  for (int comp = 0; comp < 3; ++comp) {
    const uint8_t *en = A.flags + (size_t)comp * n;
    if (A.enabled[comp]) bin(MA + comp * 3 + (left >= 0 && en[left]) + (above >= 0 && en[above]), en[k]);      // code_alf_ctu_enable_flag (:1147)
    if (comp == 0) {
      if (en[k] && A.enabled[0]) {                                             // code_alf_ctu_filter_index (:1209)
        const unsigned set = (unsigned)A.set_idx[k], n_aps = (unsigned)A.n_luma_aps;
        unsigned sym = set, max_v = 16;
        bool coded = true;
        if (n_aps > 0) {
          bin(MA + 9, set >= 16);
          if (set >= 16) { sym = set - 16; max_v = n_aps; coded = n_aps > 1; }
        }
        if (coded && CODE) {                                                   // uvg_cabac_encode_trunc_bin (cabac.c:203-229)
          int thresh = 0;
          while ((2u << thresh) <= max_v) ++thresh;
          const unsigned val = 1u << thresh, b = max_v - val;
          if (sym < val - b) enc_eps(c, sym, thresh); else enc_eps(c, sym + val - b, thresh + 1);
        }
      }
    } else if (A.enabled[comp] && en[k]) {                                     // code_alf_ctu_alternative_ctu (:1255)
      const int ones = A.flags[(size_t)(2 + comp) * n + k];
      for (int i = 0; i < ones; ++i) bin(MA + 10 + comp - 1, 1);
      if (ones < A.n_alts - 1) bin(MA + 10 + comp - 1, 0);
    }
  }
  if (A.alf_type == 2)
    for (int comp = 0; comp < 2; ++comp) {
      if (!A.cc_enabled[comp]) continue;                                       // code_cc_alf_filter_control_idc (:1321)
      const uint8_t *ctl = A.flags + (size_t)(5 + comp) * n;
      const int idc = ctl[k];
      bin(MA + 12 + (left >= 0 && ctl[left]) + (above >= 0 && ctl[above]) + 3 * comp, idc != 0);
      if (idc > 0 && CODE) {
        for (int v = idc - 1; v > 0; --v) enc_ep(c, 1);
        if (idc < A.cc_count[comp]) enc_ep(c, 0);
      }
    }
}

// PERSIST: the launch is a capped number of waves that take row after row from a ticket counter -- row r of every picture, then row r + 1:
// the order in which rows become codable behind a search that is still running (uvghip_encode_slice_rows_behind_capped) -- instead of a
// wave per row: a waiting wave holds 10 KB of LDS a search workgroup cannot use.
template <bool PERSIST = false>
__global__ void __launch_bounds__(64)
slice_rows_kernel(const pic_dev *__restrict__ pics, const pb_dev *__restrict__ pbs, const int32_t *__restrict__ sao, const uint16_t *__restrict__ sao_models,
                  int W, int H, int qp, int bitdepth, uint8_t *__restrict__ out, int row_cap, int32_t *__restrict__ row_bytes, const alf_dev *__restrict__ alfs = nullptr,
                  const int32_t *wait_flags = nullptr, int32_t *ticket = nullptr, int n_pictures = 0)
{
  __shared__ row_state Rs;
  __shared__ int s_job;
  extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];          // row_state_pb for P / B pictures (none for I)
  row_state *R = &Rs;
  row_state_pb *Q = reinterpret_cast<row_state_pb *>(pb_smem);
  const int wc = (W + 63) / 64, hc = (H + 63) / 64;
  for (;;) {
  int job = blockIdx.x;
  if (PERSIST) {
    if (threadIdx.x == 0) s_job = atomicAdd(ticket, 1);
    __syncthreads();
    job = s_job;
    if (job >= n_pictures * hc) break;
  }
  const int pic = PERSIST ? job % n_pictures : job / hc, cy = PERSIST ? job / n_pictures : job - pic * hc;
  const pic_dev D = pics[pic];
  const pb_dev *PB = pbs ? &pbs[pic] : nullptr;
  const int slice = PB ? PB->slice_type : 2, init_qp = PB ? PB->frame_qp : qp;
  const bool lane0 = threadIdx.x == 0;
  // behind a launch that is still producing the pictures: the row's start -- the models behind the first CTU of the row above, its SAO
  // models -- exists when that CTU is final (the search that wrote the models may itself still be running: uvghip_loop_plan_run_overlapped)
  if (wait_flags && cy > 0) {
    if (lane0) wait_final(wait_flags + (size_t)pic * wc * hc + (size_t)(cy - 1) * wc);
    __syncthreads();
  }
  // scans, window bytes, the row's start models
  for (int i = threadIdx.x; i < NMODELS; i += blockDim.x) {
    R->rate[i] = k_ctx_init[3][i];
    if (cy == 0) models_init_one(R->models, i, init_qp, slice);
    else R->models[i] = D.models[((size_t)((cy - 1) * wc) * 3 + 2) * NMODELS + i];
  }
  if (PB) {
    if (threadIdx.x < NM_INTER) {
      const int i = threadIdx.x;
      R->rate[MI + i] = k_ctx_init_inter[3][i];
      if (cy == 0) {
        const int v = k_ctx_init_inter[slice][i];
        const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
        int s = ((slope * (init_qp - 16)) >> 1) + offset;
        s = s < 1 ? 1 : (s > 127 ? 127 : s);
        R->models[MI + i] = (uint32_t)((s << 8) & 0x7fe0) | ((uint32_t)((s << 8) & 0x7ffe) << 16);
      } else R->models[MI + i] = PB->models_inter[((size_t)((cy - 1) * wc) * 3 + 2) * NM_INTER + i];
    }
    if (lane0) {
      for (int i = 0; i < 41; ++i) Q->hmvp[i] = 0;                 // the row's history table starts empty (encoderstate.c:1021-1028)
      icand::frame_ctx &f = Q->f;
      f.poc = PB->poc; f.is_b = PB->slice_type == 0; f.pic_w = W; f.pic_h = H; f.tmvp = PB->tmvp; f.max_cands = PB->max_merge; f.mer_level = PB->merge_level;
      f.wpp = 1; f.n_refs = PB->n_refs;
      for (int i = 0; i < 16; ++i) f.ref_pocs[i] = PB->ref_pocs[i];
      f.l_size[0] = PB->l_size[0]; f.l_size[1] = PB->l_size[1];
      for (int i = 0; i < 8; ++i) { f.l[0][i] = PB->l[0][i]; f.l[1][i] = PB->l[1][i]; }
    }
  }
  if (threadIdx.x < 2) {
    const int i = threadIdx.x;
    R->rate[NMODELS + i] = k_ctx_init_sao[3][i];
    if (cy == 0 || !sao_models) {
      const int v = k_ctx_init_sao[slice][i];
      const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
      int s = ((slope * (init_qp - 16)) >> 1) + offset;
      s = s < 1 ? 1 : (s > 127 ? 127 : s);
      R->models[NMODELS + i] = (uint32_t)((s << 8) & 0x7fe0) | ((uint32_t)((s << 8) & 0x7ffe) << 16);
    } else {
      const uint16_t *m = sao_models + ((size_t)pic * wc * hc + (size_t)(cy - 1) * wc) * 6 + 3 * i;
      const uint32_t m0 = __hip_atomic_load(&m[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), m1 = __hip_atomic_load(&m[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      R->models[NMODELS + i] = m0 | (m1 << 16);
    }
  }
  __syncthreads();
  coder c;
  c.low = 0; c.range = 510; c.bits_left = 23; c.nbuf = 0; c.buffered = 0xff;       // uvg_cabac_start
  c.out = out + ((size_t)pic * hc + cy) * row_cap; c.n = 0; c.cap = row_cap; c.zeros = 0;
  if (alfs && lane0) {
    // the ALF models of this row's start: initialised for the slice, then what the first CTU of every row above sent through them
    const alf_dev &A = alfs[pic];
    for (int i = 0; i < NM_ALF; ++i) {
      R->rate[MA + i] = k_ctx_init_alf[3][i];
      const int v = k_ctx_init_alf[slice][i];
      const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
      int s = ((slope * (init_qp - 16)) >> 1) + offset;
      s = s < 1 ? 1 : (s > 127 ? 127 : s);
      R->models[MA + i] = (uint32_t)((s << 8) & 0x7fe0) | ((uint32_t)((s << 8) & 0x7ffe) << 16);
    }
    for (int r = 0; r < cy; ++r) alf_ctu<false>(c, R, A, r * wc, wc, wc * hc);
  }
  const int max_off = (1 << ((bitdepth < 10 ? bitdepth : 10) - 5)) - 1;
  for (int cx = 0; cx < wc; ++cx) {
    const int k = cy * wc + cx, x0 = cx * 64, y0 = cy * 64;
    const int16_t *co = D.coeff + (size_t)k * 6144;
    if (wait_flags) { if (lane0) wait_final(wait_flags + (size_t)pic * wc * hc + k); }
    __syncthreads();
    for (int e = threadIdx.x; e < 17 * 17; e += blockDim.x) {       // the CTU's 16 x 16 units and one unit of border (left, above, corner)
      const int ux = e % 17 - 1, uy = e / 17 - 1, x = x0 + ux * 4, y = y0 + uy * 4;
      row_state::cui q = {};
      if (x >= 0 && y >= 0 && x < W && y < H) {
        const uvghip_scu_t *g = &D.cu[(y >> 2) * D.cu_stride + (x >> 2)];
        q.type = g->type; q.log2w = g->log2_width; q.cbf = g->cbf; q.mode = (uint8_t)(g->mv[0][0] & 0xff); q.mode_c = (uint8_t)((g->mv[0][0] >> 8) & 0xff);
        if (PB) {
          const uvghip_inter4_t t = PB->inter4[(y >> 2) * D.cu_stride + (x >> 2)];
          q.skipped = t.skipped;
          icand::unit &u = Q->tab[e];
          u.type = g->type; u.dir = g->mv_dir; u.mv[0][0] = g->mv[0][0]; u.mv[0][1] = g->mv[0][1]; u.mv[1][0] = g->mv[1][0]; u.mv[1][1] = g->mv[1][1];
          u.ref[0] = t.mv_ref0; u.ref[1] = t.mv_ref1;
          uint8_t *fl = Q->flags[e];
          fl[0] = t.skipped; fl[1] = t.merged; fl[2] = t.merge_idx; fl[3] = t.root_cbf; fl[4] = t.mv_cand0; fl[5] = t.mv_cand1; fl[6] = t.mv_ref0; fl[7] = t.mv_ref1;
        }
      } else if (PB) {
        icand::unit &u = Q->tab[e];
        u.type = 0; u.dir = 0; u.mv[0][0] = u.mv[0][1] = u.mv[1][0] = u.mv[1][1] = 0; u.ref[0] = u.ref[1] = 0;
      }
      R->cu[e] = q;
    }
    if (PB && lane0) Q->tab[17 * 17].type = 0;                      // the CTU above right: not a candidate under WPP
    __syncthreads();
    if (sao) {
      if (threadIdx.x < 34) R->sao_l[threadIdx.x] = __hip_atomic_load(sao + ((size_t)pic * wc * hc + k) * 34 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
    if (lane0 && sao) {                                             // encode_sao
      const int32_t *l = R->sao_l, *ch = l + 17;
      if (cx > 0) enc_bin(c, R, NMODELS, l[3]);
      if (cy > 0 && !l[3]) enc_bin(c, R, NMODELS, l[4]);
      if (!l[3] && !l[4]) { code_sao_color(c, R, l, 0, max_off); code_sao_color(c, R, ch, 1, max_off); code_sao_color(c, R, ch, 2, max_off); }
    }
    if (lane0 && alfs) alf_ctu<true>(c, R, alfs[pic], k, wc, wc * hc);
    // uvg_encode_coding_tree: z-order over the 4x4 units; a CU starts where a unit is aligned to its CU's size
    for (int z = 0; z < 256; ++z) {
      const int lx = z_to_xy(z) * 4, ly = z_to_xy(z >> 1) * 4, x = x0 + lx, y = y0 + ly;
      if (x >= W || y >= H) continue;                               // (uniform: every lane takes the same path)
      const row_state::cui cu = cu_of(R, x0, y0, x, y);
      const int n = 1 << cu.log2w;
      if ((lx & (n - 1)) || (ly & (n - 1))) continue;
      const int mode = cu.mode, mode_c = cu.mode_c;
      const int sep = n == 4, last4 = sep && (lx & 4) && (ly & 4);
      if (lane0) {
        for (int d = 0; (64 >> d) > n; ++d) {
          const int s = 64 >> d;
          if (!(lx & (s - 1)) && !(ly & (s - 1))) code_split_flag(c, R, x0, y0, W, H, x, y, s, 1);
        }
        code_split_flag(c, R, x0, y0, W, H, x, y, n, 0);
      }
      if (PB) {
        // ---- P / B slice: skip flag, prediction mode, the inter prediction unit (encode_coding_tree.c:1470-1640) ----
        const int e = ((ly >> 2) + 1) * 17 + (lx >> 2) + 1;
        const uint8_t *fl = Q->flags[e];
        const int skipped = fl[0], merged = fl[1], merge_idx = fl[2];
        const bool inter = cu.type == 2;
        if (lane0 && n != 4) {
          int ctx_skip = 0;
          if (x > 0 && cu_of(R, x0, y0, x - 4, y).skipped) ctx_skip++;
          if (y > 0 && cu_of(R, x0, y0, x, y - 4).skipped) ctx_skip++;
          enc_bin(c, R, MI_SKIP + ctx_skip, skipped);
        }
        if (skipped) {
          if (lane0) {
            hmvp_push(Q->hmvp, Q->tab[e]);
            for (int ui = 0; ui < PB->max_merge - 1; ui++) {
              const int symbol = ui != merge_idx;
              if (ui == 0) enc_bin(c, R, MI_MERGE_IDX, symbol); else enc_ep(c, symbol);
              if (!symbol) break;
            }
          }
          continue;
        }
        if (lane0 && n != 4) {
          const int ctx_pm = (x > 0 && cu_of(R, x0, y0, x - 4, y).type == 1) || (y > 0 && cu_of(R, x0, y0, x, y - 4).type == 1);
          enc_bin(c, R, MI_PRED_MODE + ctx_pm, cu.type == 1);
        }
        if (inter) {
          const icand::unit &me = Q->tab[e];
          if (lane0) {
            // uvg_encode_inter_prediction_unit (:769-900)
            enc_bin(c, R, MI_MERGE_FLAG, merged);
            if (merged) {
              for (int ui = 0; ui < PB->max_merge - 1; ui++) {
                const int symbol = ui != merge_idx;
                if (ui == 0) enc_bin(c, R, MI_MERGE_IDX, symbol); else enc_ep(c, symbol);
                if (!symbol) break;
              }
            } else {
              if (PB->slice_type == 0) {
                if (n + n > 12) enc_bin(c, R, MI_INTER_DIR + (7 - ((2 * ilog2_dev(n) + 1) >> 1)), me.dir == 3);
                if (me.dir < 3) enc_bin(c, R, MI_INTER_DIR + 5, me.dir == 2);
              }
              for (int l = 0; l < 2; ++l) {
                if (!(me.dir & (1 << l))) continue;
                const int lsz = PB->l_size[l], ref_frame = me.ref[l];
                if (lsz > 1) {
                  enc_bin(c, R, MI_REF_PIC + 0, ref_frame != 0);
                  if (ref_frame > 0 && lsz > 2) {
                    enc_bin(c, R, MI_REF_PIC + 1, ref_frame > 1);
                    if (ref_frame > 1 && lsz > 3)
                      for (int idx = 3; idx < lsz; idx++) { const int val = ref_frame > idx - 1; enc_ep(c, val); if (!val) break; }
                  }
                }
                // the predictor the CU chose: uvg_inter_get_mv_cand_cua on the picture's side information (a scratch copy of the CU's
                // context: the derivation clears unused lists of the neighbours it looks at, which the cu array version does not)
                icand::frame_ctx &f = Q->f;
                f.x = x; f.y = y; f.w = n; f.h = n;
                uint32_t tree = 0;                                    // the split tree down to this CU: quad splits
                for (int d = 0; (64 >> d) > n; ++d) tree |= 1u << (3 * d);
                f.split_tree = tree;
                lds_tab tab{Q->tab};
                glb_col colp{PB->col, PB->col_stride, (W + 7) / 8};
                int32_t *pred = Q->pred;
                Q->ref_idx[0] = me.ref[0]; Q->ref_idx[1] = me.ref[1];
                icand::amvp_candidates(f, tab, colp, Q->hmvp, l, Q->ref_idx, pred, &Q->ws);
                const int which = l == 0 ? fl[4] : fl[5];
                int dh = me.mv[l][0] - pred[2 * which], dv = me.mv[l][1] - pred[2 * which + 1];
                dh = dh >= 0 ? (dh + 1) >> 2 : (dh + 2) >> 2;         // uvg_change_precision(INTERNAL_MV_PREC, 2)
                dv = dv >= 0 ? (dv + 1) >> 2 : (dv + 2) >> 2;
                code_mvd(c, R, dh, dv);
                enc_bin(c, R, MI_MVP_IDX, which);
              }
            }
            hmvp_push(Q->hmvp, me);
          }
          const int first_cbf = cu.cbf;
          const int has_coeffs = fl[3] || first_cbf;
          if (lane0 && !merged) enc_bin(c, R, M_ROOT_CBF_, has_coeffs);
          if (has_coeffs) {
            const int tus = n == 64 ? 4 : 1, tn = n == 64 ? 32 : n;
            for (int tu = 0; tu < tus; ++tu) {
              const int tlx = lx + (tu & 1) * 32, tly = ly + (tu >> 1) * 32;
              const int tcbf = cu_of(R, x0, y0, x0 + tlx, y0 + tly).cbf;
              const int cb_y = tcbf & 1, cb_u = (tcbf >> 1) & 1, cb_v = (tcbf >> 2) & 1;
              if (lane0) {
                enc_bin(c, R, M_CBF_CB, cb_u); enc_bin(c, R, M_CBF_CR + cb_u, cb_v);
                if (n == 64 || cb_u || cb_v) enc_bin(c, R, M_CBF_LUMA, cb_y);       // otherwise inferred 1 (:705-716)
              }
              if (cb_y) { stage(R, co + tly * 64 + tlx, 64, tn, 0); if (lane0) code_coeffs(c, R, tn, 0); }
              if (cb_u) { stage(R, co + 4096 + (tly >> 1) * 32 + (tlx >> 1), 32, tn >> 1, 1); if (lane0) code_coeffs(c, R, tn >> 1, 1); }
              if (cb_v) { stage(R, co + 5120 + (tly >> 1) * 32 + (tlx >> 1), 32, tn >> 1, 2); if (lane0) code_coeffs(c, R, tn >> 1, 2); }
            }
          }
          continue;
        }
      }
      if (lane0) {
        code_luma_mode(c, R, x0, y0, x, y, n, mode);
        if (!sep) code_chroma_mode(c, R, mode_c, mode);
      }
      const int tus = n == 64 ? 4 : 1, tn = n == 64 ? 32 : n;
      for (int tu = 0; tu < tus; ++tu) {
        const int tlx = lx + (tu & 1) * 32, tly = ly + (tu >> 1) * 32;
        const int tcbf = cu_of(R, x0, y0, x0 + tlx, y0 + tly).cbf;
        const int cb_y = tcbf & 1, cb_u = (tcbf >> 1) & 1, cb_v = (tcbf >> 2) & 1;
        if (lane0) {
          if (!sep) { enc_bin(c, R, M_CBF_CB, cb_u); enc_bin(c, R, M_CBF_CR + cb_u, cb_v); }
          enc_bin(c, R, M_CBF_LUMA, cb_y);
        }
        if (cb_y) { stage(R, co + tly * 64 + tlx, 64, tn, 0); if (lane0) code_coeffs(c, R, tn, 0); }
        if (!sep) {
          if (cb_u) { stage(R, co + 4096 + (tly >> 1) * 32 + (tlx >> 1), 32, tn >> 1, 1); if (lane0) code_coeffs(c, R, tn >> 1, 1); }
          if (cb_v) { stage(R, co + 5120 + (tly >> 1) * 32 + (tlx >> 1), 32, tn >> 1, 2); if (lane0) code_coeffs(c, R, tn >> 1, 2); }
        } else if (last4) {
          // the 8x8 area's chroma after its last luma CU: mode (the co-located luma CU is this one), cbfs of the area's first entry, levels
          const int acbf = cu_of(R, x0, y0, x & ~7, y & ~7).cbf;
          const int au = (acbf >> 1) & 1, av = (acbf >> 2) & 1;
          if (lane0) { code_chroma_mode(c, R, mode_c, mode); enc_bin(c, R, M_CBF_CB, au); enc_bin(c, R, M_CBF_CR + au, av); }
          const int cbx = (lx & ~7) >> 1, cby = (ly & ~7) >> 1;
          if (au) { stage(R, co + 4096 + cby * 32 + cbx, 32, 4, 1); if (lane0) code_coeffs(c, R, 4, 1); }
          if (av) { stage(R, co + 5120 + cby * 32 + cbx, 32, 4, 2); if (lane0) code_coeffs(c, R, 4, 2); }
        }
      }
    }
  }
  if (lane0) {
    // end_of_sub_stream_one_bit (uvg_cabac_encode_bin_trm(1)), uvg_cabac_finish, a one bit, zeros to the byte boundary
    c.range -= 2;
    c.low += c.range;
    c.low <<= 7;
    c.range = 2 << 7;
    c.bits_left -= 7;
    if (c.bits_left < 12) cwrite(c);
    if (c.low >> (32 - c.bits_left)) {
      put_byte(c, c.buffered + 1);
      while (c.nbuf > 1) { put_byte(c, 0); c.nbuf--; }
      c.low -= 1u << (32 - c.bits_left);
    } else {
      if (c.nbuf > 0) put_byte(c, c.buffered);
      while (c.nbuf > 1) { put_byte(c, 0xff); c.nbuf--; }
    }
    uint32_t acc = 0;
    int nacc = 0;
    const int nb = 24 - c.bits_left;
    const uint32_t v = c.low >> 8;
    for (int i = nb - 1; i >= 0; --i) { acc = (acc << 1) | ((v >> i) & 1u); if (++nacc == 8) { put_byte(c, acc); acc = 0; nacc = 0; } }
    acc = (acc << 1) | 1u; if (++nacc == 8) { put_byte(c, acc); acc = 0; nacc = 0; }
    if (nacc) put_byte(c, acc << (8 - nacc));
    row_bytes[(size_t)pic * hc + cy] = c.n;            // (> row_cap: the buffer was too small, the row is truncated)
  }
  if (!PERSIST) break;
  __syncthreads();          // the row's state and s_job are free again
  }
}

}  // namespace

extern "C" size_t uvghip_slice_rows_workspace_bytes(int n_pictures) { return n_pictures > 0 ? (size_t)n_pictures * sizeof(pic_dev) : 0; }
extern "C" size_t uvghip_slice_rows_alf_workspace_bytes(int n_pictures)
{
  return n_pictures > 0 ? ((size_t)n_pictures * sizeof(pic_dev) + 255) / 256 * 256 + (size_t)n_pictures * sizeof(alf_dev) : 0;
}
extern "C" size_t uvghip_slice_rows_pb_workspace_bytes(int n_pictures)
{
  return n_pictures > 0 ? ((size_t)n_pictures * sizeof(pic_dev) + 255) / 256 * 256 + (size_t)n_pictures * sizeof(pb_dev) : 0;
}

// the picture table the kernel reads, uploaded once (synchronously); uvghip_encode_slice_rows with pictures == NULL reuses it
extern "C" int uvghip_slice_rows_prepare(const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, int n_pictures, void *workspace)
{
  UVGHIP_REQUIRE_READY();
  if (!params || !pictures || n_pictures <= 0 || !workspace) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int wc = (params->pic_w + 63) / 64;
  std::vector<pic_dev> pd(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    if (!pictures[i].cu || !pictures[i].coeff || !pictures[i].models || pictures[i].cu_stride < wc * 16)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_slice_rows_prepare: picture descriptor");
    pd[i] = pic_dev{pictures[i].cu, pictures[i].coeff, pictures[i].models, pictures[i].cu_stride, 0};
  }
  UVGHIP_TRY(hipMemcpy(workspace, pd.data(), pd.size() * sizeof(pic_dev), hipMemcpyHostToDevice));
  return 0;
}

// ... the same table in stream order (the callers that take a stream: nothing waits for it)
static int prepare_ordered(const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, int n_pictures, void *workspace, hipStream_t st)
{
  const int wc = (params->pic_w + 63) / 64;
  std::vector<pic_dev> pd(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    if (!pictures[i].cu || !pictures[i].coeff || !pictures[i].models || pictures[i].cu_stride < wc * 16)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_encode_slice_rows: picture descriptor");
    pd[i] = pic_dev{pictures[i].cu, pictures[i].coeff, pictures[i].models, pictures[i].cu_stride, 0};
  }
  return uvghip_upload_ordered(workspace, pd.data(), pd.size() * sizeof(pic_dev), st);
}

extern "C" int uvghip_encode_slice_rows(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, int n_pictures,
                                        const int32_t *sao_info, const uint16_t *sao_models, void *workspace, uint8_t *out, int row_cap,
                                        int32_t *row_bytes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || n_pictures <= 0 || !workspace || !out || row_cap <= 0 || !row_bytes || (sao_info && !sao_models))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int W = params->pic_w, H = params->pic_h, hc = (H + 63) / 64;
  if (W <= 0 || H <= 0 || (W & 7) || (H & 7) || params->qp < 0 || params->qp > 63) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (pictures)
    if (int rc = uvghip_slice_rows_prepare(params, pictures, n_pictures, workspace)) return rc;
  hipStream_t st = uvghip_stream(stream);
  slice_rows_kernel<false><<<n_pictures * hc, 64, 0, st>>>(static_cast<const pic_dev *>(workspace), nullptr, sao_info, sao_models, W, H, params->qp, bitdepth, out,
                                                    row_cap, row_bytes);
  UVGHIP_CHECK_LAUNCH();
}

// ... of pictures whose search is complete but whose in-loop filters (the SAO decisions the slice data carries) are being finished by ANOTHER
// launch beside this one -- I pictures in the flight (uvghip_loop_pb_run_inflight_ext filters them CTU by CTU): final_flags are that
// launch's per-CTU flags of these pictures ([picture][ctu], uvghip_loop_pb_inflight_final_flags), a row waits for each of its CTUs (and
// for the first CTU of the row above: its SAO models).  The caller orders the flags' zeroing before this call's stream position.
extern "C" int uvghip_encode_slice_rows_behind(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, int n_pictures,
                                               const int32_t *sao_info, const uint16_t *sao_models, const int32_t *final_flags, void *workspace, uint8_t *out,
                                               int row_cap, int32_t *row_bytes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || n_pictures <= 0 || !workspace || !out || row_cap <= 0 || !row_bytes || !sao_info || !sao_models || !final_flags)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int W = params->pic_w, H = params->pic_h, hc = (H + 63) / 64;
  if (W <= 0 || H <= 0 || (W & 7) || (H & 7) || params->qp < 0 || params->qp > 63) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  if (pictures)
    if (int rc = prepare_ordered(params, pictures, n_pictures, workspace, st)) return rc;
  slice_rows_kernel<false><<<n_pictures * hc, 64, 0, st>>>(static_cast<const pic_dev *>(workspace), nullptr, sao_info, sao_models, W, H, params->qp, bitdepth, out,
                                                    row_cap, row_bytes, nullptr, final_flags);
  UVGHIP_CHECK_LAUNCH();
}

// ... with at most max_waves rows in progress (persistent waves that take row r of every picture, then row r + 1, from `ticket`: one int32 of
// DEVICE memory the caller zeroes in stream order before the launch): behind a search that is still RUNNING a waiting row must not hold what the
// search needs -- a wave per row of a whole clip does (uvghip_loop_plan_run_overlapped).
extern "C" int uvghip_encode_slice_rows_behind_capped(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, int n_pictures,
                                                      const int32_t *sao_info, const uint16_t *sao_models, const int32_t *final_flags, int32_t *ticket, int max_waves,
                                                      void *workspace, uint8_t *out, int row_cap, int32_t *row_bytes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || n_pictures <= 0 || !workspace || !out || row_cap <= 0 || !row_bytes || !sao_info || !sao_models || !final_flags || !ticket || max_waves < 1)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int W = params->pic_w, H = params->pic_h, hc = (H + 63) / 64;
  if (W <= 0 || H <= 0 || (W & 7) || (H & 7) || params->qp < 0 || params->qp > 63) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  if (pictures)
    if (int rc = prepare_ordered(params, pictures, n_pictures, workspace, st)) return rc;
  const int rows = n_pictures * hc, grid = rows < max_waves ? rows : max_waves;
  slice_rows_kernel<true><<<grid, 64, 0, st>>>(static_cast<const pic_dev *>(workspace), nullptr, sao_info, sao_models, W, H, params->qp, bitdepth, out,
                                               row_cap, row_bytes, nullptr, final_flags, ticket, n_pictures);
  UVGHIP_CHECK_LAUNCH();
}

// ... of pictures of an --alf on / --alf full run: the CTU-level ALF syntax between a CTU's SAO syntax and its coding tree.
extern "C" int uvghip_encode_slice_rows_alf(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, const uvghip_slice_alf_t *alf, int n_pictures,
                                            const int32_t *sao_info, const uint16_t *sao_models, void *workspace, uint8_t *out, int row_cap, int32_t *row_bytes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || !pictures || !alf || n_pictures <= 0 || !workspace || !out || row_cap <= 0 || !row_bytes || (sao_info && !sao_models))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int W = params->pic_w, H = params->pic_h, hc = (H + 63) / 64;
  if (W <= 0 || H <= 0 || (W & 7) || (H & 7) || params->qp < 0 || params->qp > 63) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  if (int rc = prepare_ordered(params, pictures, n_pictures, workspace, st)) return rc;
  std::vector<alf_dev> ad(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_slice_alf_t &q = alf[i];
    if ((q.alf_type != 1 && q.alf_type != 2) || q.n_luma_aps < 0 || q.n_luma_aps > 8 || q.n_alternatives_chroma < 0 || q.n_alternatives_chroma > 8 || q.cc_filter_count[0] > 4 ||
        q.cc_filter_count[1] > 4 || ((q.enabled[0] || q.cc_enabled[0] || q.cc_enabled[1]) && (!q.ctu_flags || !q.filter_set_idx)))
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_encode_slice_rows_alf: slice descriptor");
    alf_dev &d = ad[i];
    d.alf_type = q.alf_type; d.n_luma_aps = q.n_luma_aps; d.n_alts = q.n_alternatives_chroma; d.flags = q.ctu_flags; d.set_idx = q.filter_set_idx;
    for (int c = 0; c < 3; ++c) d.enabled[c] = q.enabled[c] != 0;
    for (int c = 0; c < 2; ++c) { d.cc_enabled[c] = q.cc_enabled[c] != 0; d.cc_count[c] = q.cc_filter_count[c]; }
  }
  unsigned char *aw = static_cast<unsigned char *>(workspace) + ((size_t)n_pictures * sizeof(pic_dev) + 255) / 256 * 256;
  if (int rc = uvghip_upload_ordered(aw, ad.data(), ad.size() * sizeof(alf_dev), st)) return rc;
  slice_rows_kernel<false><<<n_pictures * hc, 64, 0, st>>>(static_cast<const pic_dev *>(workspace), nullptr, sao_info, sao_models, W, H, params->qp, bitdepth, out, row_cap, row_bytes,
                                                    reinterpret_cast<const alf_dev *>(aw));
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_encode_slice_rows_pb(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, const uvghip_slice_pb_t *pb,
                                           int n_pictures, const int32_t *sao_info, const uint16_t *sao_models, void *workspace, uint8_t *out, int row_cap,
                                           int32_t *row_bytes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || !pictures || !pb || n_pictures <= 0 || !workspace || !out || row_cap <= 0 || !row_bytes || (sao_info && !sao_models))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int W = params->pic_w, H = params->pic_h, hc = (H + 63) / 64;
  if (W <= 0 || H <= 0 || (W & 7) || (H & 7) || params->qp < 0 || params->qp > 63) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  if (int rc = prepare_ordered(params, pictures, n_pictures, workspace, st)) return rc;
  std::vector<pb_dev> pd(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_slice_pb_t &q = pb[i];
    if ((q.slice_type != 0 && q.slice_type != 1) || !q.inter4 || !q.models_inter || q.n_refs < 1 || q.n_refs > 16 || q.l_size[0] < 1 || q.l_size[0] > 8 ||
        q.l_size[1] < 0 || q.l_size[1] > 8 || q.max_merge < 1 || q.max_merge > 6 || (q.tmvp && !q.col) || q.frame_qp < 0 || q.frame_qp > 63)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_encode_slice_rows_pb: slice descriptor");
    pb_dev &d = pd[i];
    d.slice_type = q.slice_type; d.poc = q.poc; d.n_refs = q.n_refs; d.tmvp = q.tmvp; d.max_merge = q.max_merge; d.merge_level = q.merge_level; d.frame_qp = q.frame_qp;
    memcpy(d.ref_pocs, q.ref_pocs, sizeof d.ref_pocs); memcpy(d.l_size, q.l_size, sizeof d.l_size); memcpy(d.l, q.l, sizeof d.l);
    d.col = q.col; d.inter4 = q.inter4; d.models_inter = q.models_inter; d.col_stride = q.col_stride; d.pad = 0;
  }
  unsigned char *pbw = static_cast<unsigned char *>(workspace) + ((size_t)n_pictures * sizeof(pic_dev) + 255) / 256 * 256;
  if (int rc = uvghip_upload_ordered(pbw, pd.data(), pd.size() * sizeof(pb_dev), st)) return rc;
  slice_rows_kernel<false><<<n_pictures * hc, 64, sizeof(row_state_pb), st>>>(static_cast<const pic_dev *>(workspace), reinterpret_cast<const pb_dev *>(pbw), sao_info, sao_models,
                                                                       W, H, params->qp, bitdepth, out, row_cap, row_bytes);
  UVGHIP_CHECK_LAUNCH();
}
