// The SAO decision's device functions (src/sao.c:52-603, encode_sao's model adaptation src/encoderstate.c:523-608): shared by the
// picture-level kernels (sao_decide.hip, where the citations are) and the per-CTU filter job of the P / B pipeline (ctu_filter.h).
#pragma once
#include "uvghip_common.h"
#include "vvc_rdoq_tables.h"
#include "vvc_ctx_init.h"
#include <climits>

namespace saod {

struct cand {                 // model-independent part of sao_search_edge_sao / sao_search_band_sao for one (CTU, colour group)
  int32_t edge_dd[4];         // sum over buffers and categories of cnt * o^2 - 2 * o * sum, per class
  int32_t edge_bits[4];       // bits of the offsets, per class (sao_mode_bits_edge without the model-coded bins)
  int8_t edge_off[4][10];     // offsets per class [buffer * 5 + category]
  int32_t band_dd, band_bits;
  int8_t band_off[10];        // [buffer * 5 + 1 + k]
  int8_t band_pos[2];
};

__device__ inline int clampo(int v, int m) { return v < -m ? -m : (v > m ? m : v); }

// the model-independent candidates of one (CTU, colour group): E / B = the statistics of its one (luma) or two (Cb, Cr) buffers
__device__ inline void sao_candidates_one(const int32_t *const *E, const int32_t *const *B, int bufs, int omax, cand &c)
{
  for (int cls = 0; cls < 4; ++cls) {
    int dd = 0, bits = 0;
    for (int i = 0; i < 10; ++i) c.edge_off[cls][i] = 0;
    for (int b = 0; b < bufs; ++b)
      for (int cat = 1; cat <= 4; ++cat) {
        const int cs = E[b][cls * 10 + cat], cc = E[b][cls * 10 + 5 + cat];
        int o = 0;
        if (cc != 0) o = clampo((cs + (cc >> 1)) / cc, omax);
        if (cat <= 2 && o < 0) o = 0;          // sharpening offsets cannot be coded (sao.c:406-411)
        if (cat >= 3 && o > 0) o = 0;
        c.edge_off[cls][b * 5 + cat] = (int8_t)o;
        dd += cc * o * o - 2 * o * cs;
        const int a = o < 0 ? -o : o;
        bits += (a == 0 || a == omax) ? a + 1 : a + 2;
      }
    c.edge_dd[cls] = dd; c.edge_bits[cls] = bits;
  }
  // calc_sao_band_offsets (sao.c:208-262) per buffer
  int bdd = 0, bbits = 0;
  for (int i = 0; i < 10; ++i) c.band_off[i] = 0;
  c.band_pos[0] = c.band_pos[1] = 0;
  for (int b = 0; b < bufs; ++b) {
    int best = INT_MAX, best_pos = 0;
    int d0 = 0, d1 = 0, d2 = 0;          // dist of the three bands before `band`
    for (int band = 0; band < 32; ++band) {
      const int s = B[b][band], cnt = B[b][32 + band];
      int o = 0;
      if (cnt != 0) o = clampo((s + (cnt >> 1)) / cnt, omax);
      // the reference steps the offset towards 0 and keeps the LAST step's values (its best_dist is never lowered, :233-241): +-1
      const int of = o > 0 ? 1 : (o < 0 ? -1 : 0);
      const int dist = of ? cnt * of * of - 2 * of * s : 0;
      if (band >= 3 && band < 31) {          // starting positions 0..27 only (sao.c:248: band < 28), although 28 would fit
        const int tot = (int)((unsigned)d0 + (unsigned)d1 + (unsigned)d2 + (unsigned)dist);
        if (tot < best) { best = tot; best_pos = band - 3; }
      }
      d0 = d1; d1 = d2; d2 = dist;
    }
    c.band_pos[b] = (int8_t)best_pos;
    for (int q = 0; q < 4; ++q) {
      const int s = B[b][best_pos + q], cnt = B[b][32 + best_pos + q];
      const int o0 = cnt != 0 ? clampo((s + (cnt >> 1)) / cnt, omax) : 0;
      const int o = o0 > 0 ? 1 : (o0 < 0 ? -1 : 0);
      c.band_off[b * 5 + 1 + q] = (int8_t)o;
      const int a = o < 0 ? -o : o;
      bbits += a == 0 ? 1 : (a == omax ? a + 2 : a + 3);
    }
    bdd += best;
  }
  c.band_dd = bdd; c.band_bits = bbits + 5 * bufs;
}

struct sao_info { int32_t type, eo_class, ddistortion, merge_left, merge_up, band_position[2], offsets[10]; };   // sao_info_t (src/sao.h:55-63)

struct models2 { uint16_t s0[2], s1[2]; uint8_t rate[2]; };       // [0] sao_merge_flag_model, [1] sao_type_idx_model

__device__ inline double fbits(const models2 &m, int c, int bin)
{
  return (double)kEntropyBits[((((int)m.s0[c] + m.s1[c]) >> 8) << 1) ^ bin] / 32768.0;      // uvg_f_entropy_bits (rdo.c:143)
}
__device__ inline void code_bin(models2 &m, int c, int bin)          // CTX_UPDATE (cabac.h:182-193)
{
  const int r0 = m.rate[c] >> 4, r1 = m.rate[c] & 15;
  uint32_t a = m.s0[c], b = m.s1[c];
  a -= (a >> r0) & 0x7fe0u;
  b -= (b >> r1) & 0x7ffeu;
  if (bin) { a += (0x7fffu >> r0) & 0x7fe0u; b += (0x7fffu >> r1) & 0x7ffeu; }
  m.s0[c] = (uint16_t)a; m.s1[c] = (uint16_t)b;
}

// the distortion change a decision (its class / offsets or band position / offsets) makes on THIS CTU's statistics
__device__ int apply_dd(const sao_info &s, const int32_t *const *E, const int32_t *const *B, int bufs)
{
  int dd = 0;
  if (s.type == 2) {
    for (int b = 0; b < bufs; ++b)
      for (int cat = 0; cat < 5; ++cat) {
        const int o = s.offsets[5 * b + cat];
        dd += E[b][s.eo_class * 10 + 5 + cat] * o * o - 2 * o * E[b][s.eo_class * 10 + cat];
      }
  } else if (s.type == 1) {
    for (int b = 0; b < bufs; ++b)
      for (int q = 0; q < 4; ++q) {
        const int o = s.offsets[5 * b + 1 + q], band = s.band_position[b] + q;
        if (band < 32) dd += B[b][32 + band] * o * o - 2 * o * B[b][band];
      }
  }
  return dd;
}

// sao_search_best_mode for one colour group
__device__ void best_mode(const models2 &m, double lambda, int sao_type, const cand &c, const int32_t *const *E, const int32_t *const *B, int bufs,
                          sao_info &out, const sao_info *top, const sao_info *left, int32_t merge_cost[3])
{
  double prefix = 0.0;                                       // the merge flags a non-merged CTU codes as 0
  if (left) prefix += fbits(m, 0, 0);
  if (top) prefix += fbits(m, 0, 0);
  const double typed = prefix + fbits(m, 1, 1) + 1.0;        // sao_type_idx: first bin with the model, second bypass
  sao_info edge = {}, band = {};
  if (sao_type & 1) {
    edge.type = 2; edge.ddistortion = INT_MAX;
    for (int cls = 0; cls < 4; ++cls) {
      const int sum = c.edge_dd[cls] + (int)((typed + c.edge_bits[cls] + 2.0) * lambda + 0.5);
      if (sum < edge.ddistortion) {
        edge.eo_class = cls; edge.ddistortion = sum;
        for (int i = 0; i < 10; ++i) edge.offsets[i] = c.edge_off[cls][i];
      }
    }
  } else edge.ddistortion = INT_MAX;        // (with the class's distortion taken from the statistics the second pass of :507-519 gives the same number)
  if (sao_type & 2) {
    band.type = 1;
    band.band_position[0] = c.band_pos[0]; band.band_position[1] = c.band_pos[1];
    for (int i = 0; i < 10; ++i) band.offsets[i] = c.band_off[i];
    band.ddistortion = c.band_dd + (int)((typed + c.band_bits) * lambda + 0.5);
  } else band.ddistortion = INT_MAX;
  if (edge.ddistortion <= band.ddistortion) { out = edge; merge_cost[0] = edge.ddistortion; }
  else { out = band; merge_cost[0] = band.ddistortion; }
  {
    const int nothing = (int)((prefix + fbits(m, 1, 0)) * lambda + 0.5);
    if (out.ddistortion >= nothing) { out.type = 0; merge_cost[0] = nothing; }
  }
  const sao_info *cands[2] = {left, top};
  for (int i = 0; i < 2; ++i) {
    if (!cands[i]) continue;
    double b = fbits(m, 0, i == 0);                          // merge left: one bin; merge up: 0 then 1
    if (i == 1) b += fbits(m, 0, 1);
    merge_cost[i + 1] = (int)(b * lambda + 0.5) + apply_dd(*cands[i], E, B, bufs);
  }
}


// models of a slice at its start (uvg_init_contexts for the slice type, context.c:471-500)
__device__ inline void sao_models_init(models2 &m, int slice_type, int qp)
{
  for (int i = 0; i < 2; ++i) {
    const int v = k_ctx_init_sao[slice_type][i];
    const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
    int s = ((slope * (qp - 16)) >> 1) + offset;
    s = s < 1 ? 1 : (s > 127 ? 127 : s);
    m.s0[i] = (uint16_t)((s << 8) & 0x7fe0); m.s1[i] = (uint16_t)((s << 8) & 0x7ffe);
    m.rate[i] = k_ctx_init_sao[3][i];
  }
}

// One CTU of the chain: m = the models as the coder left them after the previous CTU (in: at this CTU's start, out: after its SAO
// syntax); L / C = this CTU's luma / chroma decision (out); top / left = the neighbours' (NULL at the picture's edge).
__device__ inline void sao_decide_one(models2 &m, double lambda, int sao_type, const cand &cl, const cand &cc, const int32_t *ey, const int32_t *by,
                                      const int32_t *eu, const int32_t *bu, const int32_t *ev, const int32_t *bv, sao_info &L, sao_info &C,
                                      const sao_info *top_l, const sao_info *top_c, const sao_info *left_l, const sao_info *left_c)
{
  int32_t mc_l[3] = {INT_MAX, 0, 0}, mc_c[3] = {INT_MAX, 0, 0};
  {
    const int32_t *E[2] = {ey, nullptr}, *B[2] = {by, nullptr};
    best_mode(m, lambda, sao_type, cl, E, B, 1, L, top_l, left_l, mc_l);
  }
  {
    const int32_t *E[2] = {eu, ev}, *B[2] = {bu, bv};
    best_mode(m, lambda, sao_type, cc, E, B, 2, C, top_c, left_c, mc_c);
  }
  L.merge_up = L.merge_left = 0;
  if (top_l && mc_l[2] + mc_c[2] <= mc_l[0] + mc_c[0]) { L = *top_l; C = *top_c; L.merge_up = 1; L.merge_left = 0; }
  if (left_l && mc_l[1] + mc_c[1] <= mc_l[0] + mc_c[0] && (!L.merge_up || mc_l[1] + mc_c[1] < mc_l[2] + mc_c[2])) {
    L = *left_l; C = *left_c; L.merge_left = 1; L.merge_up = 0;
  }
  // encode_sao: the bins with a model
  if (left_l) code_bin(m, 0, L.merge_left);
  if (top_l && !L.merge_left) code_bin(m, 0, L.merge_up);
  if (!L.merge_left && !L.merge_up) { code_bin(m, 1, L.type != 0); code_bin(m, 1, C.type != 0); }
}

}  // namespace saod
