// uvghip_sao_decide_pictures: the SAO decision of every CTU of a group of all-intra pictures, from the statistics
// uvghip_sao_stats_batch took on the picture uvghip_deblock_frame_sao_snapshot produced.
// replaces: uvg_sao_search_lcu (src/sao.c:670-742) with sao_search_best_mode (:490-603), sao_search_edge_sao / _band_sao
// (:362-488), calc_sao_band_offsets (:208-262), the bit estimates sao_mode_bits_* (:52-178) and, for the two context models
// those estimates read, the adaptation of encode_sao (src/encoderstate.c:523-608) from CTU to CTU.
//
// Everything the reference computes from samples here is an exact function of the statistics: uvg_sao_edge_ddistortion and
// uvg_sao_band_ddistortion are  sum cnt * o^2 - 2 * o * sum  over the categories / bands an offset applies to
// (strategies/generic/sao_shared_generics.h:53-127).  So the work splits into
//   candidates  (a thread per picture, CTU and colour group): per edge class the offsets, their distortion change and the bits
//               of the offsets; the band position / offsets of calc_sao_band_offsets -- nothing here depends on the neighbours;
//   decisions   (a thread per picture): CTUs in coding order, because a CTU's bit estimates read the two SAO models as the coder
//               leaves them after the previous CTU (state->search_cabac is a copy of state->cabac taken at the CTU's start and
//               its update flag is off during the decision), merge candidates are the left / upper CTU's decisions, and under
//               WPP a row starts from the models after the first CTU of the row above (encoderstate.c:966-975).
#include "uvghip_common.h"
#include "vvc_rdoq_tables.h"
#include "vvc_ctx_init.h"
#include <climits>

namespace {

struct cand {                 // model-independent part of sao_search_edge_sao / sao_search_band_sao for one (CTU, colour group)
  int32_t edge_dd[4];         // sum over buffers and categories of cnt * o^2 - 2 * o * sum, per class
  int32_t edge_bits[4];       // bits of the offsets, per class (sao_mode_bits_edge without the model-coded bins)
  int8_t edge_off[4][10];     // offsets per class [buffer * 5 + category]
  int32_t band_dd, band_bits;
  int8_t band_off[10];        // [buffer * 5 + 1 + k]
  int8_t band_pos[2];
};

__device__ inline int clampo(int v, int m) { return v < -m ? -m : (v > m ? m : v); }

__global__ void sao_candidates_kernel(const int32_t *__restrict__ edge_y, const int32_t *__restrict__ band_y, const int32_t *__restrict__ edge_u,
                                      const int32_t *__restrict__ band_u, const int32_t *__restrict__ edge_v, const int32_t *__restrict__ band_v,
                                      int n, int omax, cand *__restrict__ out)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n) return;
  const int k = t >> 1, grp = t & 1, bufs = grp ? 2 : 1;
  const int32_t *E[2] = {grp ? edge_u + (size_t)k * 40 : edge_y + (size_t)k * 40, edge_v + (size_t)k * 40};
  const int32_t *B[2] = {grp ? band_u + (size_t)k * 64 : band_y + (size_t)k * 64, band_v + (size_t)k * 64};
  cand &c = out[t];                    // written in place (a local copy indexed by class would live in scratch memory)
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

__global__ void sao_decide_kernel(const int32_t *__restrict__ edge_y, const int32_t *__restrict__ band_y, const int32_t *__restrict__ edge_u,
                                  const int32_t *__restrict__ band_u, const int32_t *__restrict__ edge_v, const int32_t *__restrict__ band_v,
                                  const cand *__restrict__ cands, int n_pictures, int wc, int hc, int qp, double lambda, int sao_type, int slice_type,
                                  int32_t *__restrict__ info_out, uint16_t *__restrict__ models_out, uvghip_sao_param_t *__restrict__ py,
                                  uvghip_sao_param_t *__restrict__ pu, uvghip_sao_param_t *__restrict__ pv)
{
  const int pic = blockIdx.x * blockDim.x + threadIdx.x;
  if (pic >= n_pictures) return;
  const int ctus = wc * hc;
  sao_info *luma = reinterpret_cast<sao_info *>(info_out) + (size_t)pic * ctus * 2;       // [ctu][2]: the output doubles as the state
  models2 row_start = {};                                                                 // models after the first CTU of the row above
  models2 m = {};
  for (int cy = 0; cy < hc; ++cy)
    for (int cx = 0; cx < wc; ++cx) {
      const int k = cy * wc + cx;
      const size_t g = (size_t)pic * ctus + k;
      if (cx == 0) {
        if (cy == 0) {
          for (int i = 0; i < 2; ++i) {         // uvg_init_contexts for the slice type (context.c:471-500)
            const int v = k_ctx_init_sao[slice_type][i];
            const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
            int s = ((slope * (qp - 16)) >> 1) + offset;
            s = s < 1 ? 1 : (s > 127 ? 127 : s);
            m.s0[i] = (uint16_t)((s << 8) & 0x7fe0); m.s1[i] = (uint16_t)((s << 8) & 0x7ffe);
            m.rate[i] = k_ctx_init_sao[3][i];
          }
        } else m = row_start;
      }
      sao_info &L = luma[2 * k], &C = luma[2 * k + 1];
      const sao_info *top_l = cy ? &luma[2 * (k - wc)] : nullptr, *left_l = cx ? &luma[2 * (k - 1)] : nullptr;
      const sao_info *top_c = cy ? &luma[2 * (k - wc) + 1] : nullptr, *left_c = cx ? &luma[2 * (k - 1) + 1] : nullptr;
      int32_t mc_l[3] = {INT_MAX, 0, 0}, mc_c[3] = {INT_MAX, 0, 0};
      {
        const int32_t *E[2] = {edge_y + g * 40, nullptr}, *B[2] = {band_y + g * 64, nullptr};
        best_mode(m, lambda, sao_type, cands[2 * g], E, B, 1, L, top_l, left_l, mc_l);
      }
      {
        const int32_t *E[2] = {edge_u + g * 40, edge_v + g * 40}, *B[2] = {band_u + g * 64, band_v + g * 64};
        best_mode(m, lambda, sao_type, cands[2 * g + 1], E, B, 2, C, top_c, left_c, mc_c);
      }
      L.merge_up = L.merge_left = 0;
      if (top_l && mc_l[2] + mc_c[2] <= mc_l[0] + mc_c[0]) { L = *top_l; C = *top_c; L.merge_up = 1; L.merge_left = 0; }
      if (left_l && mc_l[1] + mc_c[1] <= mc_l[0] + mc_c[0] && (!L.merge_up || mc_l[1] + mc_c[1] < mc_l[2] + mc_c[2])) {
        L = *left_l; C = *left_c; L.merge_left = 1; L.merge_up = 0;
      }
      // encode_sao: the bins with a model
      if (cx > 0) code_bin(m, 0, L.merge_left);
      if (cy > 0 && !L.merge_left) code_bin(m, 0, L.merge_up);
      if (!L.merge_left && !L.merge_up) { code_bin(m, 1, L.type != 0); code_bin(m, 1, C.type != 0); }
      if (cx == 0) row_start = m;
      uint16_t *mo = models_out + g * 6;
      mo[0] = m.s0[0]; mo[1] = m.s1[0]; mo[2] = m.rate[0]; mo[3] = m.s0[1]; mo[4] = m.s1[1]; mo[5] = m.rate[1];
      uvghip_sao_param_t a = {L.type, L.eo_class, L.band_position[0], {L.offsets[0], L.offsets[1], L.offsets[2], L.offsets[3], L.offsets[4]}};
      uvghip_sao_param_t b = {C.type, C.eo_class, C.band_position[0], {C.offsets[0], C.offsets[1], C.offsets[2], C.offsets[3], C.offsets[4]}};
      uvghip_sao_param_t d = {C.type, C.eo_class, C.band_position[1], {C.offsets[5], C.offsets[6], C.offsets[7], C.offsets[8], C.offsets[9]}};
      py[g] = a; pu[g] = b; pv[g] = d;
    }
}

}  // namespace

extern "C" size_t uvghip_sao_decide_workspace_bytes(int n_pictures, int pic_w, int pic_h)
{
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0) return 0;
  return (size_t)n_pictures * ((pic_w + 63) / 64) * ((pic_h + 63) / 64) * 2 * sizeof(cand);
}

extern "C" int uvghip_sao_decide_pictures(int bitdepth, int n_pictures, int pic_w, int pic_h, int qp, double lambda, int sao_type,
                                          const int32_t *edge_y, const int32_t *band_y, const int32_t *edge_u, const int32_t *band_u,
                                          const int32_t *edge_v, const int32_t *band_v, void *workspace, int32_t *info_out,
                                          uint16_t *models_out, uvghip_sao_param_t *params_y, uvghip_sao_param_t *params_u,
                                          uvghip_sao_param_t *params_v, void *stream)
{
  return uvghip_sao_decide_pictures_slice(bitdepth, n_pictures, pic_w, pic_h, qp, lambda, sao_type, 2, edge_y, band_y, edge_u, band_u, edge_v, band_v, workspace,
                                          info_out, models_out, params_y, params_u, params_v, stream);
}

extern "C" int uvghip_sao_decide_pictures_slice(int bitdepth, int n_pictures, int pic_w, int pic_h, int qp, double lambda, int sao_type, int slice_type,
                                                const int32_t *edge_y, const int32_t *band_y, const int32_t *edge_u, const int32_t *band_u,
                                                const int32_t *edge_v, const int32_t *band_v, void *workspace, int32_t *info_out,
                                                uint16_t *models_out, uvghip_sao_param_t *params_y, uvghip_sao_param_t *params_u,
                                                uvghip_sao_param_t *params_v, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (slice_type < 0 || slice_type > 2) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || qp < 0 || qp > 63 || !(lambda > 0) || sao_type < 1 || sao_type > 3 || !edge_y || !band_y ||
      !edge_u || !band_u || !edge_v || !band_v || !workspace || !info_out || !models_out || !params_y || !params_u || !params_v)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  static_assert(sizeof(sao_info) == 17 * 4, "sao_info_t as 17 ints");
  const int wc = (pic_w + 63) / 64, hc = (pic_h + 63) / 64, n = n_pictures * wc * hc;
  const int omax = (1 << ((bitdepth < 10 ? bitdepth : 10) - 5)) - 1;              // SAO_ABS_OFFSET_MAX (src/global.h:295)
  hipStream_t st = uvghip_stream(stream);
  cand *cw = static_cast<cand *>(workspace);
  sao_candidates_kernel<<<(2 * n + 127) / 128, 128, 0, st>>>(edge_y, band_y, edge_u, band_u, edge_v, band_v, n, omax, cw);
  sao_decide_kernel<<<(n_pictures + 63) / 64, 64, 0, st>>>(edge_y, band_y, edge_u, band_u, edge_v, band_v, cw, n_pictures, wc, hc, qp, lambda,
                                                           sao_type, slice_type, info_out, models_out, params_y, params_u, params_v);
  UVGHIP_CHECK_LAUNCH();
}
