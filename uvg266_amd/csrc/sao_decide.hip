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
#include "sao_decide_dev.h"

namespace {
using namespace saod;

__global__ void sao_candidates_kernel(const int32_t *__restrict__ edge_y, const int32_t *__restrict__ band_y, const int32_t *__restrict__ edge_u,
                                      const int32_t *__restrict__ band_u, const int32_t *__restrict__ edge_v, const int32_t *__restrict__ band_v,
                                      int n, int omax, cand *__restrict__ out)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n) return;
  const int k = t >> 1, grp = t & 1, bufs = grp ? 2 : 1;
  const int32_t *E[2] = {grp ? edge_u + (size_t)k * 40 : edge_y + (size_t)k * 40, edge_v + (size_t)k * 40};
  const int32_t *B[2] = {grp ? band_u + (size_t)k * 64 : band_y + (size_t)k * 64, band_v + (size_t)k * 64};
  sao_candidates_one(E, B, bufs, omax, out[t]);                    // written in place (a local copy indexed by class would live in scratch memory)
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
        if (cy == 0) sao_models_init(m, slice_type, qp);
        else m = row_start;
      }
      sao_info &L = luma[2 * k], &C = luma[2 * k + 1];
      const sao_info *top_l = cy ? &luma[2 * (k - wc)] : nullptr, *left_l = cx ? &luma[2 * (k - 1)] : nullptr;
      const sao_info *top_c = cy ? &luma[2 * (k - wc) + 1] : nullptr, *left_c = cx ? &luma[2 * (k - 1) + 1] : nullptr;
      sao_decide_one(m, lambda, sao_type, cands[2 * g], cands[2 * g + 1], edge_y + g * 40, band_y + g * 64, edge_u + g * 40, band_u + g * 64, edge_v + g * 40,
                     band_v + g * 64, L, C, top_l, top_c, left_l, left_c);
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
  return (size_t)n_pictures * ((pic_w + 63) / 64) * ((pic_h + 63) / 64) * 2 * sizeof(saod::cand);
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
  static_assert(sizeof(saod::sao_info) == 17 * 4, "sao_info_t as 17 ints");
  const int wc = (pic_w + 63) / 64, hc = (pic_h + 63) / 64, n = n_pictures * wc * hc;
  const int omax = (1 << ((bitdepth < 10 ? bitdepth : 10) - 5)) - 1;              // SAO_ABS_OFFSET_MAX (src/global.h:295)
  hipStream_t st = uvghip_stream(stream);
  saod::cand *cw = static_cast<saod::cand *>(workspace);
  sao_candidates_kernel<<<(2 * n + 127) / 128, 128, 0, st>>>(edge_y, band_y, edge_u, band_u, edge_v, band_v, n, omax, cw);
  sao_decide_kernel<<<(n_pictures + 63) / 64, 64, 0, st>>>(edge_y, band_y, edge_u, band_u, edge_v, band_v, cw, n_pictures, wc, hc, qp, lambda,
                                                           sao_type, slice_type, info_out, models_out, params_y, params_u, params_v);
  UVGHIP_CHECK_LAUNCH();
}
