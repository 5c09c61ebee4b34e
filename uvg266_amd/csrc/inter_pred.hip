// uvghip_inter_pred_satd_batch (include/uvg266_hip.h, part 8): luma motion compensation of n candidate motions -- uni- or bi-predicted --
// and the SATD of each prediction against the source block: what the inter search does per merge candidate (uvg_inter_pred_pu with
// predict_luma only + uvg_satd_any_size, src/search_inter.c:1758-1775), for the bi-prediction of the two best uni-predictions
// (:2018-2031) and, with the prediction written out, uvg_inter_recon_cu's luma (src/inter.c:685-748: inter_recon_unipred :400-530,
// uvg_inter_recon_bipred :532-602 with uvg_bipred_average, picture-generic.c:1132-1193).
//
// One wave per candidate.  Per list the (n + 8)^2 window at the vector's integer position is staged in LDS as sample pairs (picture-edge
// clamp = the border replication of inter_cp_with_ext_border / uvg_get_extended_block); a lane owns 8x8 tiles: horizontal and vertical
// 8-tap passes through v_dot2 on pairs give the tile's 14-bit intermediates (an integer vector is phase 0 of the same filter: exactly
// the sample << (14 - depth), so there is one code path), uni-prediction rounds them to samples, bi-prediction adds the two lists'
// intermediates first; the tile's differences go through the Hadamard in the lane's registers.
#include "uvghip_common.h"
#include "satd_tile_dev.h"
#include "vvc_tables.h"

namespace {

struct pred_args {
  const void *cur;
  const void *const *refs;
  int cur_stride, ref_stride, pic_w, pic_h, size, n, pred_stride;
  const uvghip_motion_t *cands;
  uint32_t *satd;
  void *pred;                    // optional: n blocks of size x size samples
};

// the 14-bit intermediates of one 8x8 tile at phase (fx, fy): acc[yy][j] >> 6 (ipol-generic.c:180-211)
__device__ __forceinline__ void tile_intermediates(const uint32_t *wbase, int ww, const uint32_t (&fh)[4], const uint32_t (&fv)[4], int shift1, int (&acc)[8][8])
{
#pragma unroll
  for (int yy = 0; yy < 8; ++yy)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[yy][j] = 0;
  int prev[8];
#pragma unroll
  for (int r = 0; r < 15; ++r) {
    const uint32_t *wr = wbase + r * ww;
    uint32_t P[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) P[k] = wr[k];
    int hcur[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int a = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) a = __builtin_amdgcn_sdot2(__builtin_bit_cast(pk_s16, P[j + 2 * m]), __builtin_bit_cast(pk_s16, fh[m]), a, false);
      hcur[j] = (int)(int16_t)(a >> shift1);
    }
    if (r >= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const pk_s16 pr = __builtin_bit_cast(pk_s16, __builtin_amdgcn_perm((uint32_t)hcur[j], (uint32_t)prev[j], 0x05040100u));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int yy = r - 1 - 2 * m;
          if (yy >= 0 && yy < 8) acc[yy][j] = __builtin_amdgcn_sdot2(pr, __builtin_bit_cast(pk_s16, fv[m]), acc[yy][j], false);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) prev[j] = hcur[j];
  }
#pragma unroll
  for (int yy = 0; yy < 8; ++yy)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[yy][j] = (int)(int16_t)(acc[yy][j] >> 6);
}

template <typename PX>
__global__ void __launch_bounds__(64)
inter_pred_satd_kernel(pred_args A)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t smem32[];
  constexpr int depth = px_traits<PX>::depth;
  const int n = A.size, lane = threadIdx.x, ww = n + 8;
  if ((int)blockIdx.x >= A.n) return;
  const uvghip_motion_t &M = A.cands[blockIdx.x];
  uint32_t *sWin = smem32;                        // (n + 8)^2 pairs
  uint32_t *sCurP = sWin + ww * ww;               // the source block as pairs
  uint32_t *sCoef = sCurP + n * (n >> 1);
  uint32_t *sCost = sCoef + 64;
  int16_t *sHi = (int16_t *)(sCost + 4);          // list 0's intermediates while list 1 is filtered: n * n
  const PX *cur = (const PX *)A.cur;
  for (int i = lane; i < n * (n >> 1); i += 64) {
    const int yy = i / (n >> 1), x2 = i - yy * (n >> 1);
    const PX *p = cur + (size_t)(M.y + yy) * A.cur_stride + M.x + 2 * x2;
    sCurP[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
  }
  {
    const int ph = lane >> 2, m = lane & 3;
    sCoef[lane] = (uint32_t)(uint16_t)(int16_t)VVC_LUMA_FILTER[8 * ph + 2 * m] | ((uint32_t)(uint16_t)(int16_t)VVC_LUMA_FILTER[8 * ph + 2 * m + 1] << 16);
  }
  if (lane == 0) sCost[0] = 0;
  const int tiles_x = n >> 3, tiles = tiles_x * tiles_x;
  const int dir = M.dir;
  const int n_lists = dir == 3 ? 2 : 1;
  for (int pass = 0; pass < n_lists; ++pass) {
    const int l = dir == 3 ? pass : dir - 1;
    const PX *ref = (const PX *)A.refs[M.ref[l]];
    const int mvx = M.mv[l][0], mvy = M.mv[l][1];
    const int ox = M.x + (mvx >> 4) - 4, oy = M.y + (mvy >> 4) - 4;
    __syncthreads();                               // (the previous pass is done with the window)
    for (int i = lane; i < ww * ww; i += 64) {
      const int yy = i / ww, xx = i - yy * ww;
      const PX *row = ref + (size_t)clampi(oy + yy, 0, A.pic_h - 1) * A.ref_stride;
      sWin[i] = (uint32_t)row[clampi(ox + xx, 0, A.pic_w - 1)] | ((uint32_t)row[clampi(ox + xx + 1, 0, A.pic_w - 1)] << 16);
    }
    __syncthreads();
    uint32_t fh[4], fv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { fh[m] = sCoef[(mvx & 15) * 4 + m]; fv[m] = sCoef[(mvy & 15) * 4 + m]; }
    const bool last = pass == n_lists - 1;
    for (int t = lane; t < tiles; t += 64) {
      const int ty = t / tiles_x, tx = t - ty * tiles_x;
      int acc[8][8];
      tile_intermediates(sWin + (ty * 8 + 1) * ww + tx * 8 + 1, ww, fh, fv, depth - 8, acc);
      if (!last) {                                 // bi-prediction, first list: park the intermediates
#pragma unroll
        for (int yy = 0; yy < 8; ++yy)
#pragma unroll
          for (int j = 0; j < 8; ++j) sHi[(ty * 8 + yy) * n + tx * 8 + j] = (int16_t)acc[yy][j];
        continue;
      }
      const int shift = dir == 3 ? 15 - depth : 14 - depth, off = 1 << (shift - 1);
      uint32_t d[8][4];
      const uint32_t *cb = sCurP + (ty * 8) * (n >> 1) + tx * 4;
      const pk_s16 vmax = {(short)px_traits<PX>::maxv, (short)px_traits<PX>::maxv};
#pragma unroll
      for (int yy = 0; yy < 8; ++yy)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int a0 = acc[yy][2 * q], a1 = acc[yy][2 * q + 1];
          if (dir == 3) { a0 += sHi[(ty * 8 + yy) * n + tx * 8 + 2 * q]; a1 += sHi[(ty * 8 + yy) * n + tx * 8 + 2 * q + 1]; }
          const int p0 = (a0 + off) >> shift, p1 = (a1 + off) >> shift;
          pk_s16 v = __builtin_bit_cast(pk_s16, __builtin_amdgcn_perm((uint32_t)p1, (uint32_t)p0, 0x05040100u));
          v = __builtin_elementwise_min(__builtin_elementwise_max(v, (pk_s16){0, 0}), vmax);
          if (A.pred) {
            PX *o = (PX *)A.pred + (size_t)blockIdx.x * A.pred_stride + (ty * 8 + yy) * n + tx * 8 + 2 * q;
            o[0] = (PX)(uint16_t)v.x; o[1] = (PX)(uint16_t)v.y;
          }
          d[yy][q] = pk_sub(cb[yy * (n >> 1) + q], __builtin_bit_cast(uint32_t, v));
        }
      atomicAdd(&sCost[0], satd8_tile_lane(d));
    }
  }
  __syncthreads();
  if (lane == 0) A.satd[blockIdx.x] = sCost[0] >> (depth - 8);
}

}  // namespace

extern "C" int uvghip_inter_pred_satd_batch(int bitdepth, const void *cur, int cur_stride, const void *const *refs_dev, int ref_stride, int pic_w, int pic_h,
                                            int size, const uvghip_motion_t *cands, int n, uint32_t *satd, void *pred, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!cur || !refs_dev || !cands || !satd || n < 0 || pic_w <= 0 || pic_h <= 0 || cur_stride < pic_w || ref_stride < pic_w ||
      (size != 8 && size != 16 && size != 32 && size != 64))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n == 0) return 0;
  const size_t lds = ((size_t)(size + 8) * (size + 8) + (size_t)size * (size / 2) + 64 + 4) * 4 + (size_t)size * size * 2;
  pred_args A{cur, refs_dev, cur_stride, ref_stride, pic_w, pic_h, size, n, size * size, cands, satd, pred};
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) hipLaunchKernelGGL(inter_pred_satd_kernel<uint8_t>, dim3(n), dim3(64), lds, st, A);
  else hipLaunchKernelGGL(inter_pred_satd_kernel<uint16_t>, dim3(n), dim3(64), lds, st, A);
  UVGHIP_CHECK_LAUNCH();
}
