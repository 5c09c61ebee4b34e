// uvghip_me_search_batch (include/uvg266_hip.h, part 8): the motion search of one reference picture for n prediction units, resident
// on the device from the starting points to the quarter-sample vector -- what search_pu_inter_ref + search_frac do per call
// (src/search_inter.c:1300-1500, 1029-1226): select_starting_point (:297-375), early_terminate (:491-539), hexagon_search (:767-847),
// the four fractional steps with SATD (:1133-1216), the motion vector difference cost against both AMVP predictors (calc_mvd_cost
// :449-488, get_mvd_coding_cost :378-393, select_mv_cand :396-446).
//
// One wave per unit.  The search is a chain of data-dependent steps (every step's points depend on the best point so far), so the
// parallelism inside a unit is across the samples of a point and across the points of a step:
//   * integer steps: the unit's source block sits in LDS; a step's points (4 + 4 of the small diamond, 6 of the hexagon, 3 per hexagon
//     move, 8 of the final square) are evaluated together -- lanes are split into groups of n * n / 8 (at most 64), a group sums one
//     point's absolute differences over 8-sample row segments (reference samples straight from L2 / HBM with the picture-edge clamp
//     that uvg_image_calc_sad's border cases amount to), a segmented butterfly adds the group up;
//   * fractional steps: the (n + 8)^2 window around the integer vector is staged in LDS once as sample pairs; a lane owns one 8x8
//     tile of one candidate: 15 x 8 horizontal intermediates and the vertical pass through v_dot2 on pairs, then the Hadamard in its
//     own registers (satd_tile_dev.h) -- the scheme of ipol.hip's frac_satd_tile_kernel;
//   * the step's bookkeeping (cost = SAD + bits * lambda_sqrt in double, strict "<" in the reference's order of points) is wave-uniform:
//     every lane reads the same LDS results and takes the same branch.
#include "uvghip_common.h"
#include "satd_tile_dev.h"
#include "vvc_tables.h"

namespace {

struct me_args {
  const void *cur;
  const void *const *refs;          // device table of reference luma planes
  int cur_stride, ref_stride, pic_w, pic_h;
  double lambda_sqrt;
  int fme_level, size, n;
  const uvghip_me_job_t *jobs;
  uvghip_me_result_t *out;
};

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ unsigned golomb_bits(unsigned s)        // get_ep_ex_golomb_bitcost
{
  unsigned bins = 0;
  if (s >= 1u << 8) { bins += 16; s >>= 8; }
  if (s >= 1u << 4) { bins += 8; s >>= 4; }
  if (s >= 1u << 2) { bins += 4; s >>= 2; }
  if (s >= 1u << 1) bins += 2;
  return bins;
}
__device__ __forceinline__ int to_quarter(int v) { return v >= 0 ? (v + 1) >> 2 : (v + 2) >> 2; }
__device__ __forceinline__ int mvd_bits(int dx, int dy)             // get_mvd_coding_cost: an integer number of bits
{
  const int ax = iabs(dx), ay = iabs(dy);
  return 4 + (ax == 1) + (ay == 1) + (int)golomb_bits((unsigned)ax) + (int)golomb_bits((unsigned)ay);
}
// select_mv_cand's cheaper predictor for a vector in 1/16 units: -> bits, *which = the predictor
__device__ __forceinline__ int best_predictor_bits(const int32_t (&cand)[2][2], int mx, int my, int *which)
{
  const int c1 = mvd_bits(to_quarter(mx - cand[0][0]), to_quarter(my - cand[0][1]));
  const bool same = cand[0][0] == cand[1][0] && cand[0][1] == cand[1][1];
  const int c2 = same ? c1 : mvd_bits(to_quarter(mx - cand[1][0]), to_quarter(my - cand[1][1]));
  if (which) *which = c2 < c1 ? 1 : 0;
  return c1 < c2 ? c1 : c2;
}

// the search state every lane carries identically
struct best_t { double cost, bits; int mx, my; };      // vector in 1/16 units

template <typename PX> struct wave_ctx {
  const PX *ref;
  int stride, W, H, x, y, n;
  const PX *sCur;                  // the unit's source block, n x n, in LDS
  uint32_t *sSad;                  // results of the points of a step
  double lambda_sqrt;
  int32_t cand[2][2];
};

// SADs of K points (integer offsets px[k], py[k]) -> C.sSad[k]
template <typename PX>
__device__ void sad_points(const wave_ctx<PX> &C, int K, const int *px, const int *py)
{
  constexpr int depth = px_traits<PX>::depth;
  const int lane = threadIdx.x, n = C.n, segs_row = n >> 3, segs = n * segs_row;
  const int lpp = segs < 64 ? segs : 64, ppp = 64 / lpp;
  for (int p0 = 0; p0 < K; p0 += ppp) {
    const int p = p0 + lane / lpp, sl = lane % lpp;
    int acc = 0;
    if (p < K) {
      const int rx = C.x + px[p], ry = C.y + py[p];
      for (int s = sl; s < segs; s += lpp) {
        const int row = s / segs_row, c8 = (s - row * segs_row) * 8;
        int r[8];
        load_row_clamped<PX, 8>(C.ref, C.stride, C.W, C.H, rx + c8, ry + row, r);
        const PX *c = C.sCur + row * n + c8;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += iabs((int)c[i] - r[i]);
      }
    }
    for (int off = lpp >> 1; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sl == 0 && p < K) C.sSad[p] = (uint32_t)acc >> (depth - 8);
  }
  __syncthreads();
}

// check_mv_cost over the K points just measured, in order (wave-uniform); -> index of the last point that became the best, or -1
template <typename PX>
__device__ int take_points(const wave_ctx<PX> &C, int K, const int *px, const int *py, best_t &B)
{
  int last = -1;
  for (int k = 0; k < K; ++k) {
    double cost = (double)C.sSad[k];
    if (cost >= B.cost) continue;
    const int bits = best_predictor_bits(C.cand, px[k] * 16, py[k] * 16, nullptr);
    cost += (double)bits * C.lambda_sqrt;
    if (cost >= B.cost) continue;
    B.cost = cost; B.bits = (double)bits; B.mx = px[k] * 16; B.my = py[k] * 16;
    last = k;
  }
  __syncthreads();                 // sSad is free again
  return last;
}

template <typename PX>
__global__ void __launch_bounds__(64)
me_search_kernel(me_args A)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t smem32[];
  constexpr int depth = px_traits<PX>::depth;
  const int n = A.size, lane = threadIdx.x;
  if ((int)blockIdx.x >= A.n) return;
  const uvghip_me_job_t &J = A.jobs[blockIdx.x];        // (read through the cache: an indexed private copy would live in scratch)
  // LDS: window (n + 8)^2 dwords | source pairs n * n / 2 dwords | source samples n * n | coefficients 64 | costs 8 | sads 8 | points 16
  const int ww = n + 8;
  uint32_t *sWin = smem32;
  uint32_t *sCurP = sWin + ww * ww;
  uint32_t *sCoef = sCurP + n * (n >> 1);
  uint32_t *sCost = sCoef + 64;
  uint32_t *sSad = sCost + 8;
  int *sPx = (int *)(sSad + 8), *sPy = sPx + 8;
  PX *sCur = (PX *)(sPy + 8);
  const PX *cur = (const PX *)A.cur;
  for (int i = lane; i < n * n; i += 64) { const int yy = i / n, xx = i - yy * n; sCur[i] = cur[(size_t)(J.y + yy) * A.cur_stride + J.x + xx]; }
  for (int i = lane; i < n * (n >> 1); i += 64) {
    const int yy = i / (n >> 1), x2 = i - yy * (n >> 1);
    const PX *p = cur + (size_t)(J.y + yy) * A.cur_stride + J.x + 2 * x2;
    sCurP[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
  }
  {
    const int ph = lane >> 2, m = lane & 3;
    sCoef[lane] = (uint32_t)(uint16_t)(int16_t)VVC_LUMA_FILTER[8 * ph + 2 * m] | ((uint32_t)(uint16_t)(int16_t)VVC_LUMA_FILTER[8 * ph + 2 * m + 1] << 16);
  }
  __syncthreads();

  wave_ctx<PX> C;
  C.ref = (const PX *)A.refs[J.ref]; C.stride = A.ref_stride; C.W = A.pic_w; C.H = A.pic_h; C.x = J.x; C.y = J.y; C.n = n;
  C.sCur = sCur; C.sSad = sSad; C.lambda_sqrt = A.lambda_sqrt;
  C.cand[0][0] = J.mv_cand[0][0]; C.cand[0][1] = J.mv_cand[0][1]; C.cand[1][0] = J.mv_cand[1][0]; C.cand[1][1] = J.mv_cand[1][1];
  best_t B;
  B.cost = 1.7976931348623157e308; B.bits = 2147483647.0; B.mx = 0; B.my = 0;

  bool skip_me = false;
  const bool frac_only = J.n_start < 0;          // only search_frac, around the integer vector extra_mv (a multiple of 16)
  if (frac_only) { B.mx = J.extra_mv[0]; B.my = J.extra_mv[1]; B.cost = 0; B.bits = 0; }
  // ---- select_starting_point: the zero vector, the reference picture's own vector unless a merge candidate has it, the merge vectors ----
  if (!frac_only) {
    int K = 0;
    const int ex = J.extra_mv[0] >> 4, ey = J.extra_mv[1] >> 4;
    bool in_merge = false;
    for (int i = 0; i < J.n_start; ++i) in_merge |= J.start[i][0] == ex * 16 && J.start[i][1] == ey * 16;
    if (lane == 0) {
      sPx[K] = 0; sPy[K] = 0;
    }
    ++K;
    if ((ex != 0 || ey != 0) && !in_merge) { if (lane == 0) { sPx[K] = ex; sPy[K] = ey; } ++K; }
    for (int i = 0; i < J.n_start; ++i) {
      const int sx = (J.start[i][0] + 8) >> 4, sy = (J.start[i][1] + 8) >> 4;
      if (sx == 0 && sy == 0) continue;
      if (lane == 0) { sPx[K] = sx; sPy[K] = sy; }
      ++K;
    }
    __syncthreads();
    sad_points(C, K, sPx, sPy);
    take_points(C, K, sPx, sPy, B);
  }
  // ---- early_terminate: two rounds of the small diamond ----
  if (!frac_only) {
    const int dxs[7] = {0, -1, 0, 1, 0, -1, 0}, dys[7] = {-1, 0, 1, 0, -1, 0, 0};
    int mx = B.mx >> 4, my = B.my >> 4, first = 0, lastp = 3;
    for (int k = 0; k < 2; ++k) {
      const double threshold = B.cost;
      const int K = lastp - first + 1;
      if (lane < K) {
        const int d = first + lane;
        sPx[lane] = mx + (d == 1 || d == 5 ? -1 : (d == 3 ? 1 : 0));
        sPy[lane] = my + (d == 0 || d == 4 ? -1 : (d == 2 ? 1 : 0));
      }
      __syncthreads();
      sad_points(C, K, sPx, sPy);
      const int hit = take_points(C, K, sPx, sPy, B);
      const int best_index = hit < 0 ? 6 : first + hit;
      mx += dxs[best_index]; my += dys[best_index];
      if (B.cost >= threshold) { skip_me = true; break; }
      first = (best_index + 3) % 4;
      lastp = first + 2;
    }
  }
  // ---- hexagon_search ----
  if (!skip_me && !frac_only) {
    const int hx[9] = {0, 1, 2, 1, -1, -2, -1, 1, 2}, hy[9] = {0, -2, 0, 2, 2, 0, -2, -2, 0};
    int mx = B.mx >> 4, my = B.my >> 4, best_index = 0;
    if (lane < 6) { sPx[lane] = mx + hx[1 + lane]; sPy[lane] = my + hy[1 + lane]; }
    __syncthreads();
    sad_points(C, 6, sPx, sPy);
    {
      const int hit = take_points(C, 6, sPx, sPy, B);
      if (hit >= 0) best_index = 1 + hit;
    }
    while (best_index != 0) {
      const int start = best_index == 1 ? 6 : (best_index == 8 ? 1 : best_index - 1);
      mx += hx[best_index]; my += hy[best_index];
      best_index = 0;
      if (lane < 3) { sPx[lane] = mx + hx[start + lane]; sPy[lane] = my + hy[start + lane]; }
      __syncthreads();
      sad_points(C, 3, sPx, sPy);
      const int hit = take_points(C, 3, sPx, sPy, B);
      if (hit >= 0) best_index = start + hit;
    }
    if (lane < 8) {
      const int sx[8] = {0, -1, 1, 0, -1, 1, -1, 1}, sy[8] = {-1, 0, 0, 1, -1, -1, 1, 1};
      sPx[lane] = mx + sx[lane]; sPy[lane] = my + sy[lane];
    }
    __syncthreads();
    sad_points(C, 8, sPx, sPy);
    take_points(C, 8, sPx, sPy, B);
  }
  uvghip_me_result_t R;
  R.int_mv[0] = B.mx; R.int_mv[1] = B.my; R.int_cost = B.cost; R.int_bits = B.bits;

  // ---- search_frac: the window around the integer vector, then four steps of four candidates (the first also prices the centre) ----
  if (A.fme_level > 0) {
    const int bx = C.x + (B.mx >> 4), by = C.y + (B.my >> 4);
    for (int i = lane; i < ww * ww; i += 64) {
      const int yy = i / ww, xx = i - yy * ww;
      const PX *row = C.ref + (size_t)clampi(by - 4 + yy, 0, C.H - 1) * C.stride;
      sWin[i] = (uint32_t)row[clampi(bx - 4 + xx, 0, C.W - 1)] | ((uint32_t)row[clampi(bx - 4 + xx + 1, 0, C.W - 1)] << 16);
    }
    __syncthreads();
    const int tiles_x = n >> 3, tiles = tiles_x * tiles_x;
    const int wp_shift = 14 - depth, wp_off = 1 << (wp_shift - 1);
    const pk_s16 vmax = {(short)px_traits<PX>::maxv, (short)px_traits<PX>::maxv};
    const int sqx[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1}, sqy[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
    int mvx = (B.mx >> 4) * 2, mvy = (B.my >> 4) * 2;      // half-sample units, then quarter
    double cost = 0, bitcost = 0;
    int best_index = 0, i0 = 1;
    for (int step = 0; step < A.fme_level; ++step) {
      const int mv_shift = step < 2 ? 3 : 2;
      const int K = step == 0 ? 5 : 4;                       // step 0 also measures the integer position (index 4 of the pass)
      if (lane < 8) sCost[lane] = 0;
      __syncthreads();
      // candidate c of this pass: vector relative to the window's integer position, 1/16 units
      const int cx0 = mvx * (1 << mv_shift) - (B.mx >> 4) * 16, cy0 = mvy * (1 << mv_shift) - (B.my >> 4) * 16;
      for (int task = lane; task < tiles * K; task += 64) {
        const int c = task / tiles, t = task - c * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
        const int rel_x = c == 4 ? 0 : cx0 + sqx[i0 + c] * (1 << mv_shift), rel_y = c == 4 ? 0 : cy0 + sqy[i0 + c] * (1 << mv_shift);
        const int ix = rel_x >> 4, iy = rel_y >> 4;
        uint32_t fh[4], fv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { fh[m] = sCoef[(rel_x & 15) * 4 + m]; fv[m] = sCoef[(rel_y & 15) * 4 + m]; }
        const uint32_t *wbase = sWin + (ty * 8 + 1 + iy) * ww + tx * 8 + 1 + ix;
        int acc[8][8];
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
            hcur[j] = (int)(int16_t)(a >> (depth - 8));
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
        uint32_t d[8][4];
        const uint32_t *cb = sCurP + (ty * 8) * (n >> 1) + tx * 4;
#pragma unroll
        for (int yy = 0; yy < 8; ++yy)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int p0 = ((acc[yy][2 * q] >> 6) + wp_off) >> wp_shift, p1 = ((acc[yy][2 * q + 1] >> 6) + wp_off) >> wp_shift;
            pk_s16 v = __builtin_bit_cast(pk_s16, __builtin_amdgcn_perm((uint32_t)p1, (uint32_t)p0, 0x05040100u));
            v = __builtin_elementwise_min(__builtin_elementwise_max(v, (pk_s16){0, 0}), vmax);
            d[yy][q] = pk_sub(cb[yy * (n >> 1) + q], __builtin_bit_cast(uint32_t, v));
          }
        atomicAdd(&sCost[c], satd8_tile_lane(d));
      }
      __syncthreads();
      if (step == 0) {
        unsigned c0 = sCost[4] >> (depth - 8);
        const int bits = best_predictor_bits(C.cand, B.mx & ~15, B.my & ~15, nullptr);       // (the integer vector: a multiple of 16 already)
        c0 += (unsigned)((double)bits * C.lambda_sqrt);
        cost = (double)c0; bitcost = (double)bits;
      }
      for (int j = 0; j < 4; ++j) {
        unsigned cj = sCost[j] >> (depth - 8);
        const int bits = best_predictor_bits(C.cand, (mvx + sqx[i0 + j]) * (1 << mv_shift), (mvy + sqy[i0 + j]) * (1 << mv_shift), nullptr);
        cj += (unsigned)((double)bits * C.lambda_sqrt);
        if ((double)cj < cost) { cost = (double)cj; bitcost = (double)bits; best_index = i0 + j; }
      }
      __syncthreads();
      i0 += 4;
      if (step == 1 || step == A.fme_level - 1) {
        mvx += sqx[best_index]; mvy += sqy[best_index];
        if (step == (A.fme_level - 1 < 1 ? A.fme_level - 1 : 1)) { mvx *= 2; mvy *= 2; best_index = 0; i0 = 1; }
      }
    }
    B.mx = mvx * 4; B.my = mvy * 4; B.cost = cost; B.bits = bitcost;
  }
  int which = 0;
  best_predictor_bits(C.cand, B.mx, B.my, &which);
  {
    const bool same = C.cand[0][0] == C.cand[1][0] && C.cand[0][1] == C.cand[1][1];
    if (same) which = 0;                                   // select_mv_cand(..., NULL): the first of two equal predictors
  }
  R.mv[0] = B.mx; R.mv[1] = B.my; R.cost = B.cost; R.bits = B.bits; R.mv_cand = which; R.skipped_hexagon = skip_me;
  if (lane == 0) A.out[blockIdx.x] = R;
}

}  // namespace

extern "C" int uvghip_me_search_batch(int bitdepth, const void *cur, int cur_stride, const void *const *refs_dev, int ref_stride, int pic_w, int pic_h,
                                      double lambda_sqrt, int fme_level, int size, const uvghip_me_job_t *jobs, int n, uvghip_me_result_t *results,
                                      void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!cur || !refs_dev || !jobs || !results || n < 0 || pic_w <= 0 || pic_h <= 0 || cur_stride < pic_w || ref_stride < pic_w ||
      (size != 8 && size != 16 && size != 32 && size != 64) || (fme_level != 0 && fme_level != 4) || !(lambda_sqrt > 0))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n == 0) return 0;
  const size_t b = bitdepth == 8 ? 1 : 2;
  const size_t lds = ((size_t)(size + 8) * (size + 8) + (size_t)size * (size / 2) + 64 + 8 + 8 + 16) * 4 + (size_t)size * size * b;
  me_args A{cur, refs_dev, cur_stride, ref_stride, pic_w, pic_h, lambda_sqrt, fme_level, size, n, jobs, results};
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) hipLaunchKernelGGL(me_search_kernel<uint8_t>, dim3(n), dim3(64), lds, st, A);
  else hipLaunchKernelGGL(me_search_kernel<uint16_t>, dim3(n), dim3(64), lds, st, A);
  UVGHIP_CHECK_LAUNCH();
}
