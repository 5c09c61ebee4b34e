// "ipol" strategy group on gfx950: sub-sample interpolation for motion
// compensation and fractional motion estimation.
// Bit-exact with src/strategies/generic/ipol-generic.c:
//   sample_quarterpel_luma(_hi) :134-211   8-tap, 1/16 phases (filter.c:62-80)
//   sample_octpel_chroma(_hi)   :681-758   4-tap, 1/32 phases (filter.c:82-116)
//   get_extended_block          :761-810   = edge-replicated reads, done here by coordinate clamping
//   filter_hpel/qpel_blocks_*   :213-679   the four candidate blocks of each fractional-ME step are the
//        predictions at +-1/2 and +-1/4 sample offsets (search_inter.c:1133-1216); uvghip_frac_satd_batch
//        evaluates any list of such offsets and returns the SATD the search compares (:1172).
//   bipred_average              picture-generic.c:1132-1247
//
// Data flow: per block the (w+7) x (h+7) reference window (+1 for fractional
// ME) is read from HBM once into LDS with edge replication; the horizontal
// pass writes int16 intermediates to LDS, the vertical pass reads them back
// and either stores the block (MC) or feeds the row-per-lane Hadamard (ME).
#include <type_traits>
#include "uvghip_common.h"
#include "percall.h"
#include "ref_abi.h"
#include "satd_dev.h"
#include "satd_tile_dev.h"
#include "vvc_tables.h"

// Stage ref[(y0+yy), (x0+xx)] for yy < wh, xx < ww into LDS (pitch wp), clamped to the picture.
template <typename PX>
__device__ __forceinline__ void stage_window(const PX *__restrict__ ref, int stride, int pic_w, int pic_h, int x0, int y0,
                                             int ww, int wh, uint16_t *win, int wp)
{
  for (int i = threadIdx.x; i < ww * wh; i += blockDim.x) {
    const int yy = i / ww, xx = i - yy * ww;
    win[yy * wp + xx] = ref[(size_t)clampi(y0 + yy, 0, pic_h - 1) * stride + clampi(x0 + xx, 0, pic_w - 1)];
  }
}

// Horizontal pass: tmp[y][x] = (sum_k f[k] * win[y][x + xo + k]) >> shift1 for y < rows, x < w.
template <int TAPS>
__device__ __forceinline__ void hor_pass(const uint16_t *win, int wp, int xo, int yo, int w, int rows, const int8_t *f,
                                         int shift1, int16_t *tmp, int tp)
{
  for (int i = threadIdx.x; i < w * rows; i += blockDim.x) {
    const int y = i / w, x = i - y * w;
    const uint16_t *p = win + (y + yo) * wp + x + xo;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc += f[k] * (int)p[k];
    tmp[y * tp + x] = (int16_t)(acc >> shift1);
  }
}

// ---------------------------------------------------------------- motion compensation ----
template <typename PX, int TAPS>
__global__ void __launch_bounds__(256)
mc_kernel(const PX *__restrict__ ref, int stride, int pic_w, int pic_h, int w, int h,
          const uvghip_mc_blk_t *__restrict__ blks, int hi, void *__restrict__ dst)
{
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  constexpr int OFF = TAPS == 8 ? 3 : 1;
  const int ww = w + TAPS - 1, wh = h + TAPS - 1, wp = ww + 1;
  uint16_t *win = smem;
  int16_t *tmp = reinterpret_cast<int16_t *>(smem + wp * wh);
  const uvghip_mc_blk_t b = blks[blockIdx.x];
  const int8_t *tab = TAPS == 8 ? VVC_LUMA_FILTER : VVC_CHROMA_FILTER;
  const int8_t *fh = tab + TAPS * b.fx, *fv = tab + TAPS * b.fy;
  stage_window<PX>(ref, stride, pic_w, pic_h, b.x - OFF, b.y - OFF, ww, wh, win, wp);
  __syncthreads();
  constexpr int depth = px_traits<PX>::depth;
  hor_pass<TAPS>(win, wp, 0, 0, w, wh, fh, depth - 8, tmp, w);
  __syncthreads();
  const int wp_shift = 14 - depth, wp_off = 1 << (wp_shift - 1);
  for (int i = threadIdx.x; i < w * h; i += blockDim.x) {
    const int y = i / w, x = i - y * w;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc += fv[k] * (int)tmp[(y + k) * w + x];
    acc >>= 6;
    if (hi) reinterpret_cast<int16_t *>(dst)[(size_t)blockIdx.x * w * h + i] = (int16_t)acc;
    else reinterpret_cast<PX *>(dst)[(size_t)blockIdx.x * w * h + i] = (PX)clampi((acc + wp_off) >> wp_shift, 0, px_traits<PX>::maxv);
  }
}

extern "C" int uvghip_mc_batch(int bitdepth, const void *ref, int ref_stride, int pic_w, int pic_h, int is_chroma,
                               int width, int height, const uvghip_mc_blk_t *blks, int n, int hi, void *dst, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (width < 1 || height < 1 || width > 64 || height > 64) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  const int taps = is_chroma ? 4 : 8;
  const size_t lds = ((size_t)(width + taps) * (height + taps - 1) + (size_t)width * (height + taps - 1)) * 2;
  hipStream_t st = uvghip_stream(stream);
#define MC(PX, T) mc_kernel<PX, T><<<n, 256, lds, st>>>((const PX *)ref, ref_stride, pic_w, pic_h, width, height, blks, hi, dst)
  if (bitdepth == 8) { if (is_chroma) MC(uint8_t, 4); else MC(uint8_t, 8); }
  else { if (is_chroma) MC(uint16_t, 4); else MC(uint16_t, 8); }
#undef MC
  UVGHIP_CHECK_LAUNCH();
}

// --------------------------------------------------------- fractional ME: sample + SATD ----
template <typename PX>
__global__ void __launch_bounds__(256)
frac_satd_kernel(const PX *__restrict__ cur, int cur_stride, const PX *__restrict__ ref, int ref_stride, int pic_w, int pic_h,
                 int w, int h, const uvghip_blk_t *__restrict__ blks, const int16_t *__restrict__ cands, int n_cand,
                 uint32_t *__restrict__ costs)
{
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  // window covers integer offsets -1..0 around the block plus the 8-tap support
  const int ww = w + 8, wh = h + 8, wp = ww + 1;
  uint16_t *win = smem;
  int16_t *tmp = reinterpret_cast<int16_t *>(smem + wp * wh);          // (h+7) x w
  uint16_t *sCur = reinterpret_cast<uint16_t *>(tmp + (h + 7) * w);    // h x w
  const uvghip_blk_t b = blks[blockIdx.x];
  stage_window<PX>(ref, ref_stride, pic_w, pic_h, b.ref_x - 4, b.ref_y - 4, ww, wh, win, wp);
  for (int i = threadIdx.x; i < w * h; i += blockDim.x) {
    const int y = i / w, x = i - y * w;
    sCur[i] = cur[(size_t)(b.cur_y + y) * cur_stride + b.cur_x + x];
  }
  constexpr int depth = px_traits<PX>::depth;
  const int wp_shift = 14 - depth, wp_off = 1 << (wp_shift - 1);
  const satd_tiling T = make_tiling(w, h);
  for (int c = 0; c < n_cand; ++c) {
    const int mvx = cands[2 * c], mvy = cands[2 * c + 1];
    const int ix = mvx >> 4, iy = mvy >> 4;                              // -1 or 0 (|mv| < 16)
    const int8_t *fh = VVC_LUMA_FILTER + 8 * (mvx & 15), *fv = VVC_LUMA_FILTER + 8 * (mvy & 15);
    __syncthreads();                                                     // window/cur staged; previous tmp consumed
    hor_pass<8>(win, wp, 1 + ix, 1 + iy, w, h + 7, fh, depth - 8, tmp, w);
    __syncthreads();
    // vertical pass fused with the Hadamard: every lane produces one row segment of the prediction
    auto diff = [&](int x, int y, auto &d) {
      constexpr int N = sizeof(d) / sizeof(d[0]);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += fv[k] * (int)tmp[(y + k) * w + x + i];
        const int p = clampi(((acc >> 6) + wp_off) >> wp_shift, 0, px_traits<PX>::maxv);
        d[i] = (int)sCur[y * w + x + i] - p;
      }
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave takes every 4th group-task; satd_block distributes over the 64 lanes of a wave,
    // so give each wave a quarter of the rows by offsetting tiles: simplest is one wave per candidate pass
    int total = 0;
    if (wave == 0) total = satd_block<PX>(T, lane, 64, true, diff);
    if (threadIdx.x == 0) costs[(size_t)blockIdx.x * n_cand + c] = (uint32_t)total >> (depth - 8);
  }
}

// ---- fractional ME, tile per lane (blocks whose sides are multiples of 8) ----------------------------------------
// A lane owns one (block, 8x8 tile, candidate): it filters its tile's 15 x 8 horizontal intermediates row by row
// (window rows in LDS as dword pair rows, so the eight taps of an output are four aligned dwords feeding v_dot2),
// scatters each finished pair of intermediate rows into the 64 vertical accumulators (again v_dot2 on row pairs),
// rounds/clips the 64 predicted samples, subtracts the current tile and runs the Hadamard in its own registers
// (satd_tile_dev.h).  Lanes of a wave work on different candidates and tiles: the filter phases are per-lane data
// (coefficient pairs from an LDS table), not control flow, so there is no divergence and no barrier after staging.
template <typename PX>
__global__ void __launch_bounds__(256)
frac_satd_tile_kernel(const PX *__restrict__ cur, int cur_stride, const PX *__restrict__ ref, int ref_stride, int pic_w, int pic_h,
                      int w, int h, const uvghip_blk_t *__restrict__ blks, int n, int bpw, const int16_t *__restrict__ cands,
                      int n_cand, uint32_t *__restrict__ costs)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t smem32[];
  constexpr int depth = px_traits<PX>::depth;
  const int ww = w + 8, wh = h + 8;                     // window samples per row / rows; dword i of a row = (s[i], s[i+1])
  const int tiles_x = w >> 3, tiles = tiles_x * (h >> 3);
  uint32_t *sWin = smem32;                               // [bpw][wh][ww]
  uint32_t *sCur = sWin + (size_t)bpw * wh * ww;         // [bpw][h][w/2] packed pairs
  uint32_t *sCoef = sCur + (size_t)bpw * h * (w >> 1);   // [16 phases][4 pairs]
  uint32_t *sCost = sCoef + 64;                          // [bpw][n_cand]
  const int blk0 = blockIdx.x * bpw;
  const int here = min(bpw, n - blk0);
  for (int i = threadIdx.x; i < here * wh * ww; i += 256) {
    const int b = i / (wh * ww), r = i - b * (wh * ww), y = r / ww, x = r - y * ww;
    const uvghip_blk_t B = blks[blk0 + b];
    const PX *row = ref + (size_t)clampi(B.ref_y - 4 + y, 0, pic_h - 1) * ref_stride;
    sWin[i] = (uint32_t)row[clampi(B.ref_x - 4 + x, 0, pic_w - 1)] | ((uint32_t)row[clampi(B.ref_x - 4 + x + 1, 0, pic_w - 1)] << 16);
  }
  for (int i = threadIdx.x; i < here * h * (w >> 1); i += 256) {
    const int b = i / (h * (w >> 1)), r = i - b * (h * (w >> 1)), y = r / (w >> 1), x2 = r - y * (w >> 1);
    const uvghip_blk_t B = blks[blk0 + b];
    const PX *p = cur + (size_t)(B.cur_y + y) * cur_stride + B.cur_x + 2 * x2;
    sCur[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
  }
  if (threadIdx.x < 64) {
    const int ph = threadIdx.x >> 2, m = threadIdx.x & 3;
    sCoef[threadIdx.x] = (uint32_t)(uint16_t)(int16_t)VVC_LUMA_FILTER[8 * ph + 2 * m] | ((uint32_t)(uint16_t)(int16_t)VVC_LUMA_FILTER[8 * ph + 2 * m + 1] << 16);
  }
  for (int i = threadIdx.x; i < bpw * n_cand; i += 256) sCost[i] = 0;
  __syncthreads();

  const int wp_shift = 14 - depth, wp_off = 1 << (wp_shift - 1);
  const pk_s16 vmax = {(short)px_traits<PX>::maxv, (short)px_traits<PX>::maxv};
  const int per_blk = tiles * n_cand, ntasks = here * per_blk;
  for (int task = threadIdx.x; task < ntasks; task += 256) {
    const int b = task / per_blk, r0 = task - b * per_blk, c = r0 / tiles, t = r0 - c * tiles;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int mvx = cands[2 * c], mvy = cands[2 * c + 1];
    const int ix = mvx >> 4, iy = mvy >> 4;                                       // -1 or 0 (|mv| < 16)
    uint32_t fh[4], fv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { fh[m] = sCoef[(mvx & 15) * 4 + m]; fv[m] = sCoef[(mvy & 15) * 4 + m]; }
    const uint32_t *wbase = sWin + ((size_t)b * wh + (ty * 8 + 1 + iy)) * ww + tx * 8 + 1 + ix;
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
        hcur[j] = (int)(int16_t)(a >> (depth - 8));                               // ipol-generic.c:170 (int16 intermediates)
      }
      if (r >= 1) {
        // row pair (r-1, r) carries taps (2m, 2m+1) of output row yy = r - 1 - 2m
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
    const uint32_t *cb = sCur + ((size_t)b * h + ty * 8) * (w >> 1) + tx * 4;
#pragma unroll
    for (int yy = 0; yy < 8; ++yy)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p0 = ((acc[yy][2 * q] >> 6) + wp_off) >> wp_shift, p1 = ((acc[yy][2 * q + 1] >> 6) + wp_off) >> wp_shift;
        pk_s16 v = __builtin_bit_cast(pk_s16, __builtin_amdgcn_perm((uint32_t)p1, (uint32_t)p0, 0x05040100u));
        v = __builtin_elementwise_min(__builtin_elementwise_max(v, (pk_s16){0, 0}), vmax);
        d[yy][q] = pk_sub(cb[yy * (w >> 1) + q], __builtin_bit_cast(uint32_t, v));
      }
    atomicAdd(&sCost[b * n_cand + c], satd8_tile_lane(d));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < here * n_cand; i += 256) costs[(size_t)blk0 * n_cand + i] = sCost[i] >> (depth - 8);
}

extern "C" int uvghip_frac_satd_batch(int bitdepth, const void *cur, int cur_stride, const void *ref, int ref_stride,
                                      int pic_w, int pic_h, int width, int height, const uvghip_blk_t *blks, int n,
                                      const int16_t *cand_mv, int n_cand, uint32_t *costs, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (width < 4 || height < 4 || (width & 3) || (height & 3) || width > 64 || height > 64 || n_cand < 1)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
  if ((width & 7) == 0 && (height & 7) == 0 && n_cand <= 64) {
    const int tiles = (width / 8) * (height / 8);
    int bpw = 256 / (tiles * n_cand); if (bpw < 1) bpw = 1; if (bpw > 16) bpw = 16;
    const size_t per_blk = (size_t)(width + 8) * (height + 8) * 4 + (size_t)height * (width / 2) * 4 + (size_t)n_cand * 4;
    while (bpw > 1 && bpw * per_blk + 256 > 60 * 1024) --bpw;
    const size_t tl = bpw * per_blk + 256;
    if (tl <= 64 * 1024) {
      const int grid = (n + bpw - 1) / bpw;
      if (bitdepth == 8)
        frac_satd_tile_kernel<uint8_t><<<grid, 256, tl, st>>>((const uint8_t *)cur, cur_stride, (const uint8_t *)ref, ref_stride, pic_w, pic_h, width, height, blks, n, bpw, cand_mv, n_cand, costs);
      else
        frac_satd_tile_kernel<uint16_t><<<grid, 256, tl, st>>>((const uint16_t *)cur, cur_stride, (const uint16_t *)ref, ref_stride, pic_w, pic_h, width, height, blks, n, bpw, cand_mv, n_cand, costs);
      UVGHIP_CHECK_LAUNCH();
    }
  }
  const size_t lds = ((size_t)(width + 9) * (height + 8) + (size_t)(height + 7) * width + (size_t)width * height) * 2;
  if (bitdepth == 8)
    frac_satd_kernel<uint8_t><<<n, 256, lds, st>>>((const uint8_t *)cur, cur_stride, (const uint8_t *)ref, ref_stride, pic_w, pic_h, width, height, blks, cand_mv, n_cand, costs);
  else
    frac_satd_kernel<uint16_t><<<n, 256, lds, st>>>((const uint16_t *)cur, cur_stride, (const uint16_t *)ref, ref_stride, pic_w, pic_h, width, height, blks, cand_mv, n_cand, costs);
  UVGHIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------- bi-prediction average ----
// mode bit0/bit1: operand L0/L1 is a 14-bit int16 intermediate instead of pixels (picture-generic.c:1132-1193)
template <typename PX>
__global__ void __launch_bounds__(256)
bipred_kernel(const void *__restrict__ l0, const void *__restrict__ l1, int mode, size_t total, PX *__restrict__ dst)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  constexpr int depth = px_traits<PX>::depth;
  const int shift = 15 - depth, off = 1 << (shift - 1);
  const int s0 = (mode & 1) ? (int)reinterpret_cast<const int16_t *>(l0)[i] : (int)(int16_t)(reinterpret_cast<const PX *>(l0)[i] << (14 - depth));
  const int s1 = (mode & 2) ? (int)reinterpret_cast<const int16_t *>(l1)[i] : (int)(int16_t)(reinterpret_cast<const PX *>(l1)[i] << (14 - depth));
  dst[i] = (PX)clampi((s0 + s1 + off) >> shift, 0, px_traits<PX>::maxv);
}

extern "C" int uvghip_bipred_average_batch(int bitdepth, const void *l0, const void *l1, int mode, size_t total,
                                           void *dst, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (total == 0) return 0;
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) bipred_kernel<uint8_t><<<grid, 256, 0, st>>>(l0, l1, mode, total, (uint8_t *)dst);
  else bipred_kernel<uint16_t><<<grid, 256, 0, st>>>(l0, l1, mode, total, (uint16_t *)dst);
  UVGHIP_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- extended blocks ----
// uvg_get_extended_block(_wraparound) (ipol-generic.c:761-883) for n blocks at once: always the copy (a device batch
// has no use for "pointer into the frame"), rows clamped to the picture, columns edge-replicated or -- wrap-around
// variant, :836-855 -- taken modulo the picture width; the pad_b_simd rows are zeroed like the reference's.
template <typename PX>
__global__ void __launch_bounds__(256)
ext_block_kernel(const PX *__restrict__ src, int stride, int src_w, int src_h, int wrap, int blk_h, int pad_l, int pad_t,
                 int pad_b, int es, int rows_total, const uvghip_tu_t *__restrict__ pos, PX *__restrict__ dst)
{
  const uvghip_tu_t b = pos[blockIdx.x];
  PX *out = dst + (size_t)blockIdx.x * rows_total * es;
  const int live_rows = pad_t + blk_h + pad_b;
  for (int i = threadIdx.x; i < rows_total * es; i += blockDim.x) {
    const int yy = i / es, xx = i - yy * es;
    PX v = 0;
    if (yy < live_rows) {
      const int y = clampi(b.y - pad_t + yy, 0, src_h - 1);
      int x = b.x - pad_l + xx;
      if (wrap) { if (x < 0) x += src_w; else if (x >= src_w) x -= src_w; }
      else x = clampi(x, 0, src_w - 1);
      v = src[(size_t)y * stride + x];
    }
    out[i] = v;
  }
}

extern "C" int uvghip_extended_block_batch(int bitdepth, const void *src, int src_stride, int src_w, int src_h, int wraparound,
                                           int blk_w, int blk_h, int pad_l, int pad_r, int pad_t, int pad_b, int pad_b_simd,
                                           const uvghip_tu_t *pos, int n, void *dst, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if ((bitdepth != 8 && bitdepth != 10) || blk_w <= 0 || blk_h <= 0 || pad_l < 0 || pad_r < 0 || pad_t < 0 || pad_b < 0 ||
      pad_b_simd < 0 || src_w <= 0 || src_h <= 0)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  const int es = pad_l + blk_w + pad_r, rows = pad_t + blk_h + pad_b + pad_b_simd;
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) ext_block_kernel<uint8_t><<<n, 256, 0, st>>>((const uint8_t *)src, src_stride, src_w, src_h, wraparound, blk_h, pad_l, pad_t, pad_b, es, rows, pos, (uint8_t *)dst);
  else ext_block_kernel<uint16_t><<<n, 256, 0, st>>>((const uint16_t *)src, src_stride, src_w, src_h, wraparound, blk_h, pad_l, pad_t, pad_b, es, rows, pos, (uint16_t *)dst);
  UVGHIP_CHECK_LAUNCH();
}

// ------------------------------------------------ drop-in "ipol" strategy functions (host buffers) ----
// The ipol typedefs (strategies-ipol.h:62-114) take `const encoder_control_t *` first, but the generic
// implementations never read it (ipol-generic.c:134-758: bit depth is the compile-time UVG_BIT_DEPTH), so
// the pointer is carried as an opaque `const void *` and the depth comes from the registrar's argument.
// One call = stage the extended source block -> upload -> mc kernel -> download (percall.h).
namespace {

// sample_quarterpel_luma / sample_octpel_chroma and their 14-bit `_hi` forms.  `src` points at the block's
// top-left inside a caller-padded extended block (TAPS/2-1 samples before, TAPS/2 after, ipol-generic.c:161-165).
template <typename PX, int TAPS, bool HI>
void sample_hip(const void *, PX *src, int16_t src_stride, int width, int height,
                typename std::conditional<HI, int16_t, PX>::type *dst, int16_t dst_stride, int8_t, int8_t,
                const int32_t mv[2])
{
  using OUT = typename std::conditional<HI, int16_t, PX>::type;
  constexpr int OFF = TAPS / 2 - 1;
  const int rw = width + TAPS - 1, rh = height + TAPS - 1;
  percall_ctx *c = percall_get((size_t)rw * rh * sizeof(PX) + (size_t)width * height * sizeof(OUT) + 2048);
  const size_t os = c->stage_block(src - (ptrdiff_t)OFF * src_stride - OFF, (size_t)src_stride, rw, rh, sizeof(PX));
  const size_t ob = c->take(sizeof(uvghip_mc_blk_t));
  const int mask = TAPS == 8 ? 15 : 31;
  *c->hp<uvghip_mc_blk_t>(ob) = uvghip_mc_blk_t{OFF, OFF, mv[0] & mask, mv[1] & mask};
  c->upload(0, c->used);
  const size_t oo = c->take((size_t)width * height * sizeof(OUT));
  c->must(uvghip_mc_batch(px_traits<PX>::depth, c->dp<PX>(os), rw, rw, rh, TAPS == 4, width, height,
                          c->dp<uvghip_mc_blk_t>(ob), 1, HI, c->dp<OUT>(oo), c->stream), "mc launch");
  c->download(oo, (size_t)width * height * sizeof(OUT));
  c->sync();
  for (int y = 0; y < height; ++y)
    memcpy(dst + (ptrdiff_t)y * dst_stride, c->hp<OUT>(oo) + (size_t)y * width, (size_t)width * sizeof(OUT));
}

// filter_{hpel,qpel}_blocks_{hor_ver,diag}_luma (ipol-generic.c:213-679).  STEP 0..3 in the order search_frac
// calls them (search_inter.c:1142-1166).  The four candidate blocks a step returns are the motion-compensated
// predictions at square[1 + 4*(STEP&1) + j] half-samples (STEP < 2) or quarter-samples around the chosen
// half-sample offset (STEP >= 2) -- identity proven against the reference by tools/refcheck/rc_ipol.inc -- so
// every call is computed from `src` alone; the caller's `hor_intermediate` / `hor_first_cols` scratch (a private
// hand-over between the four functions of one strategy) is left untouched.  `src` = block origin minus the 1-sample
// ME border, inside an extended block with 3 samples before and 4 after (search_inter.c:1085-1120).
constexpr int FME_STRIDE = 64;  // LCU_WIDTH: row stride of filtered[4][LCU_LUMA_SIZE]
template <typename PX, int STEP>
void fme_blocks_hip(const void *, PX *src, int16_t src_stride, int width, int height, PX (*filtered)[FME_STRIDE * FME_STRIDE],
                    void * /*hor_intermediate*/, int8_t /*fme_level*/, void * /*hor_first_cols*/, int8_t hpel_off_x,
                    int8_t hpel_off_y)
{
  static const int8_t square[9][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
  const int rw = width + 8, rh = height + 8;
  const size_t out_bytes = (size_t)4 * width * height * sizeof(PX);
  percall_ctx *c = percall_get((size_t)rw * rh * sizeof(PX) + out_bytes + 2048);
  const size_t os = c->stage_block(src - (ptrdiff_t)3 * src_stride - 3, (size_t)src_stride, rw, rh, sizeof(PX));
  const size_t ob = c->take(4 * sizeof(uvghip_mc_blk_t));
  for (int j = 0; j < 4; ++j) {
    const int8_t *d = square[1 + (STEP & 1) * 4 + j];
    const int mvx = STEP < 2 ? d[0] * 8 : hpel_off_x * 8 + d[0] * 4;   // 1/16 sample units
    const int mvy = STEP < 2 ? d[1] * 8 : hpel_off_y * 8 + d[1] * 4;
    c->hp<uvghip_mc_blk_t>(ob)[j] = uvghip_mc_blk_t{4 + (mvx >> 4), 4 + (mvy >> 4), mvx & 15, mvy & 15};
  }
  c->upload(0, c->used);
  const size_t oo = c->take(out_bytes);
  c->must(uvghip_mc_batch(px_traits<PX>::depth, c->dp<PX>(os), rw, rw, rh, 0, width, height,
                          c->dp<uvghip_mc_blk_t>(ob), 4, 0, c->dp<PX>(oo), c->stream), "fme launch");
  c->download(oo, out_bytes);
  c->sync();
  for (int j = 0; j < 4; ++j)
    for (int y = 0; y < height; ++y)
      memcpy(filtered[j] + (size_t)y * FME_STRIDE, c->hp<PX>(oo) + ((size_t)j * height + y) * width, (size_t)width * sizeof(PX));
}

// get_extended_block / get_extended_block_wraparound (strategies-ipol.h:67-94, ipol-generic.c:761-883).  Inside the
// picture the reference returns pointers into the caller's frame -- pure pointer arithmetic, done here exactly so.
// Otherwise the rows the block can reach (clamped) are staged, the copy is built on the device and lands in args->buf.
static inline int ext_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
template <typename PX, int WRAP>
void get_extended_block_hip(ref_epol_args<PX> *a)
{
  const int min_y = a->blk_y - a->pad_t, max_y = a->blk_y + a->blk_h + a->pad_b + a->pad_b_simd - 1;
  const int min_x = a->blk_x - a->pad_l, max_x = a->blk_x + a->blk_w + a->pad_r - (WRAP ? 0 : 1);
  if (!(min_y < 0 || max_y >= a->src_h || min_x < 0 || max_x >= a->src_w)) {
    *a->ext = a->src + (ptrdiff_t)(a->blk_y - a->pad_t) * a->src_s + (a->blk_x - a->pad_l);
    *a->ext_origin = a->src + (ptrdiff_t)a->blk_y * a->src_s + a->blk_x;
    *a->ext_s = a->src_s;
    return;
  }
  const int es = a->pad_l + a->blk_w + a->pad_r, rows = a->pad_t + a->blk_h + a->pad_b + a->pad_b_simd;
  // stage the source rows the block touches after clamping (whole rows: the wrap-around variant reads both ends)
  const int y_lo = ext_clamp(min_y, 0, a->src_h - 1), y_hi = ext_clamp(a->blk_y + a->blk_h + a->pad_b - 1, 0, a->src_h - 1);
  const int nrows = y_hi - y_lo + 1;
  percall_ctx *c = percall_get((size_t)nrows * a->src_w * sizeof(PX) + (size_t)rows * es * sizeof(PX) + 2048);
  const size_t os = c->stage_block(a->src + (ptrdiff_t)y_lo * a->src_s, (size_t)a->src_s, a->src_w, nrows, sizeof(PX));
  const size_t op = c->take(sizeof(uvghip_tu_t));
  *c->hp<uvghip_tu_t>(op) = uvghip_tu_t{a->blk_x, a->blk_y - y_lo};
  c->upload(0, c->used);
  const size_t oo = c->take((size_t)rows * es * sizeof(PX));
  // the staged slab is a picture of nrows rows whose row 0 is picture row y_lo: clamping to it equals clamping to the picture
  c->must(uvghip_extended_block_batch(px_traits<PX>::depth, c->dp<PX>(os), a->src_w, a->src_w, nrows, WRAP, a->blk_w, a->blk_h,
                                      a->pad_l, a->pad_r, a->pad_t, a->pad_b, a->pad_b_simd, c->dp<uvghip_tu_t>(op), 1,
                                      c->dp<PX>(oo), c->stream), "extended block");
  c->download(oo, (size_t)rows * es * sizeof(PX));
  c->sync();
  memcpy(a->buf, c->hp<PX>(oo), (size_t)rows * es * sizeof(PX));
  *a->ext = a->buf;
  *a->ext_s = es;
  *a->ext_origin = a->buf + (ptrdiff_t)a->pad_t * es + a->pad_l;
}

template <typename PX>
int register_ipol(void *opaque)
{
  int ok = 1;
#define REG(type, fn) ok &= uvghip_do_register(opaque, type, (void *)(fn))
  REG("filter_hpel_blocks_hor_ver_luma", (&fme_blocks_hip<PX, 0>));
  REG("filter_hpel_blocks_diag_luma", (&fme_blocks_hip<PX, 1>));
  REG("filter_qpel_blocks_hor_ver_luma", (&fme_blocks_hip<PX, 2>));
  REG("filter_qpel_blocks_diag_luma", (&fme_blocks_hip<PX, 3>));
  REG("sample_quarterpel_luma", (&sample_hip<PX, 8, false>));
  REG("sample_octpel_chroma", (&sample_hip<PX, 4, false>));
  REG("sample_quarterpel_luma_hi", (&sample_hip<PX, 8, true>));
  REG("sample_octpel_chroma_hi", (&sample_hip<PX, 4, true>));
  REG("get_extended_block", (&get_extended_block_hip<PX, 0>));
  REG("get_extended_block_wraparound", (&get_extended_block_hip<PX, 1>));
#undef REG
  return ok;
}

}  // namespace

extern "C" int uvg_strategy_register_ipol_hip(void *opaque, uint8_t bitdepth)
{
  if (!uvghip_ready() && uvghip_init(0) != 0) return 0;
  return bitdepth == 8 ? register_ipol<uint8_t>(opaque) : register_ipol<uint16_t>(opaque);
}
