// "alf" strategy group on gfx950: adaptive loop filter classification, 7x7 /
// 5x5 diamond filters and covariance statistics over device-resident planes.
// Bit-exact with src/strategies/generic/alf-generic.c:
//   alf_derive_classification_blk  :49-288
//   alf_filter_7x7_blk / 5x5_blk   :290-737
//   alf_calc_covariance + alf_get_blk_stats  :742-999
// Picture borders are handled by coordinate clamping (= the 4-sample border
// replication adjust_pixels does upstream, alf.c:937-1113); the virtual boundary
// sits at row 60 of every 64 luma rows / 30 of 32 chroma rows (alf.h:32-33).
//
// Statistics kernel (the one dense contraction of the path, sum_px e e^T):
// one 128-thread workgroup per CTU.  A strip of 4 rows (<= 256 samples) at a
// time: phase A computes the 13x4 clipped tap sums e[k][b] of every sample
// into LDS (int16), phase B gives each thread one (k,l) pair: per sample two
// 8-byte LDS reads feed 16 v_mad_i32_i24, accumulated in int32 over a 4x4
// block, then in int64 registers while the block class stays the same, and
// flushed to the CTU's [class] slab in HBM on a class change (single owner per
// entry, no atomics).  12-bit operands rule out i8 MFMA without digit
// splitting; at 1456 MAC/sample the kernel is ALU-bound.
#include "uvghip_common.h"

__device__ __forceinline__ int clip_pair(int clip, int ref, int v0, int v1)
{
  return clampi(v0 - ref, -clip, clip) + clampi(v1 - ref, -clip, clip);
}

template <typename PX>
__device__ __forceinline__ int pxc(const PX *p, int stride, int w, int h, int x, int y)
{
  return p[(size_t)clampi(y, 0, h - 1) * stride + clampi(x, 0, w - 1)];
}

// ------------------------------------------------------------------ classification ----
template <typename PX>
__global__ void __launch_bounds__(256)
alf_classify_kernel(const PX *__restrict__ rec, int stride, int w, int h, int shift, uint8_t *__restrict__ cls, int cls_stride,
                    int unit_row0)
{
  const int bx4 = blockIdx.x * blockDim.x + threadIdx.x, by4 = unit_row0 + blockIdx.y;
  const int bx = bx4 * 4, by = by4 * 4;
  if (bx >= w || by >= h) return;
  constexpr int vbh = 64, vb_pos = 60;
  int sv = 0, sh = 0, sd0 = 0, sd1 = 0;
  const int ymod = by & (vbh - 1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if ((ymod == vb_pos - 4 && r == 3) || (ymod == vb_pos && r == 0)) continue;
    const int y = by - 2 + 2 * r;
    int y_up2 = y + 2, y_dn = y - 1;
    if (y > 0 && (y & (vbh - 1)) == vb_pos - 2) y_up2 = y + 1;
    else if (y > 0 && (y & (vbh - 1)) == vb_pos) y_dn = y;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int x = bx - 2 + 2 * c;
#define P(dx, yy) pxc<PX>(rec, stride, w, h, x + (dx), (yy))
      const int p00 = P(0, y), p11 = P(1, y + 1);
      const int y0 = p00 << 1, y1 = p11 << 1;
      const int p10 = P(1, y), p01 = P(0, y + 1);
      sv += abs(y0 - P(0, y_dn) - p01) + abs(y1 - p10 - P(1, y_up2));
      sh += abs(y0 - p10 - P(-1, y)) + abs(y1 - P(2, y + 1) - p01);
      sd0 += abs(y0 - P(-1, y_dn) - p11) + abs(y1 - p00 - P(2, y_up2));
      sd1 += abs(y0 - P(-1, y + 1) - P(1, y_dn)) + abs(y1 - P(0, y_up2) - P(2, y));
#undef P
    }
  }
  const int th[16] = {0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4};
  const int act = clampi(((sv + sh) * ((ymod == vb_pos - 4 || ymod == vb_pos) ? 96 : 64)) >> shift, 0, 15);
  int class_idx = th[act];
  int hv1, hv0, d1, d0, dir_hv, dir_d, hvd1, hvd0, main_dir, sec_dir;
  if (sv > sh) { hv1 = sv; hv0 = sh; dir_hv = 1; } else { hv1 = sh; hv0 = sv; dir_hv = 3; }
  if (sd0 > sd1) { d1 = sd0; d0 = sd1; dir_d = 0; } else { d1 = sd1; d0 = sd0; dir_d = 2; }
  if ((uint32_t)d1 * (uint32_t)hv0 > (uint32_t)hv1 * (uint32_t)d0) { hvd1 = d1; hvd0 = d0; main_dir = dir_d; sec_dir = dir_hv; }
  else { hvd1 = hv1; hvd0 = hv0; main_dir = dir_hv; sec_dir = dir_d; }
  int strength = 0;
  if (hvd1 > 2 * hvd0) strength = 1;
  if (hvd1 * 2 > 9 * hvd0) strength = 2;
  if (strength) class_idx += (((main_dir & 1) << 1) + strength) * 5;
  const int tt[8] = {0, 1, 0, 2, 2, 3, 1, 3};
  cls[by4 * cls_stride + bx4] = (uint8_t)(class_idx | (tt[main_dir * 2 + (sec_dir >> 1)] << 5));
}

static int alf_classify_launch(int bitdepth, const void *rec, int rec_stride, int width, int height, int shift, uint8_t *cls,
                               int cls_stride, int row0, int row1, hipStream_t st, const char *who)
{
  if ((bitdepth != 8 && bitdepth != 10) || width <= 0 || height <= 0 || (width & 3) || (height & 3) || row0 < 0 ||
      row1 > height || row0 >= row1 || (row0 & 3) || (row1 & 3))
    return uvghip_set_error(hipErrorInvalidValue, who);
  constexpr int THREADS = 64;          // one wave per workgroup: four times as many workgroups to spread over the CUs
  dim3 grid((width / 4 + THREADS - 1) / THREADS, (row1 - row0) / 4);
  if (bitdepth == 8) alf_classify_kernel<uint8_t><<<grid, THREADS, 0, st>>>((const uint8_t *)rec, rec_stride, width, height, shift, cls, cls_stride, row0 / 4);
  else alf_classify_kernel<uint16_t><<<grid, THREADS, 0, st>>>((const uint16_t *)rec, rec_stride, width, height, shift, cls, cls_stride, row0 / 4);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_alf_classify_frame(int bitdepth, const void *rec, int rec_stride, int width, int height, int shift,
                                         uint8_t *cls, int cls_stride, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return alf_classify_launch(bitdepth, rec, rec_stride, width, height, shift, cls, cls_stride, 0, height, uvghip_stream(stream), __func__);
}

extern "C" int uvghip_alf_classify_band(int bitdepth, const void *rec, int rec_stride, int width, int height, int shift,
                                        uint8_t *cls, int cls_stride, int row0, int row1, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return alf_classify_launch(bitdepth, rec, rec_stride, width, height, shift, cls, cls_stride, row0, row1, uvghip_stream(stream), __func__);
}

// ------------------------------------------------------------------------- filter ----
__device__ static const int8_t kPerm7[4][13] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12}, {9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6, 12},
                                                {0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11, 12}, {9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6, 12}};

// the same, padded to 16 bytes per transpose: fetched as three words (statistics kernel)
__device__ static const int8_t kPermPad[4][16] __attribute__((aligned(16))) = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0, 0, 0}, {9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6, 12, 0, 0, 0},
    {0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11, 12, 0, 0, 0}, {9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6, 12, 0, 0, 0}};

#define ALF_FILTER_SPLIT 4
// A rectangle (at most 64 x 64) is filtered by ALF_FILTER_SPLIT workgroups, each a band of its rows.  The band's source
// window -- its rows +-3 (+-2 chroma), the rectangle's columns +-3, coordinates clamped to the picture exactly as the
// per-sample fetch of the reference's padded picture would see them -- is staged once in LDS; a thread then filters four
// horizontally adjacent samples (one row of a 4x4 block: same class, same transpose, so the 12 coefficient / clip pairs are
// fetched once per four samples) and writes them with one store.
template <typename PX, bool CHROMA>
__global__ void __launch_bounds__(256)
alf_filter_kernel(const PX *__restrict__ src, int sstride, PX *__restrict__ dst, int dstride, int pic_w, int pic_h,
                  const uvghip_rect_t *__restrict__ rects, const int32_t *__restrict__ set_idx,
                  const int16_t *__restrict__ coef_sets, const int16_t *__restrict__ clip_sets,
                  const uint8_t *__restrict__ cls, int cls_stride)
{
  constexpr int NSET = CHROMA ? 7 : 25 * 13;
  constexpr int vbh = CHROMA ? 32 : 64, vb_pos = CHROMA ? 30 : 60;
  constexpr int DEPTH = px_traits<PX>::depth;
  constexpr int HALF = CHROMA ? 2 : 3;
  constexpr int WP = 64 + 2 * 3 + 2;        // window pitch in samples (72: rows start 16-byte aligned for 16-bit samples)
  constexpr int MAXROWS = 64 / ALF_FILTER_SPLIT + 2 * 3 + 4;
  __shared__ int16_t sCoef[25 * 13], sClip[25 * 13];
  __shared__ PX sWin[MAXROWS * WP];
  const int rect_i = blockIdx.x / ALF_FILTER_SPLIT, part = blockIdx.x % ALF_FILTER_SPLIT;
  const int si = set_idx[rect_i];
  if (si < 0) return;                       // CTU not filtered: dst keeps what it has (alf.c:5088)
  const uvghip_rect_t R = rects[rect_i];
  for (int i = threadIdx.x; i < NSET; i += blockDim.x) { sCoef[i] = coef_sets[(size_t)si * NSET + i]; sClip[i] = clip_sets[(size_t)si * NSET + i]; }
  // the band: whole 4-row block rows
  const int brows = (R.h + 3) >> 2, per = (brows + ALF_FILTER_SPLIT - 1) / ALF_FILTER_SPLIT;
  const int yb0 = min(R.h, part * per * 4), yb1 = min(R.h, (part + 1) * per * 4);
  if (yb0 >= yb1) return;
  const int wrows = yb1 - yb0 + 2 * HALF, wcols = R.w + 2 * HALF;
  for (int i = threadIdx.x; i < wrows * wcols; i += blockDim.x) {
    const int ry = i / wcols, rx = i - ry * wcols;
    sWin[ry * WP + rx] = (PX)pxc<PX>(src, sstride, pic_w, pic_h, R.x - HALF + rx, R.y + yb0 - HALF + ry);
  }
  __syncthreads();
  const int shift = DEPTH - 1, offset = 1 << (shift - 1);
  const int quads = (R.w + 3) >> 2;
  for (int i = threadIdx.x; i < quads * (yb1 - yb0); i += blockDim.x) {
    const int yy = i / quads, xq = (i - yy * quads) * 4;
    const int x = R.x + xq, y = R.y + yb0 + yy;
    const int y_vb = y & (vbh - 1);
    int lim = 3;
    if (y_vb < vb_pos && y_vb >= vb_pos - (CHROMA ? 2 : 4)) lim = vb_pos - 1 - y_vb;
    else if (y_vb >= vb_pos && y_vb <= vb_pos + (CHROMA ? 1 : 3)) lim = y_vb - vb_pos;
    const int r1 = min(1, lim), r2 = min(2, lim), r3 = min(3, lim);
    const bool near_vb = y_vb == vb_pos - 1 || y_vb == vb_pos;
    const PX *W0 = sWin + (yy + HALF) * WP + xq + HALF;          // the first of the four samples inside the window
    int sum[4] = {0, 0, 0, 0}, cur[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = W0[q];
    // one tap pair (coefficient cf, clip cc) at (+ax, +ay) / (-ax, -ay) for the four samples
#define TAP(cf_, cc_, ax, ay)                                                                   \
    {                                                                                           \
      const int cf__ = (cf_), cc__ = (cc_);                                                     \
      const PX *pa = W0 + (ay) * WP + (ax), *pb = W0 - (ay) * WP - (ax);                        \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) sum[q] += cf__ * clip_pair(cc__, cur[q], pa[q], pb[q]); \
    }
    if constexpr (!CHROMA) {
      const int cl = cls[(y >> 2) * cls_stride + (x >> 2)];
      const int16_t *cf = sCoef + (cl & 31) * 13, *cc = sClip + (cl & 31) * 13;
      const int8_t *pm = kPerm7[cl >> 5];
#define T(k, ax, ay, bx_, by_) TAP(cf[pm[k]], cc[pm[k]], ax, ay)
      T(0, 0, r3, 0, -r3);
      T(1, 1, r2, -1, -r2); T(2, 0, r2, 0, -r2); T(3, -1, r2, 1, -r2);
      T(4, 2, r1, -2, -r1); T(5, 1, r1, -1, -r1); T(6, 0, r1, 0, -r1); T(7, -1, r1, 1, -r1); T(8, -2, r1, 2, -r1);
      T(9, 3, 0, -3, 0); T(10, 2, 0, -2, 0); T(11, 1, 0, -1, 0);
#undef T
    } else {
#define T(k, ax, ay, bx_, by_) TAP(sCoef[k], sClip[k], ax, ay)
      T(0, 0, r2, 0, -r2);
      T(1, 1, r1, -1, -r1); T(2, 0, r1, 0, -r1); T(3, -1, r1, 1, -r1);
      T(4, 2, 0, -2, 0); T(5, 1, 0, -1, 0);
#undef T
    }
#undef TAP
    unsigned out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sq = near_vb ? (sum[q] + (1 << (shift + 2))) >> (shift + 3) : (sum[q] + offset) >> shift;
      out[q] = (unsigned)clampi(sq + cur[q], 0, px_traits<PX>::maxv);
    }
    PX *d = dst + (size_t)y * dstride + x;
    if (xq + 4 <= R.w && ((reinterpret_cast<uintptr_t>(d) & (4 * sizeof(PX) - 1)) == 0)) {
      if constexpr (sizeof(PX) == 1) *reinterpret_cast<uint32_t *>(d) = out[0] | out[1] << 8 | out[2] << 16 | out[3] << 24;
      else *reinterpret_cast<uint2 *>(d) = make_uint2(out[0] | out[1] << 16, out[2] | out[3] << 16);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (xq + q < R.w) d[q] = (PX)out[q];
    }
  }
}

extern "C" int uvghip_alf_filter_batch(int bitdepth, const void *src, int src_stride, void *dst, int dst_stride, int pic_w,
                                       int pic_h, int is_chroma, const uvghip_rect_t *rects, const int32_t *set_idx, int n,
                                       const int16_t *coef_sets, const int16_t *clip_sets, const uint8_t *cls,
                                       int cls_stride, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
#define F(PX, C) alf_filter_kernel<PX, C><<<n * ALF_FILTER_SPLIT, 256, 0, st>>>((const PX *)src, src_stride, (PX *)dst, dst_stride, pic_w, pic_h, rects, set_idx, coef_sets, clip_sets, cls, cls_stride)
  if (bitdepth == 8) { if (is_chroma) F(uint8_t, true); else F(uint8_t, false); }
  else { if (is_chroma) F(uint16_t, true); else F(uint16_t, false); }
#undef F
  UVGHIP_CHECK_LAUNCH();
}

// --------------------------------------------------------------------- statistics ----
// The covariance is the one dense contraction of the path: per class, C = [E | d]^T [E | d] with E the
// n_samples x 52 matrix of clipped tap sums and d = org - rec; ee[k][l][b0][b1] = C[4k+b0][4l+b1], y[k][b] = C[4k+b][52],
// pix_acc = C[52][52].  It runs on the i8 matrix cores with exact integer arithmetic: every 12-bit signed entry is
// split as v = 128 * h + l (l = v & 127 in 0..127, h = v >> 7 in -16..16), so
//     C = 16384 * H^T H + 128 * (H^T L + (H^T L)^T) + L^T L
// and the three products are v_mfma_i32_32x32x32_i8 accumulations (|sums| < 2^31 for the <= 4096 samples of a rectangle).
// A and B fragments come from the same [entry][sample] byte layout with the same code, so the contraction does not depend
// on how the hardware orders the 32 k values inside an instruction.  A workgroup owns a rectangle: its 4x4 blocks are
// walked in class order (every class padded to an even number of blocks = whole K = 32 chunks), the int32 accumulators
// are combined into the int64 outputs once per class, and every output entry of the rectangle is written exactly once
// (zeros for absent classes) -- no atomics, no memset of the 540 KB per rectangle.
typedef int alf_v4i __attribute__((ext_vector_type(4)));
typedef int alf_v16i __attribute__((ext_vector_type(16)));
constexpr int ALF_KP = 272;        // bytes per operand row: 256 samples + pad (68 dwords = 4 mod 64: b128 reads of 16 rows tile the banks)
constexpr int ALF_SLOTS = 16;      // 4x4 blocks per phase-A chunk (256 samples, one per thread)

// COMPACT: the output is one uvghip record (UVGHIP_ALF_REC_WORDS int64) per class PRESENT in the rectangle -- the upper
// triangle k <= l of ee (ee[k][l][b0][b1] == ee[l][k][b1][b0]), y, pix_acc -- in class order, plus the rectangle's class
// mask in `present`; absent classes cost no write at all.  `ee` then points at the records, `yv` / `pix` are unused.
template <typename PX, bool CHROMA, bool COMPACT>
__global__ void __launch_bounds__(256)
alf_stats_kernel(const PX *__restrict__ org, int ostride, const PX *__restrict__ rec, int rstride, int pic_w, int pic_h,
                 const uvghip_rect_t *__restrict__ rects, const uint8_t *__restrict__ cls, int cls_stride,
                 long long *__restrict__ ee, int32_t *__restrict__ yv, long long *__restrict__ pix,
                 uint32_t *__restrict__ present)
{
  constexpr int NC = CHROMA ? 7 : 13, NCLS = CHROMA ? 1 : 25;
  constexpr int NE = NC * 4;                   // entries of e; index NE is d = org - rec
  constexpr int NT = CHROMA ? 1 : 2;           // 32-row tiles of the (NE + 1)-square result
  constexpr int NROW = 32 * NT;
  constexpr int vbh = CHROMA ? 32 : 64, vb_pos = CHROMA ? 30 : 60;
  constexpr int DEPTH = px_traits<PX>::depth;
  // tile jobs: hh and ll are symmetric (upper tile triangle), hl needs all tiles
  constexpr int N_SYM = NT * (NT + 1) / 2, N_FULL = NT * NT, N_TILES = 2 * N_SYM + N_FULL;
  __shared__ __attribute__((aligned(16))) int8_t sH[NROW * ALF_KP], sL[NROW * ALF_KP];
  __shared__ int sAcc[N_TILES][32][33];
  __shared__ uint8_t sBlkCls[256];
  __shared__ uint16_t sSlot[256 + 32];         // class-ordered block list, 0xffff = padding slot
  __shared__ uint8_t sSlotCls[256 + 32];
  __shared__ int sCnt[32], sPos[32], sFill[32];
  __shared__ uint8_t sPairK[91], sPairL[91];   // COMPACT: pair index -> (k, l), k <= l
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const uvghip_rect_t R = rects[blockIdx.x];
  long long *E = ee + (size_t)blockIdx.x * NCLS * (COMPACT ? UVGHIP_ALF_REC_WORDS : 13 * 13 * 16);
  int32_t *Y = COMPACT ? nullptr : yv + (size_t)blockIdx.x * NCLS * 13 * 4;
  long long *PA = COMPACT ? nullptr : pix + (size_t)blockIdx.x * NCLS;
  int n_done = 0;                               // COMPACT: records written so far (classes are visited in increasing order)
  int clipv[4];
  clipv[0] = 1 << DEPTH;
#pragma unroll
  for (int i = 1; i < 4; ++i) clipv[i] = 1 << (7 - 2 * i + DEPTH - 8);     // alf.c:5248-5260

  // ---- class-ordered slot list (stable counting sort; every class padded to an even number of blocks) ----
  const int bw = (R.w + 3) / 4, bhh = (R.h + 3) / 4, nblk = bw * bhh;      // <= 256 (rectangles are at most 64x64)
  if (t < 32) { sCnt[t] = 0; sFill[t] = 0; }
  if (t < 91) { int k = 0, base = 0; while (t >= base + 13 - k) { base += 13 - k; ++k; } sPairK[t] = (uint8_t)k; sPairL[t] = (uint8_t)(k + (t - base)); }
  const int half_fp = CHROMA ? 2 : 3;
  const bool interior = R.x >= half_fp && R.y >= half_fp && R.x + R.w + half_fp <= pic_w && R.y + R.h + half_fp <= pic_h;
  for (int i = t; i < NROW * ALF_KP; i += 256) { sH[i] = 0; sL[i] = 0; }   // rows NE+1.. stay zero for good
  __syncthreads();
  for (int i = t; i < nblk; i += 256) {
    const int by = i / bw, bx = i - by * bw;
    const int c = CHROMA ? 0 : (cls[((R.y >> 2) + by) * cls_stride + (R.x >> 2) + bx] & 31);
    sBlkCls[i] = (uint8_t)c;
    atomicAdd(&sCnt[c], 1);
  }
  __syncthreads();
  if (t == 0) { int run = 0; for (int c = 0; c < 32; ++c) { sPos[c] = run; run += (sCnt[c] + 1) & ~1; } }
  __syncthreads();
  const int nslot = sPos[31] + ((sCnt[31] + 1) & ~1);                      // even
  // (the order of the blocks inside a class does not matter: integer sums)
  for (int i = t; i < nblk; i += 256) {
    const int c = sBlkCls[i];
    const int rank = atomicAdd(&sFill[c], 1);
    sSlot[sPos[c] + rank] = (uint16_t)i; sSlotCls[sPos[c] + rank] = (uint8_t)c;
  }
  if (t < 32 && (sCnt[t] & 1)) { sSlot[sPos[t] + sCnt[t]] = 0xffffu; sSlotCls[sPos[t] + sCnt[t]] = (uint8_t)t; }
  __syncthreads();

  // ---- which tiles this wave accumulates: hh on wave 0, ll on wave 1, hl split over waves 2 and 3 ----
  alf_v16i acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) acc[j] = alf_v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // wave 0: H^T H tiles (0,0) (0,1) (1,1) -> sAcc 0..N_SYM-1;  wave 1: L^T L likewise -> N_SYM..;  waves 2, 3: H^T L tile row
  // (wave - 2): tiles (a, 0) (a, 1) -> 2 * N_SYM + a * NT + b.  Everything below branches on the (uniform) wave index
  // explicitly: no run-time indexing into register arrays.
  const int8_t *opA = wave == 1 ? sL : sH;     // A operand digit plane
  const int8_t *opB = wave == 0 ? sH : sL;     // B operand digit plane
  const int hl_row = wave >= 2 ? wave - 2 : 0;
  const bool has_work = wave < 2 || hl_row < NT;

  unsigned done_mask = 0;                       // classes already written
  int cur_cls = -1;
  // combine the int32 tile sums into the class's int64 outputs (all threads) and clear the accumulators
  auto flush = [&]() {
    if (cur_cls >= 0 && has_work) {
      const int s0 = wave < 2 ? wave * N_SYM : 2 * N_SYM + hl_row * NT;
      constexpr int NJ = NT == 1 ? 1 : 3;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (wave >= 2 && j >= NT) break;       // an hl tile row has NT tiles
#pragma unroll
        for (int r = 0; r < 16; ++r) sAcc[s0 + j][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[j][r];
        acc[j] = alf_v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    __syncthreads();
    if (cur_cls >= 0) {
      auto sym = [&](int base, int i, int j2) -> long long {      // symmetric product stored as its upper tile triangle
        int ti = i >> 5, tj = j2 >> 5;
        if (ti > tj) { const int x = i; i = j2; j2 = x; ti = i >> 5; tj = j2 >> 5; }
        const int q = ti == 0 ? tj : NT + tj - 1;                  // (0,0) (0,1) (1,1) -> 0 1 2
        return sAcc[base + q][i & 31][j2 & 31];
      };
      auto full = [&](int i, int j2) -> long long { return sAcc[2 * N_SYM + (i >> 5) * NT + (j2 >> 5)][i & 31][j2 & 31]; };
      auto cval = [&](int i, int j2) -> long long {
        return sym(0, i, j2) * 16384 + (full(i, j2) + full(j2, i)) * 128 + sym(N_SYM, i, j2);
      };
      if constexpr (COMPACT) {
        long long *Rc = E + (size_t)n_done * UVGHIP_ALF_REC_WORDS;
        for (int idx = t; idx < 91 * 16; idx += 256) {
          const int b1 = idx & 3, b0 = (idx >> 2) & 3, p = idx >> 4;
          const int k = sPairK[p], l = sPairL[p];                     // pair p -> (k, l), k <= l: row k holds 13 - k pairs
          Rc[idx] = (l < NC) ? cval(4 * k + b0, 4 * l + b1) : 0;
        }
        int32_t *Yc = reinterpret_cast<int32_t *>(Rc + 91 * 16);
        for (int idx = t; idx < 13 * 4; idx += 256) Yc[idx] = (idx >> 2) < NC ? (int32_t)cval(idx, NE) : 0;
        if (t == 0) { Rc[91 * 16 + 26] = cval(NE, NE); Rc[91 * 16 + 27] = 0; }
        ++n_done;
      } else {
        long long *Ec = E + (size_t)cur_cls * 13 * 13 * 16;
        for (int idx = t; idx < 13 * 13 * 16; idx += 256) {
          const int b1 = idx & 3, b0 = (idx >> 2) & 3, kl = idx >> 4, k = kl / 13, l = kl - k * 13;
          Ec[idx] = (k < NC && l < NC) ? cval(4 * k + b0, 4 * l + b1) : 0;
        }
        for (int idx = t; idx < 13 * 4; idx += 256) Y[cur_cls * 13 * 4 + idx] = (idx >> 2) < NC ? (int32_t)cval(idx, NE) : 0;
        if (t == 0) PA[cur_cls] = cval(NE, NE);
      }
      done_mask |= 1u << cur_cls;
    }
    __syncthreads();
  };

  for (int c0 = 0; c0 < nslot; c0 += ALF_SLOTS) {
    const int ns = min(ALF_SLOTS, nslot - c0);                    // even
    // ---- phase A: one sample per thread: clipped tap-pair sums -> digit planes ----
    // The pairs are visited in the geometric order of the filter (alf_filter_kernel) -- (+ax, +ay) / (-ax, -ay) with the row
    // offsets limited at the virtual boundary -- and the sums of pair g go to the rows of the coefficient that pair feeds
    // under the block's transpose, kPermPad[transpose][g] (alf-generic.c:742-905 enumerates the same pairs per transpose).
    {
      const int j = t >> 4, p = t & 15;
      constexpr int NG = CHROMA ? 6 : 12;
      int pr[NG][4];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int b = 0; b < 4; ++b) pr[g][b] = 0;
      int dval = 0, cur = 0;
      uint32_t pw[3] = {0x03020100u, 0x07060504u, 0x0b0a0908u};          // identity
      if (j < ns) {
        const int blk = sSlot[c0 + j];
        if (blk != 0xffff) {
          const int by = blk / bw, bx = blk - by * bw;
          const int xx = bx * 4 + (p & 3), yy = by * 4 + (p >> 2);
          if (xx < R.w && yy < R.h) {
            const int x = R.x + xx, y = R.y + yy;
            int trv = 0;
            if constexpr (!CHROMA) {
              trv = cls[(y >> 2) * cls_stride + (x >> 2)] >> 5;
              const uint32_t *pm = reinterpret_cast<const uint32_t *>(kPermPad[trv]);
              pw[0] = pm[0]; pw[1] = pm[1]; pw[2] = pm[2];
            }
            (void)trv;
            const int y_vb = y & (vbh - 1);
            int lim = 3;
            if (y_vb < vb_pos && y_vb >= vb_pos - (CHROMA ? 2 : 4)) lim = vb_pos - 1 - y_vb;
            else if (y_vb >= vb_pos && y_vb <= vb_pos + (CHROMA ? 1 : 3)) lim = y_vb - vb_pos;
            const int r1 = min(1, lim), r2 = min(2, lim), r3 = min(3, lim);
            // alf-generic.c:742-905 clamps a pair's row offset at the virtual boundary only where its loop variable is
            // negative; under transposes 1 and 3 the pairs (-1, 2), (-1, 1), (-2, 1) are visited with a positive one and keep
            // their full offsets.  Reproduced.
            const bool odd = CHROMA ? false : (trv & 1) != 0;
            const int q2 = odd ? 2 : r2, q1 = odd ? 1 : r1;
            // every sample of the footprint first (one batch of loads in flight), then the arithmetic
            constexpr int NP = 2 * NG + 1;
            int sv[NP];
            int dxs[NG], dys[NG];
            if constexpr (!CHROMA) {
              const int tx[12] = {0, 1, 0, -1, 2, 1, 0, -1, -2, 3, 2, 1};
              const int ty[12] = {r3, r2, r2, q2, r1, r1, r1, q1, q1, 0, 0, 0};
#pragma unroll
              for (int g = 0; g < 12; ++g) { dxs[g] = tx[g]; dys[g] = ty[g]; }
            } else {
              const int tx[6] = {0, 1, 0, -1, 2, 1};
              const int ty[6] = {r2, r1, r1, r1, 0, 0};
#pragma unroll
              for (int g = 0; g < 6; ++g) { dxs[g] = tx[g]; dys[g] = ty[g]; }
              (void)r3; (void)q2; (void)q1;
            }
            if (interior) {
              const PX *c0 = rec + (size_t)y * rstride + x;
              sv[2 * NG] = c0[0];
#pragma unroll
              for (int g = 0; g < NG; ++g) { const int o = dys[g] * rstride + dxs[g]; sv[2 * g] = c0[o]; sv[2 * g + 1] = c0[-o]; }
            } else {
              sv[2 * NG] = pxc<PX>(rec, rstride, pic_w, pic_h, x, y);
#pragma unroll
              for (int g = 0; g < NG; ++g) {
                sv[2 * g] = pxc<PX>(rec, rstride, pic_w, pic_h, x + dxs[g], y + dys[g]);
                sv[2 * g + 1] = pxc<PX>(rec, rstride, pic_w, pic_h, x - dxs[g], y - dys[g]);
              }
            }
            cur = sv[2 * NG];
            dval = (int)org[(size_t)y * ostride + x] - cur;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              const int d0 = sv[2 * g] - cur, d1 = sv[2 * g + 1] - cur;
              pr[g][0] = d0 + d1;                      // clipv[0] = 1 << depth never clips a sample difference
#pragma unroll
              for (int b = 1; b < 4; ++b) pr[g][b] = clampi(d0, -clipv[b], clipv[b]) + clampi(d1, -clipv[b], clipv[b]);
            }
          }
        }
      }
      if (j < ns) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int k = (int)((pw[g >> 2] >> (8 * (g & 3))) & 255);
          int8_t *ph = sH + (4 * k) * ALF_KP + t, *pl = sL + (4 * k) * ALF_KP + t;
#pragma unroll
          for (int b = 0; b < 4; ++b) { ph[b * ALF_KP] = (int8_t)(pr[g][b] >> 7); pl[b * ALF_KP] = (int8_t)(pr[g][b] & 127); }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {                                     // the centre coefficient: the sample itself
          sH[(4 * (NC - 1) + b) * ALF_KP + t] = (int8_t)(cur >> 7); sL[(4 * (NC - 1) + b) * ALF_KP + t] = (int8_t)(cur & 127);
        }
        sH[NE * ALF_KP + t] = (int8_t)(dval >> 7); sL[NE * ALF_KP + t] = (int8_t)(dval & 127);
      }
    }
    __syncthreads();
    // ---- phase B: K = 32 samples (two slots of one class) per step ----
    for (int q = 0; q < ns; q += 2) {
      const int c = sSlotCls[c0 + q];
      if (c != cur_cls) { flush(); cur_cls = c; }
      const int koff = q * 16 + 16 * (lane >> 5);
      auto frag = [&](const int8_t *plane, int tile) {
        return *reinterpret_cast<const alf_v4i *>(plane + (tile * 32 + (lane & 31)) * ALF_KP + koff);
      };
      if (wave < 2) {                           // symmetric products: the B fragments are the A fragments
        const alf_v4i f0 = frag(opA, 0);
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0, f0, acc[0], 0, 0, 0);
        if constexpr (NT == 2) {
          const alf_v4i f1 = frag(opA, 1);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0, f1, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1, f1, acc[2], 0, 0, 0);
        }
      } else if (has_work) {
        const alf_v4i fa = frag(opA, hl_row), fb0 = frag(opB, 0);
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb0, acc[0], 0, 0, 0);
        if constexpr (NT == 2) {
          const alf_v4i fb1 = frag(opB, 1);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb1, acc[1], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  flush();
  cur_cls = -1;
  if constexpr (COMPACT) {
    if (t == 0) present[blockIdx.x] = done_mask;
    return;
  }
  // classes without a block in this rectangle: zeros
  for (int c = 0; c < NCLS; ++c) {
    if (done_mask >> c & 1) continue;
    for (int idx = t; idx < 13 * 13 * 16; idx += 256) E[(size_t)c * 13 * 13 * 16 + idx] = 0;
    if (t < 13 * 4) Y[c * 13 * 4 + t] = 0;
    if (t == 0) PA[c] = 0;
  }
}

extern "C" int uvghip_alf_stats_batch(int bitdepth, const void *org, int org_stride, const void *rec, int rec_stride, int pic_w,
                                      int pic_h, int is_chroma, const uvghip_rect_t *rects, int n, const uint8_t *cls,
                                      int cls_stride, int64_t *ee, int32_t *y, int64_t *pix_acc, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (bitdepth != 8 && bitdepth != 10) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
#define K(PX, C) alf_stats_kernel<PX, C, false><<<n, 256, 0, st>>>((const PX *)org, org_stride, (const PX *)rec, rec_stride, pic_w, pic_h, rects, cls, cls_stride, (long long *)ee, y, (long long *)pix_acc, nullptr)
  if (bitdepth == 8) { if (is_chroma) K(uint8_t, true); else K(uint8_t, false); }
  else { if (is_chroma) K(uint16_t, true); else K(uint16_t, false); }
#undef K
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_alf_stats_compact_batch(int bitdepth, const void *org, int org_stride, const void *rec, int rec_stride,
                                              int pic_w, int pic_h, int is_chroma, const uvghip_rect_t *rects, int n,
                                              const uint8_t *cls, int cls_stride, int64_t *records, uint32_t *present,
                                              void *stream)
{
  UVGHIP_REQUIRE_READY();
  if ((bitdepth != 8 && bitdepth != 10) || !records || !present) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
#define K(PX, C) alf_stats_kernel<PX, C, true><<<n, 256, 0, st>>>((const PX *)org, org_stride, (const PX *)rec, rec_stride, pic_w, pic_h, rects, cls, cls_stride, (long long *)records, nullptr, nullptr, present)
  if (bitdepth == 8) { if (is_chroma) K(uint8_t, true); else K(uint8_t, false); }
  else { if (is_chroma) K(uint16_t, true); else K(uint16_t, false); }
#undef K
  UVGHIP_CHECK_LAUNCH();
}

// ---- consumers of the compact records ----
__device__ __forceinline__ int alf_tri_index(int k, int l) { return k * 13 - (k * (k - 1)) / 2 + (l - k); }   // k <= l

// The reference layout of alf_covariance (alf.h:176-182) back from the records: one workgroup per (rectangle, class).
__global__ void __launch_bounds__(256)
alf_cov_expand_kernel(const long long *__restrict__ records, const uint32_t *__restrict__ present, int ncls,
                      long long *__restrict__ ee, int32_t *__restrict__ yv, long long *__restrict__ pix)
{
  const int r = blockIdx.x / ncls, c = blockIdx.x - r * ncls, t = threadIdx.x;
  const uint32_t mask = present[r];
  const bool has = (mask >> c) & 1;
  const long long *R = records + ((size_t)r * ncls + __popc(mask & ((1u << c) - 1))) * UVGHIP_ALF_REC_WORDS;
  long long *E = ee + ((size_t)r * ncls + c) * 13 * 13 * 16;
  for (int idx = t; idx < 13 * 13 * 16; idx += 256) {
    const int b1 = idx & 3, b0 = (idx >> 2) & 3, kl = idx >> 4, k = kl / 13, l = kl - k * 13;
    long long v = 0;
    if (has) v = k <= l ? R[alf_tri_index(k, l) * 16 + b0 * 4 + b1] : R[alf_tri_index(l, k) * 16 + b1 * 4 + b0];
    E[idx] = v;
  }
  const int32_t *Yr = reinterpret_cast<const int32_t *>(R + 91 * 16);
  if (t < 52) yv[((size_t)r * ncls + c) * 52 + t] = has ? Yr[t] : 0;
  if (t == 0) pix[(size_t)r * ncls + c] = has ? R[91 * 16 + 26] : 0;
}

extern "C" int uvghip_alf_cov_expand(const int64_t *records, const uint32_t *present, int n, int is_chroma, int64_t *ee,
                                     int32_t *y, int64_t *pix_acc, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (n <= 0) return 0;
  const int ncls = is_chroma ? 1 : 25;
  alf_cov_expand_kernel<<<n * ncls, 256, 0, uvghip_stream(stream)>>>((const long long *)records, present, ncls, (long long *)ee, y, (long long *)pix_acc);
  UVGHIP_CHECK_LAUNCH();
}

// Frame-level sums per class (what alf.c:792-835 accumulates over the CTUs before deriving the filters), as
// UVGHIP_ALF_SUM_WORDS int64 per class: the ee triangle, then y widened to int64, then pix_acc.  A workgroup owns
// (class, a slice of the rectangles); slices meet through 64-bit atomics on the zeroed output.
// (a slice is a short serial chain of dependent loads -- mask, then the record it selects: many short slices, not few long ones)
constexpr int ALF_REDUCE_PER_SLICE = 12;
__global__ void __launch_bounds__(256)
alf_cov_reduce_kernel(const long long *__restrict__ records, const uint32_t *__restrict__ present, int n, int ncls,
                      unsigned long long *__restrict__ sums)
{
  const int c = blockIdx.x, slice = blockIdx.y, t = threadIdx.x;
  const int per = (n + (int)gridDim.y - 1) / (int)gridDim.y;
  const int r0 = slice * per, r1 = min(n, r0 + per);
  constexpr int NV = (UVGHIP_ALF_SUM_WORDS + 255) / 256;
  long long acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0;
  bool any = false;
  for (int r = r0; r < r1; ++r) {
    const uint32_t mask = present[r];
    if (!((mask >> c) & 1)) continue;
    any = true;
    const long long *R = records + ((size_t)r * ncls + __popc(mask & ((1u << c) - 1))) * UVGHIP_ALF_REC_WORDS;
    const int32_t *Yr = reinterpret_cast<const int32_t *>(R + 91 * 16);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + 256 * j;
      if (idx < 91 * 16) acc[j] += R[idx];
      else if (idx < 91 * 16 + 52) acc[j] += Yr[idx - 91 * 16];
      else if (idx == 91 * 16 + 52) acc[j] += R[91 * 16 + 26];
    }
  }
  if (!any) return;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = t + 256 * j;
    if (idx < UVGHIP_ALF_SUM_WORDS && acc[j]) atomicAdd(&sums[(size_t)c * UVGHIP_ALF_SUM_WORDS + idx], (unsigned long long)acc[j]);
  }
}

extern "C" int uvghip_alf_cov_reduce(const int64_t *records, const uint32_t *present, int n, int is_chroma, int64_t *sums,
                                     void *stream)
{
  UVGHIP_REQUIRE_READY();
  const int ncls = is_chroma ? 1 : 25;
  hipStream_t st = uvghip_stream(stream);
  UVGHIP_TRY(hipMemsetAsync(sums, 0, (size_t)ncls * UVGHIP_ALF_SUM_WORDS * 8, st));
  if (n <= 0) return 0;
  const int slices = (n + ALF_REDUCE_PER_SLICE - 1) / ALF_REDUCE_PER_SLICE;
  alf_cov_reduce_kernel<<<dim3(ncls, slices), 256, 0, st>>>((const long long *)records, present, n, ncls, (unsigned long long *)sums);
  UVGHIP_CHECK_LAUNCH();
}
