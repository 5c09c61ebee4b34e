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
// split as v = 256 * h + l (l = (int8)v in -128..127, h = (v + 128) >> 8 in -8..8), so
//     C = 65536 * H^T H + 256 * (H^T L + (H^T L)^T) + L^T L
// and the three products are v_mfma_i32_32x32x32_i8 accumulations (|sums| < 2^31 for the <= 4096 samples of a rectangle).
// A and B fragments come from the same [entry][sample] byte layout with the same code, so the contraction does not depend
// on how the hardware orders the 32 k values inside an instruction.  A workgroup owns a rectangle: its 4x4 blocks are
// walked in class order (every class padded to a multiple of four blocks = whole K = 64 groups), the int32 accumulators
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
__global__ void __launch_bounds__(256, 3)
alf_stats_kernel(const PX *__restrict__ org, int ostride, const PX *__restrict__ rec, int rstride, int pic_w, int pic_h,
                 const uvghip_rect_t *__restrict__ rects, const uint8_t *__restrict__ cls, int cls_stride,
                 long long *__restrict__ ee, int32_t *__restrict__ yv, long long *__restrict__ pix,
                 uint32_t *__restrict__ present)
{
  constexpr int NC = CHROMA ? 7 : 13, NCLS = CHROMA ? 1 : 25;
  constexpr int NE = NC * 4;                   // entries of e; index NE is d = org - rec
  constexpr int NT = CHROMA ? 1 : 2;           // 32-row tiles of the (NE + 1)-square result
  constexpr int vbh = CHROMA ? 32 : 64, vb_pos = CHROMA ? 30 : 60;
  constexpr int DEPTH = px_traits<PX>::depth;
  // tile jobs: hh and ll are symmetric (upper tile triangle: NT (NT + 1) / 2 tiles each), hl needs all NT * NT tiles
  // digit planes: rows 0..NE hold data, the rest of the last 32-row tile reads as zero -- one shared zero row (the last) instead
  // of 11; the int32 tiles are combined per output tile pair (four tiles at a time): 49 KB per workgroup, three per CU
  constexpr int NROW_LDS = NT == 2 ? 56 : 32;
  constexpr int N_ACC = NT == 2 ? 4 : 3;
  __shared__ __attribute__((aligned(16))) int8_t sHL[2 * NROW_LDS * ALF_KP];   // the two digit planes, H then L
  int8_t *const sH = sHL, *const sL = sHL + NROW_LDS * ALF_KP;
  __shared__ int sAcc[N_ACC][32][33];
  __shared__ uint8_t sBlkCls[256];
  __shared__ uint16_t sSlot[256 + 96];         // class-ordered block list, 0xffff = padding slot (every class to a multiple of 4)
  __shared__ uint8_t sSlotCls[256 + 96];
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

  // ---- class-ordered slot list (counting sort; every class padded to a multiple of four blocks) ----
  const int bw = (R.w + 3) / 4, bhh = (R.h + 3) / 4, nblk = bw * bhh;      // <= 256 (rectangles are at most 64x64)
  if (t < 32) { sCnt[t] = 0; sFill[t] = 0; }
  if (t < 91) { int k = 0, base = 0; while (t >= base + 13 - k) { base += 13 - k; ++k; } sPairK[t] = (uint8_t)k; sPairL[t] = (uint8_t)(k + (t - base)); }
  const int half_fp = CHROMA ? 2 : 3;
  const bool interior = R.x >= half_fp && R.y >= half_fp && R.x + R.w + half_fp <= pic_w && R.y + R.h + half_fp <= pic_h;
  for (int i = t; i < 2 * NROW_LDS * ALF_KP; i += 256) sHL[i] = 0;   // rows NE+1.. stay zero for good
  __syncthreads();
  for (int i = t; i < nblk; i += 256) {
    const int by = i / bw, bx = i - by * bw;
    const int c = CHROMA ? 0 : (cls[((R.y >> 2) + by) * cls_stride + (R.x >> 2) + bx] & 31);
    sBlkCls[i] = (uint8_t)c;
    atomicAdd(&sCnt[c], 1);
  }
  __syncthreads();
  if (t == 0) { int run = 0; for (int c = 0; c < 32; ++c) { sPos[c] = run; run += (sCnt[c] + 3) & ~3; } }
  __syncthreads();
  const int nslot = sPos[31] + ((sCnt[31] + 3) & ~3);                      // a multiple of 4
  // (the order of the blocks inside a class does not matter: integer sums)
  for (int i = t; i < nblk; i += 256) {
    const int c = sBlkCls[i];
    const int rank = atomicAdd(&sFill[c], 1);
    sSlot[sPos[c] + rank] = (uint16_t)i; sSlotCls[sPos[c] + rank] = (uint8_t)c;
  }
  if (t < 32)
    for (int k = sCnt[t]; k < ((sCnt[t] + 3) & ~3); ++k) { sSlot[sPos[t] + k] = 0xffffu; sSlotCls[sPos[t] + k] = (uint8_t)t; }
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
    if (cur_cls < 0) return;                        // (uniform) nothing accumulated yet
    constexpr int NPAIR = NT == 2 ? 3 : 1;          // output tile pairs (0,0) (0,1) (1,1)
    long long *Rc = nullptr, *Ec = nullptr;
    if (cur_cls >= 0) {
      if constexpr (COMPACT) Rc = E + (size_t)n_done * UVGHIP_ALF_REC_WORDS;
      else Ec = E + (size_t)cur_cls * 13 * 13 * 16;
    }
#pragma unroll
    for (int pq = 0; pq < NPAIR; ++pq) {
      const int pa = pq == 2 ? 1 : 0, pb = pq == 0 ? 0 : 1;
      auto put = [&](const alf_v16i &v, int dst) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sAcc[dst][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = v[r];
      };
      if (cur_cls >= 0) {
        // sAcc[0] = hh(pa,pb), [1] = ll(pa,pb), [2] = hl(pa,pb), [3] = hl(pb,pa) where pa != pb
        if (wave == 0) put(acc[NT == 2 ? pq : 0], 0);
        else if (wave == 1) put(acc[NT == 2 ? pq : 0], 1);
        else if (wave == 2) { if (pa == 0) put(acc[NT == 2 ? pb : 0], 2); }
        else if (NT == 2) { if (pa == 1) put(acc[1], 2); else if (pb == 1) put(acc[0], 3); }
      }
      __syncthreads();
      if (cur_cls >= 0) {
        // entry (i, j2) of the (NE + 1)-square result, symmetric: looked up with its lower tile first
        auto in_pair = [&](int i, int j2) { const int ti = i >> 5, tj = j2 >> 5; return min(ti, tj) == pa && max(ti, tj) == pb; };
        auto cval = [&](int i, int j2) -> long long {
          if ((i >> 5) > (j2 >> 5)) { const int x = i; i = j2; j2 = x; }
          const int a = i & 31, b = j2 & 31;
          const long long hl = (long long)sAcc[2][a][b] + (long long)(pa == pb ? sAcc[2][b][a] : sAcc[N_ACC - 1][b][a]);
          return (long long)sAcc[0][a][b] * 65536 + hl * 256 + (long long)sAcc[1][a][b];
        };
        if constexpr (COMPACT) {
          for (int idx = t; idx < 91 * 16; idx += 256) {
            const int b1 = idx & 3, b0 = (idx >> 2) & 3, p = idx >> 4;
            const int k = sPairK[p], l = sPairL[p];                     // pair p -> (k, l), k <= l: row k holds 13 - k pairs
            if (l >= NC) { if (pq == 0) Rc[idx] = 0; continue; }
            if (in_pair(4 * k + b0, 4 * l + b1)) Rc[idx] = cval(4 * k + b0, 4 * l + b1);
          }
          int32_t *Yc = reinterpret_cast<int32_t *>(Rc + 91 * 16);
          for (int idx = t; idx < 13 * 4; idx += 256) {
            if ((idx >> 2) >= NC) { if (pq == 0) Yc[idx] = 0; continue; }
            if (in_pair(idx, NE)) Yc[idx] = (int32_t)cval(idx, NE);
          }
          if (t == 0 && in_pair(NE, NE)) { Rc[91 * 16 + 26] = cval(NE, NE); Rc[91 * 16 + 27] = 0; }
        } else {
          for (int idx = t; idx < 13 * 13 * 16; idx += 256) {
            const int b1 = idx & 3, b0 = (idx >> 2) & 3, kl = idx >> 4, k = kl / 13, l = kl - k * 13;
            if (k >= NC || l >= NC) { if (pq == 0) Ec[idx] = 0; continue; }
            if (in_pair(4 * k + b0, 4 * l + b1)) Ec[idx] = cval(4 * k + b0, 4 * l + b1);
          }
          for (int idx = t; idx < 13 * 4; idx += 256) {
            if ((idx >> 2) >= NC) { if (pq == 0) Y[cur_cls * 13 * 4 + idx] = 0; continue; }
            if (in_pair(idx, NE)) Y[cur_cls * 13 * 4 + idx] = (int32_t)cval(idx, NE);
          }
          if (t == 0 && in_pair(NE, NE)) PA[cur_cls] = cval(NE, NE);
        }
      }
      __syncthreads();
    }
    if (cur_cls >= 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = alf_v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (COMPACT) ++n_done;
      done_mask |= 1u << cur_cls;
    }
  };

  for (int c0 = 0; c0 < nslot; c0 += ALF_SLOTS) {
    const int ns = min(ALF_SLOTS, nslot - c0);                    // a multiple of 4
    // ---- phase A: one sample per thread: clipped tap-pair sums -> digit planes ----
    // The pairs are visited in the geometric order of the filter (alf_filter_kernel) -- (+ax, +ay) / (-ax, -ay) with the row
    // offsets limited at the virtual boundary -- and the sums of pair g go to the rows of the coefficient that pair feeds
    // under the block's transpose, kPermPad[transpose][g] (alf-generic.c:742-905 enumerates the same pairs per transpose).
    {
      // wave w, group a (four slots of one class), sample p of the block: slot 4a + w of the chunk, K column 64a + 4p + w:
      // the four slots of a group fill one contiguous K block of 64 for the matrix cores
      const int j = 4 * ((t >> 4) & 3) + (t >> 6), p = t & 15;
      const int kcol = 64 * ((t >> 4) & 3) + 4 * p + (t >> 6);
      constexpr int NG = CHROMA ? 6 : 12;
      constexpr int PL = NROW_LDS * ALF_KP;                               // sHL: H plane, then L plane
      // digits of v = 256 * h + l, l = (int8)v, h = (v + 128) >> 8: the low byte is stored as it is, the high digit is
      // byte 2 of (v + 128) << 8 (one v_add_lshl, stored with ds_write_b8_d16_hi)
      auto put = [&](int row, int v) {
        int8_t *q = sHL + row * ALF_KP + kcol;
        q[0] = (int8_t)(((uint32_t)(v + 128) << 8) >> 16);
        q[PL] = (int8_t)v;
      };
      bool live = false;
      int x = 0, y = 0;
      if (j < ns) {
        const int blk = sSlot[c0 + j];
        if (blk != 0xffff) {
          const int by = blk / bw, bx = blk - by * bw;
          const int xx = bx * 4 + (p & 3), yy = by * 4 + (p >> 2);
          if (xx < R.w && yy < R.h) { live = true; x = R.x + xx; y = R.y + yy; }
        }
      }
      if (live) {
        uint32_t pw[3] = {0x03020100u, 0x07060504u, 0x0b0a0908u};          // identity
        int trv = 0;
        if constexpr (!CHROMA) {
          trv = cls[(y >> 2) * cls_stride + (x >> 2)] >> 5;
          const uint32_t *pm = reinterpret_cast<const uint32_t *>(kPermPad[trv]);
          pw[0] = pm[0]; pw[1] = pm[1]; pw[2] = pm[2];
        }
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
        int sv[2 * NG + 1];
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
          // 32-bit element offsets from the plane base (one scalar base + per-lane offset per load; the five distinct row
          // offsets are multiplied out once)
          const uint32_t base = (uint32_t)(y * rstride + x);
          sv[2 * NG] = rec[base];
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int o = dys[g] * rstride + dxs[g];
            sv[2 * g] = rec[base + (uint32_t)o]; sv[2 * g + 1] = rec[base - (uint32_t)o];
          }
        } else {
          sv[2 * NG] = pxc<PX>(rec, rstride, pic_w, pic_h, x, y);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            sv[2 * g] = pxc<PX>(rec, rstride, pic_w, pic_h, x + dxs[g], y + dys[g]);
            sv[2 * g + 1] = pxc<PX>(rec, rstride, pic_w, pic_h, x - dxs[g], y - dys[g]);
          }
        }
        const int cur = sv[2 * NG];
        const int dval = (int)org[(uint32_t)(y * ostride + x)] - cur;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int k = (int)((pw[g >> 2] >> (8 * (g & 3))) & 255);
          const int d0 = sv[2 * g] - cur, d1 = sv[2 * g + 1] - cur;
          put(4 * k, d0 + d1);                       // clipv[0] = 1 << depth never clips a sample difference
#pragma unroll
          for (int b = 1; b < 4; ++b) put(4 * k + b, clampi(d0, -clipv[b], clipv[b]) + clampi(d1, -clipv[b], clipv[b]));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) put(4 * (NC - 1) + b, cur);            // the centre coefficient: the sample itself
        put(NE, dval);
      } else if (j < ns) {                                                // padding slot / outside the rectangle: a zero column
#pragma unroll
        for (int q = 0; q <= NE; ++q) { sHL[q * ALF_KP + kcol] = 0; sHL[PL + q * ALF_KP + kcol] = 0; }
      }
    }
    __syncthreads();
    // ---- phase B: a group (four slots of one class) = a K block of 64 = two matrix-core steps of K = 32; the fragments of
    //      both steps are fetched before the first product is issued ----
    for (int q = 0; q < ns; q += 4) {
      const int c = sSlotCls[c0 + q];
      if (c != cur_cls) { flush(); cur_cls = c; }
      const int koff = 64 * (q >> 2) + 16 * (lane >> 5);          // first half of the group's K block; the second is + 32
      auto frag = [&](const int8_t *plane, int tile, int half) {
        const int row = min(tile * 32 + (lane & 31), NROW_LDS - 1);
        return *reinterpret_cast<const alf_v4i *>(plane + row * ALF_KP + koff + 32 * half);
      };
      if (wave < 2) {                           // symmetric products: the B fragments are the A fragments
        const alf_v4i f0a = frag(opA, 0, 0), f0b = frag(opA, 0, 1);
        if constexpr (NT == 2) {
          const alf_v4i f1a = frag(opA, 1, 0), f1b = frag(opA, 1, 1);
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0a, f0a, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0a, f1a, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1a, f1a, acc[2], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0b, f0b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0b, f1b, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1b, f1b, acc[2], 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0a, f0a, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0b, f0b, acc[0], 0, 0, 0);
        }
      } else if (has_work) {
        const alf_v4i faa = frag(opA, hl_row, 0), fab = frag(opA, hl_row, 1), fb0a = frag(opB, 0, 0), fb0b = frag(opB, 0, 1);
        if constexpr (NT == 2) {
          const alf_v4i fb1a = frag(opB, 1, 0), fb1b = frag(opB, 1, 1);
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(faa, fb0a, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(faa, fb1a, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fab, fb0b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fab, fb1b, acc[1], 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(faa, fb0a, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fab, fb0b, acc[0], 0, 0, 0);
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
