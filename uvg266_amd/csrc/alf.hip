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
alf_classify_kernel(const PX *__restrict__ rec, int stride, int w, int h, int shift, uint8_t *__restrict__ cls, int cls_stride)
{
  const int bx4 = blockIdx.x * blockDim.x + threadIdx.x, by4 = blockIdx.y;
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

extern "C" int uvghip_alf_classify_frame(int bitdepth, const void *rec, int rec_stride, int width, int height, int shift,
                                         uint8_t *cls, int cls_stride, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (width <= 0 || height <= 0 || (width & 3) || (height & 3)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  dim3 grid((width / 4 + 255) / 256, height / 4);
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) alf_classify_kernel<uint8_t><<<grid, 256, 0, st>>>((const uint8_t *)rec, rec_stride, width, height, shift, cls, cls_stride);
  else alf_classify_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t *)rec, rec_stride, width, height, shift, cls, cls_stride);
  UVGHIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------- filter ----
__device__ static const int8_t kPerm7[4][13] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12}, {9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6, 12},
                                                {0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11, 12}, {9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6, 12}};

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
  __shared__ int16_t sCoef[25 * 13], sClip[25 * 13];
  const int si = set_idx[blockIdx.x];
  if (si < 0) return;                       // CTU not filtered: dst keeps what it has (alf.c:5088)
  const uvghip_rect_t R = rects[blockIdx.x];
  for (int i = threadIdx.x; i < NSET; i += blockDim.x) { sCoef[i] = coef_sets[(size_t)si * NSET + i]; sClip[i] = clip_sets[(size_t)si * NSET + i]; }
  __syncthreads();
  const int shift = DEPTH - 1, offset = 1 << (shift - 1);
  for (int i = threadIdx.x; i < R.w * R.h; i += blockDim.x) {
    const int yy = i / R.w, x = R.x + (i - yy * R.w), y = R.y + yy;
    const int y_vb = y & (vbh - 1);
    int lim = 3;
    if (y_vb < vb_pos && y_vb >= vb_pos - (CHROMA ? 2 : 4)) lim = vb_pos - 1 - y_vb;
    else if (y_vb >= vb_pos && y_vb <= vb_pos + (CHROMA ? 1 : 3)) lim = y_vb - vb_pos;
    const int r1 = min(1, lim), r2 = min(2, lim), r3 = min(3, lim);
    const bool near_vb = y_vb == vb_pos - 1 || y_vb == vb_pos;
#define S(dx, dy) pxc<PX>(src, sstride, pic_w, pic_h, x + (dx), y + (dy))
    const int cur = S(0, 0);
    int sum = 0;
    if constexpr (!CHROMA) {
      const int cl = cls[(y >> 2) * cls_stride + (x >> 2)];
      const int16_t *cf = sCoef + (cl & 31) * 13, *cc = sClip + (cl & 31) * 13;
      const int8_t *pm = kPerm7[cl >> 5];
#define T(k, ax, ay, bx_, by_) sum += cf[pm[k]] * clip_pair(cc[pm[k]], cur, S(ax, ay), S(bx_, by_))
      T(0, 0, r3, 0, -r3);
      T(1, 1, r2, -1, -r2); T(2, 0, r2, 0, -r2); T(3, -1, r2, 1, -r2);
      T(4, 2, r1, -2, -r1); T(5, 1, r1, -1, -r1); T(6, 0, r1, 0, -r1); T(7, -1, r1, 1, -r1); T(8, -2, r1, 2, -r1);
      T(9, 3, 0, -3, 0); T(10, 2, 0, -2, 0); T(11, 1, 0, -1, 0);
#undef T
    } else {
#define T(k, ax, ay, bx_, by_) sum += sCoef[k] * clip_pair(sClip[k], cur, S(ax, ay), S(bx_, by_))
      T(0, 0, r2, 0, -r2);
      T(1, 1, r1, -1, -r1); T(2, 0, r1, 0, -r1); T(3, -1, r1, 1, -r1);
      T(4, 2, 0, -2, 0); T(5, 1, 0, -1, 0);
#undef T
    }
#undef S
    sum = near_vb ? (sum + (1 << (shift + 2))) >> (shift + 3) : (sum + offset) >> shift;
    dst[(size_t)y * dstride + x] = (PX)clampi(sum + cur, 0, px_traits<PX>::maxv);
  }
}

extern "C" int uvghip_alf_filter_batch(int bitdepth, const void *src, int src_stride, void *dst, int dst_stride, int pic_w,
                                       int pic_h, int is_chroma, const uvghip_rect_t *rects, const int32_t *set_idx, int n,
                                       const int16_t *coef_sets, const int16_t *clip_sets, const uint8_t *cls,
                                       int cls_stride, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
#define F(PX, C) alf_filter_kernel<PX, C><<<n, 256, 0, st>>>((const PX *)src, src_stride, (PX *)dst, dst_stride, pic_w, pic_h, rects, set_idx, coef_sets, clip_sets, cls, cls_stride)
  if (bitdepth == 8) { if (is_chroma) F(uint8_t, true); else F(uint8_t, false); }
  else { if (is_chroma) F(uint16_t, true); else F(uint16_t, false); }
#undef F
  UVGHIP_CHECK_LAUNCH();
}

// --------------------------------------------------------------------- statistics ----
// e[k][b] of one sample (alf-generic.c:742-905).  pat maps the tap visiting order of each
// transpose to the coefficient index; center is the last coefficient.
template <typename PX, bool CHROMA>
__device__ inline void covariance_sample(int16_t *e /*[13][4]*/, const PX *rec, int stride, int pic_w, int pic_h, int x, int y,
                                         int tr, int vb_distance, const int *clipv)
{
  constexpr int half = CHROMA ? 2 : 3;
  constexpr int NC = CHROMA ? 7 : 13;
  int top = -4, bot = 4;
  if (vb_distance >= -3 && vb_distance < 0) { bot = -vb_distance - 1; top = -bot; }
  else if (vb_distance >= 0 && vb_distance < 3) { top = -vb_distance; bot = -top; }
  int acc[NC][4];
#pragma unroll
  for (int k = 0; k < NC; ++k)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[k][b] = 0;
  const int cur = pxc<PX>(rec, stride, pic_w, pic_h, x, y);
  // Visit the upper half of the diamond in the canonical (transpose 0) order; the coefficient a tap feeds
  // under transpose t is found by mapping the tap position instead of re-ordering the loops:
  //   t=1: (i,j) -> (j,i)   t=2: (i,j) -> (i,-j)   t=3: (i,j) -> (j,-i) composed as in the reference loops.
  // Equivalent formulation used here: enumerate coefficient slots k in the order the reference's loops
  // for transpose t produce them, and compute which sample pair each slot reads.
  int k = 0;
  auto add = [&](int kk, int dx0, int dy0, int dx1, int dy1) {
    const int a = pxc<PX>(rec, stride, pic_w, pic_h, x + dx0, y + dy0), b2 = pxc<PX>(rec, stride, pic_w, pic_h, x + dx1, y + dy1);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[kk][b] += clip_pair(clipv[b], cur, a, b2);
  };
  auto rowp = [&](int i) { return max(i, top); };
  auto rown = [&](int i) { return -max(i, -bot); };
  if (tr == 0 || tr == 2) {
#pragma unroll
    for (int i = -half; i < 0; ++i) {
      const int n = 2 * (half + i) + 1;
#pragma unroll
      for (int s = 0; s < n; ++s, ++k) {
        const int j = tr == 0 ? -half - i + s : half + i - s;
        add(k, j, rowp(i), -j, rown(i));
      }
    }
#pragma unroll
    for (int j = -half; j < 0; ++j, ++k) add(k, j, 0, -j, 0);
  } else {
#pragma unroll
    for (int j = -half; j < 0; ++j) {
      const int n = 2 * (half + j) + 1;
#pragma unroll
      for (int s = 0; s < n; ++s, ++k) {
        const int i = tr == 1 ? -half - j + s : half + j - s;
        add(k, j, rowp(i), -j, rown(i));
      }
    }
#pragma unroll
    for (int i = -half; i < 0; ++i, ++k) add(k, 0, rowp(i), 0, rown(i));
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[NC - 1][b] += cur;
#pragma unroll
  for (int kk = 0; kk < NC; ++kk)
#pragma unroll
    for (int b = 0; b < 4; ++b) e[kk * 4 + b] = (int16_t)acc[kk][b];
}

#define ALF_SPLIT 4
template <typename PX, bool CHROMA>
__global__ void __launch_bounds__(128)
alf_stats_kernel(const PX *__restrict__ org, int ostride, const PX *__restrict__ rec, int rstride, int pic_w, int pic_h,
                 const uvghip_rect_t *__restrict__ rects, const uint8_t *__restrict__ cls, int cls_stride,
                 long long *__restrict__ ee, int32_t *__restrict__ yv, long long *__restrict__ pix)
{
  constexpr int NC = CHROMA ? 7 : 13, NCLS = CHROMA ? 1 : 25;
  constexpr int NPAIR = NC * (NC + 1) / 2;
  constexpr int vbh = CHROMA ? 32 : 64, vb_pos = CHROMA ? 30 : 60;
  constexpr int DEPTH = px_traits<PX>::depth;
  __shared__ __attribute__((aligned(16))) int16_t sE[256 * 52];
  __shared__ int16_t sY[256];
  __shared__ uint8_t sCls[64];
  // ALF_SPLIT workgroups share one rectangle (each takes a contiguous run of the class-sorted blocks); the outputs are zeroed by the
  // host wrapper and every flush is an atomic add
  const int rect_i = blockIdx.x / ALF_SPLIT, part = blockIdx.x % ALF_SPLIT;
  const uvghip_rect_t R = rects[rect_i];
  long long *E = ee + (size_t)rect_i * NCLS * 13 * 13 * 16;
  int32_t *Y = yv + (size_t)rect_i * NCLS * 13 * 4;
  long long *PA = pix + (size_t)rect_i * NCLS;
  int clipv[4];
  clipv[0] = 1 << DEPTH;
#pragma unroll
  for (int i = 1; i < 4; ++i) clipv[i] = 1 << (7 - 2 * i + DEPTH - 8);     // alf.c:5248-5260

  // role of this thread in phase B
  const int t = threadIdx.x;
  int pk = 0, pl = 0;
  if (t < NPAIR) { int rem = t; while (rem >= NC - pk) { rem -= NC - pk; ++pk; } pl = pk + rem; }
  const bool is_pair = t < NPAIR, is_y = t >= NPAIR && t < NPAIR + NC, is_pix = t == NPAIR + NC;
  const int yk = t - NPAIR;
  long long acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  int cur_cls = -1;
  __syncthreads();

  auto flush = [&]() {
    if (cur_cls < 0) return;
    if (is_pair) {
      long long *e0 = E + ((size_t)cur_cls * 13 * 13 + pk * 13 + pl) * 16;
      long long *e1 = E + ((size_t)cur_cls * 13 * 13 + pl * 13 + pk) * 16;
#pragma unroll
      for (int b0 = 0; b0 < 4; ++b0)
#pragma unroll
        for (int b1 = 0; b1 < 4; ++b1) {
          if (acc[b0 * 4 + b1] == 0) continue;
          atomicAdd(reinterpret_cast<unsigned long long *>(e0 + b0 * 4 + b1), (unsigned long long)acc[b0 * 4 + b1]);
          if (pk != pl) atomicAdd(reinterpret_cast<unsigned long long *>(e1 + b1 * 4 + b0), (unsigned long long)acc[b0 * 4 + b1]);     // mirrored lower triangle (alf-generic.c:982-996)
        }
    } else if (is_y) {
#pragma unroll
      for (int b = 0; b < 4; ++b) if (acc[b]) atomicAdd(&Y[(cur_cls * 13 + yk) * 4 + b], (int32_t)acc[b]);
    } else if (is_pix) {
      if (acc[0]) atomicAdd(reinterpret_cast<unsigned long long *>(&PA[cur_cls]), (unsigned long long)acc[0]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
  };

  // The rectangle's 4x4 blocks are walked in class order (counting sort in LDS): the per-thread accumulators then change
  // class at most 25 times per rectangle instead of at nearly every block, and every change costs a read-modify-write
  // of the thread's 2 x 16 covariance entries in global memory.  Sums are integers, so the order does not matter.
  __shared__ uint8_t sBlkCls[256], sOrder[256];
  __shared__ int sCnt[32], sPos[32];
  const int bw = (R.w + 3) / 4, bhh = (R.h + 3) / 4, nblk = bw * bhh;      // <= 256 (rectangles are at most 64x64)
  if (t < 32) sCnt[t] = 0;
  __syncthreads();
  for (int i = t; i < nblk; i += blockDim.x) {
    const int by = i / bw, bx = i - by * bw;
    const int c = CHROMA ? 0 : (cls[((R.y >> 2) + by) * cls_stride + (R.x >> 2) + bx] & 31);
    sBlkCls[i] = (uint8_t)c;
    atomicAdd(&sCnt[c], 1);
  }
  __syncthreads();
  if (t == 0) { int run = 0; for (int c = 0; c < 32; ++c) { sPos[c] = run; run += sCnt[c]; } }
  __syncthreads();
  // stable placement (rank among the earlier blocks of the same class): every workgroup of the rectangle must derive
  // the same order, an atomic cursor would not
  for (int i = t; i < nblk; i += blockDim.x) {
    const int c = sBlkCls[i];
    int rank = 0;
    for (int j = 0; j < i; ++j) rank += sBlkCls[j] == c;
    sOrder[sPos[c] + rank] = (uint8_t)i;
  }
  __syncthreads();

  // contiguous share of the class-sorted list: a workgroup meets only a few classes, so it flushes only a few times
  const int nchunks = (nblk + 15) >> 4, per_part = (nchunks + ALF_SPLIT - 1) / ALF_SPLIT;
  for (int c0 = part * per_part * 16; c0 < min(nblk, (part + 1) * per_part * 16); c0 += 16) {
    const int nb = min(16, nblk - c0);
    // ---- phase A: tap sums of every sample of the chunk's blocks (slot = block j of the chunk, sample p of the block) ----
    for (int i = t; i < nb * 16; i += blockDim.x) {
      const int j = i >> 4, p = i & 15, blk = sOrder[c0 + j];
      const int by = blk / bw, bx = blk - by * bw;
      const int xx = bx * 4 + (p & 3), yy = by * 4 + (p >> 2);
      if (xx < R.w && yy < R.h) {
        const int x = R.x + xx, y = R.y + yy;
        int tr = 0;
        if constexpr (!CHROMA) tr = cls[(y >> 2) * cls_stride + (x >> 2)] >> 5;
        covariance_sample<PX, CHROMA>(sE + (size_t)i * 52, rec, rstride, pic_w, pic_h, x, y, tr, (y % vbh) - vb_pos, clipv);
        sY[i] = (int16_t)((int)org[(size_t)y * ostride + x] - (int)rec[(size_t)y * rstride + x]);
      } else {
        // sample outside the rectangle: contributes nothing
#pragma unroll
        for (int q = 0; q < 52; ++q) sE[(size_t)i * 52 + q] = 0;
        sY[i] = 0;
      }
    }
    if (t < nb) sCls[t] = sBlkCls[sOrder[c0 + t]];
    __syncthreads();
    // ---- phase B: outer products, block by block ----
    if (is_pair || is_y || is_pix) {
      for (int j = 0; j < nb; ++j) {
        const int c = sCls[j];
        if (c != cur_cls) { flush(); cur_cls = c; }
        int part[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) part[i] = 0;
        for (int p = j * 16; p < j * 16 + 16; ++p) {
          const int16_t *e = sE + (size_t)p * 52;
          if (is_pair) {
            const short4 ek = *reinterpret_cast<const short4 *>(e + pk * 4), el = *reinterpret_cast<const short4 *>(e + pl * 4);
            const int a[4] = {ek.x, ek.y, ek.z, ek.w}, bb[4] = {el.x, el.y, el.z, el.w};
#pragma unroll
            for (int b0 = 0; b0 < 4; ++b0)
#pragma unroll
              for (int b1 = 0; b1 < 4; ++b1) part[b0 * 4 + b1] += a[b0] * bb[b1];
          } else if (is_y) {
            const short4 ek = *reinterpret_cast<const short4 *>(e + yk * 4);
            const int yl = sY[p];
            part[0] += ek.x * yl; part[1] += ek.y * yl; part[2] += ek.z * yl; part[3] += ek.w * yl;
          } else {
            const int yl = sY[p];
            part[0] += yl * yl;
          }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += part[i];
      }
    }
    __syncthreads();
  }
  flush();
}

extern "C" int uvghip_alf_stats_batch(int bitdepth, const void *org, int org_stride, const void *rec, int rec_stride, int pic_w,
                                      int pic_h, int is_chroma, const uvghip_rect_t *rects, int n, const uint8_t *cls,
                                      int cls_stride, int64_t *ee, int32_t *y, int64_t *pix_acc, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
  const int ncls = is_chroma ? 1 : 25;
  UVGHIP_TRY(hipMemsetAsync(ee, 0, (size_t)n * ncls * 13 * 13 * 16 * 8, st));
  UVGHIP_TRY(hipMemsetAsync(y, 0, (size_t)n * ncls * 13 * 4 * 4, st));
  UVGHIP_TRY(hipMemsetAsync(pix_acc, 0, (size_t)n * ncls * 8, st));
#define K(PX, C) alf_stats_kernel<PX, C><<<n * ALF_SPLIT, 128, 0, st>>>((const PX *)org, org_stride, (const PX *)rec, rec_stride, pic_w, pic_h, rects, cls, cls_stride, (long long *)ee, y, (long long *)pix_acc)
  if (bitdepth == 8) { if (is_chroma) K(uint8_t, true); else K(uint8_t, false); }
  else { if (is_chroma) K(uint16_t, true); else K(uint16_t, false); }
#undef K
  UVGHIP_CHECK_LAUNCH();
}
