// Intra prediction on gfx950: reference construction + smoothing, planar, DC,
// angular (incl. wide angles), PDPC -- and the fused rough mode search
// (predict every candidate mode and cost it with min(SATD, 2*SAD) without the
// predictions ever leaving the CU).
//
// Bit-exact with the reference path
//   uvg_intra_build_reference   src/intra.c:756-1342   (frame-level semantics, MRL 0, no ISP)
//   intra_filter_reference      src/intra.c:190-225
//   intra_predict_regular       src/intra.c:660-753    (filtered/unfiltered choice, PDPC)
//   intra_pred_dc               src/intra.c:236-273
//   uvg_angular_pred_generic    src/strategies/generic/intra-generic.c:55-295
//   uvg_intra_pred_planar_generic :306-361, uvg_pdpc_planar_dc_generic :414-437
//   get_cost_dual               src/search_intra.c:133-158
//
// Data flow of the search kernel: per CU the 4N+1 reference samples and the
// N*N original samples are read from HBM once and staged in LDS (raw +
// smoothed reference rows, original block and its transpose).  Every
// (mode, 8x8 tile) pair is one task for an 8-lane group: each lane predicts
// one 8-pixel row in registers, the difference row goes straight into the
// row-per-lane Hadamard (picture.hip), and only two integers per task reach
// LDS.  Horizontal modes are evaluated in the reference's transposed "work"
// domain against the transposed original: SAD and the Hadamard magnitude
// multiset are transpose-invariant, so costs are identical.
#include <cstdio>
#include <cstdlib>
#include "uvghip_common.h"
#include <mutex>
#include "percall.h"
#include "ref_abi.h"
#include "satd_dev.h"
#include "satd_tile_dev.h"

#include "intra_pred_dev.h"

// Cooperative construction of the four reference rows of one block (threads tid0 + k*nthreads).
template <typename PX>
__device__ inline void build_ref_rows(const PX *__restrict__ rec, int stride, int x, int y, int w, int h,
                                      int avail_top, int avail_left, uint16_t *top, uint16_t *left, int refn,
                                      int tid, int nthreads)
{
  const int dc = 1 << (px_traits<PX>::depth - 1);
  if (avail_left < 1) avail_left = 1;
  if (avail_top < 1) avail_top = 1;
  for (int i = tid; i < refn - 1; i += nthreads) {
    int lv, tv;
    if (x > 0) lv = rec[(size_t)(y + min(i, avail_left - 1)) * stride + x - 1];
    else lv = y > 0 ? rec[(size_t)(y - 1) * stride + x] : dc;
    if (y > 0) tv = rec[(size_t)(y - 1) * stride + x + min(i, avail_top - 1)];
    else tv = x > 0 ? rec[(size_t)y * stride + x - 1] : dc;
    left[1 + i] = (uint16_t)lv;
    top[1 + i] = (uint16_t)tv;
  }
  if (tid == 0) {
    int c;
    if (x > 0 && y > 0) c = rec[(size_t)(y - 1) * stride + x - 1];
    else if (x > 0) c = rec[(size_t)y * stride + x - 1];          // == left[1]
    else c = y > 0 ? rec[(size_t)(y - 1) * stride + x] : dc;       // == left[1]
    top[0] = left[0] = (uint16_t)c;
  }
}
// Same rows, but every thread first issues all of its (at most KR per row) global loads and only then
// stores to LDS, so the round trips overlap instead of running one after the other.
// pic_rows > 0: `rec` is a stack of pictures of pic_rows rows each (block rows are stack coordinates) -- "is there a row above"
// is asked of the block's row inside its own picture.
template <typename PX, int KR>
__device__ __forceinline__ void build_ref_rows_batched(const PX *__restrict__ rec, int stride, int x, int y_abs,
                                                       int avail_top, int avail_left, uint16_t *top, uint16_t *left,
                                                       int refn, int tid, int nthreads, int pic_rows = 0)
{
  rec += (size_t)(pic_rows > 0 ? y_abs / pic_rows : 0) * pic_rows * stride;       // origin of the block's picture
  const int y = pic_rows > 0 ? y_abs % pic_rows : y_abs;
  const int dc = 1 << (px_traits<PX>::depth - 1);
  if (avail_left < 1) avail_left = 1;
  if (avail_top < 1) avail_top = 1;
  int lv[KR], tv[KR], c = dc;
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int i = tid + k * nthreads;
    lv[k] = tv[k] = dc;
    if (i < refn - 1) {
      if (x > 0) lv[k] = rec[(size_t)(y + min(i, avail_left - 1)) * stride + x - 1];
      else if (y > 0) lv[k] = rec[(size_t)(y - 1) * stride + x];
      if (y > 0) tv[k] = rec[(size_t)(y - 1) * stride + x + min(i, avail_top - 1)];
      else if (x > 0) tv[k] = rec[(size_t)y * stride + x - 1];
    }
  }
  if (tid == 0) {
    if (x > 0 && y > 0) c = rec[(size_t)(y - 1) * stride + x - 1];
    else if (x > 0) c = rec[(size_t)y * stride + x - 1];
    else if (y > 0) c = rec[(size_t)(y - 1) * stride + x];
  }
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int i = tid + k * nthreads;
    if (i < refn - 1) { left[1 + i] = (uint16_t)lv[k]; top[1 + i] = (uint16_t)tv[k]; }
  }
  if (tid == 0) top[0] = left[0] = (uint16_t)c;
}
// intra.c:190-225 (needs the raw rows complete: call after a barrier)
__device__ inline void filter_ref_rows(const uint16_t *top, const uint16_t *left, uint16_t *ftop, uint16_t *fleft,
                                       int w, int h, int refn, int tid, int nthreads)
{
  for (int i = tid; i < refn; i += nthreads) {
    int fl, ft;
    if (i == 0) fl = ft = (left[1] + 2 * left[0] + top[1] + 2) >> 2;
    else {
      fl = i < 2 * h ? (left[i - 1] + 2 * left[i] + left[i + 1] + 2) >> 2 : left[i];
      ft = i < 2 * w ? (top[i - 1] + 2 * top[i] + top[i + 1] + 2) >> 2 : top[i];
    }
    fleft[i] = (uint16_t)fl;
    ftop[i] = (uint16_t)ft;
  }
}
// ------------------------------------------------------------------ prediction kernel ----
// One workgroup per block; every (mode, work-domain row, 4-sample segment) is one thread task.
template <typename PX>
__global__ void __launch_bounds__(256)
intra_pred_kernel(const PX *__restrict__ rec, int stride, int is_chroma, int w, int h,
                  const uvghip_intra_blk_t *__restrict__ blks, const int8_t *__restrict__ modes, int n_modes,
                  PX *__restrict__ out, int refn)
{
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t *top = smem, *left = smem + refn, *ftop = smem + 2 * refn, *fleft = smem + 3 * refn;
  mode_info *sM = reinterpret_cast<mode_info *>(smem + 4 * refn);
  __shared__ int sDC;
  const uvghip_intra_blk_t b = blks[blockIdx.x];
  build_ref_rows<PX>(rec, stride, b.x, b.y, w, h, b.avail_top, b.avail_left, top, left, refn, threadIdx.x, blockDim.x);
  for (int m = threadIdx.x; m < n_modes; m += blockDim.x) sM[m] = make_mode_info(modes[m], w, h, is_chroma);
  __syncthreads();
  filter_ref_rows(top, left, ftop, fleft, w, h, refn, threadIdx.x, blockDim.x);
  if (threadIdx.x == 0) sDC = dc_value(top, left, w, h);
  __syncthreads();
  const ref_rows R{top, left, ftop, fleft};
  const int maxv = px_traits<PX>::maxv;
  const int wh = w * h, segs = wh / 4;
  PX *o = out + (size_t)blockIdx.x * n_modes * wh;
  for (int t = threadIdx.x; t < n_modes * segs; t += blockDim.x) {
    const int m = t / segs, s = t - m * segs;
    const mode_info M = sM[m];
    const bool transposed = M.mode >= 2 && !M.vertical;
    const int wd = transposed ? h : w, hd = transposed ? w : h;
    const int yd = s / (wd / 4), xd0 = (s - yd * (wd / 4)) * 4;
    int v[4];
    predict_row<4>(M, R, sDC, is_chroma, wd, hd, yd, xd0, maxv, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = transposed ? yd : xd0 + i, py = transposed ? xd0 + i : yd;
      o[(size_t)m * wh + py * w + px] = (PX)v[i];
    }
  }
}

static int ref_len(int w, int h)
{
  // enough for every in-range read of angular_pred incl. wide angles (intra.c:855-858: 2*dim + (dim << s) + 2)
  const int lw = 31 - __builtin_clz(w), lh = 31 - __builtin_clz(h);
  const int st = lw > lh ? lw - lh : 0, sl = lh > lw ? lh - lw : 0;
  int a = 2 * w + (w << st) + 4, b = 2 * h + (h << sl) + 4;
  int r = a > b ? a : b;
  r = (r + 7) & ~7;
  return r > 352 ? 352 : r;   // uvg_intra_ref rows hold INTRA_REF_LENGTH = 358 samples (intra.h:46)
}

extern "C" int uvghip_intra_pred_batch(int bitdepth, const void *rec, int rec_stride, int is_chroma, int width, int height,
                                       const uvghip_intra_blk_t *blks, int n, const int8_t *modes, int n_modes,
                                       void *preds_out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  auto ok = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32; };
  if (!ok(width) || !ok(height) || n_modes < 1 || n_modes > 128) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  const int refn = ref_len(width, height);
  const size_t lds = (size_t)4 * refn * 2 + (size_t)n_modes * sizeof(mode_info);
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8)
    intra_pred_kernel<uint8_t><<<n, 256, lds, st>>>((const uint8_t *)rec, rec_stride, is_chroma, width, height, blks, modes, n_modes, (uint8_t *)preds_out, refn);
  else
    intra_pred_kernel<uint16_t><<<n, 256, lds, st>>>((const uint16_t *)rec, rec_stride, is_chroma, width, height, blks, modes, n_modes, (uint16_t *)preds_out, refn);
  UVGHIP_CHECK_LAUNCH();
}

// ------------------------------------------------------ per-block-mode plane kernel ----
// Each block carries its own (already decided) mode; the prediction is written into a plane at
// the block's position -- what uvg_intra_recon_cu's predict step leaves in lcu->rec (intra.c:1537).
// One thread per 4-sample segment: n*n/4 threads per block, 256/(n*n/4) blocks per workgroup
// (64 4x4 blocks ... one 32x32 block); reference rows of 2n+4 samples per block in LDS.
template <typename PX, int N, bool CHROMA = false>      // N: the block size, a compile-time constant (all the index arithmetic folds)
__global__ void __launch_bounds__(256)
intra_pred_plane_kernel(const PX *__restrict__ rec, int stride, const uvghip_intra_blk_t *__restrict__ blks,
                        int n_blks, const int8_t *__restrict__ modes, PX *__restrict__ out, int out_stride, int pic_rows)
{
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  constexpr int n = N;
  constexpr int lgn = N == 4 ? 2 : (N == 8 ? 3 : (N == 16 ? 4 : 5));
  const int lg_tpb = 2 * lgn - 2, tpb = 1 << lg_tpb, bpg = 256 >> lg_tpb;
  const int RS = 2 * n + 4;
  uint16_t *sRef = smem;                                                     // [bpg][4][RS]
  int *sDC = reinterpret_cast<int *>(smem + (size_t)bpg * 4 * RS);            // [bpg], then x, y
  int *sX = sDC + bpg, *sY = sX + bpg;
  mode_info *sM = reinterpret_cast<mode_info *>(sY + bpg);                    // [bpg]
  const int blk0 = blockIdx.x * bpg;
  const int here = min(bpg, n_blks - blk0);
  if (here <= 0) return;
  const int myb = threadIdx.x >> lg_tpb, mytid = threadIdx.x & (tpb - 1);
  uint16_t *base = sRef + (size_t)myb * 4 * RS;
  if (myb < here) {
    const uvghip_intra_blk_t b = blks[blk0 + myb];
    build_ref_rows_batched<PX, 3>(rec, stride, b.x, b.y, b.avail_top, b.avail_left, base, base + RS, RS, mytid, tpb, pic_rows);
    if (mytid == 0) { sM[myb] = make_mode_info(modes[blk0 + myb], n, n, CHROMA ? 1 : 0); sX[myb] = b.x; sY[myb] = b.y; }
  }
  __syncthreads();
  if (myb < here) {
    filter_ref_rows(base, base + RS, base + 2 * RS, base + 3 * RS, n, n, RS, mytid, tpb);
    if (mytid == 0) sDC[myb] = dc_value(base, base + RS, n, n);
  }
  __syncthreads();
  if (myb >= here) return;
  const ref_rows R{base, base + RS, base + 2 * RS, base + 3 * RS};
  const mode_info M = sM[myb];
  const bool transposed = M.mode >= 2 && !M.vertical;
  const int maxv = px_traits<PX>::maxv;
  const int yd = mytid >> (lgn - 2), xd0 = (mytid & ((n >> 2) - 1)) * 4;
  int v[4];
  predict_row<4>(M, R, sDC[myb], CHROMA ? 1 : 0, n, n, yd, xd0, maxv, v);
  PX *o = out + (size_t)sY[myb] * out_stride + sX[myb];
  if (!transposed) {
    PX *q = o + (size_t)yd * out_stride + xd0;
    if constexpr (sizeof(PX) == 1) *reinterpret_cast<u32_unaligned *>(q) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    else { q[0] = (PX)v[0]; q[1] = (PX)v[1]; q[2] = (PX)v[2]; q[3] = (PX)v[3]; }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[(size_t)(xd0 + i) * out_stride + yd] = (PX)v[i];
  }
}

static int pred_plane_launch(int bitdepth, const void *rec, int rec_stride, int size, const uvghip_intra_blk_t *blks, int n,
                             const int8_t *modes, void *pred_plane, int pred_stride, bool chroma, hipStream_t st, const char *who,
                             int pic_rows = 0)
{
  if (!(size == 4 || size == 8 || size == 16 || size == 32) || (bitdepth != 8 && bitdepth != 10) || pic_rows < 0) return uvghip_set_error(hipErrorInvalidValue, who);
  if (n <= 0) return 0;
  const int bpg = 256 / (size * size / 4);
  const int grid = (n + bpg - 1) / bpg;
  const size_t lds = (size_t)bpg * 4 * (2 * size + 4) * 2 + (size_t)bpg * (sizeof(mode_info) + 12) + 16;
#define PP(PX, N, C) intra_pred_plane_kernel<PX, N, C><<<grid, 256, lds, st>>>((const PX *)rec, rec_stride, blks, n, modes, (PX *)pred_plane, pred_stride, pic_rows)
#define PPS(PX, C) do { if (size == 4) PP(PX, 4, C); else if (size == 8) PP(PX, 8, C); else if (size == 16) PP(PX, 16, C); else PP(PX, 32, C); } while (0)
  if (chroma) { if (bitdepth == 8) PPS(uint8_t, true); else PPS(uint16_t, true); }
  else { if (bitdepth == 8) PPS(uint8_t, false); else PPS(uint16_t, false); }
#undef PPS
#undef PP
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_intra_pred_plane_batch(int bitdepth, const void *rec, int rec_stride, int size,
                                             const uvghip_intra_blk_t *blks, int n, const int8_t *modes,
                                             void *pred_plane, int pred_stride, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return pred_plane_launch(bitdepth, rec, rec_stride, size, blks, n, modes, pred_plane, pred_stride, false, uvghip_stream(stream), __func__);
}

extern "C" int uvghip_intra_pred_plane_chroma_batch(int bitdepth, const void *rec, int rec_stride, int size,
                                                    const uvghip_intra_blk_t *blks, int n, const int8_t *modes,
                                                    void *pred_plane, int pred_stride, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return pred_plane_launch(bitdepth, rec, rec_stride, size, blks, n, modes, pred_plane, pred_stride, true, uvghip_stream(stream), __func__);
}

extern "C" int uvghip_intra_pred_plane_stacked_batch(int bitdepth, const void *rec, int rec_stride, int size,
                                                     const uvghip_intra_blk_t *blks, int n, const int8_t *modes,
                                                     void *pred_plane, int pred_stride, int pic_rows, int is_chroma, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return pred_plane_launch(bitdepth, rec, rec_stride, size, blks, n, modes, pred_plane, pred_stride, is_chroma != 0, uvghip_stream(stream),
                           __func__, pic_rows);
}

// costs[n][n_modes] -> best[n] = index (into the mode list) of the first minimum, the tie-break of
// the reference's strict "<" scans (search_intra.c:1089-1101 keeps the earlier candidate on ties)
// Eight lanes per row: lane j scans candidates j, j+8, ... (the group reads 32 contiguous bytes per step),
// then the (cost, index) pairs are min-reduced as 64-bit keys cost:index, so ties keep the lower index.
__global__ void __launch_bounds__(256)
argmin_rows_kernel(const uint32_t *__restrict__ costs, int n, int n_modes, const int8_t *__restrict__ modes,
                   int8_t *__restrict__ best_mode, uint32_t *__restrict__ best_cost)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 3, j = t & 7;
  const bool on = i < n;
  const uint32_t *c = costs + (size_t)(on ? i : 0) * n_modes;
  uint64_t key = ~0ull;
  for (int m = j; m < n_modes; m += 8) {
    const uint64_t k = ((uint64_t)c[m] << 32) | (uint32_t)m;
    key = k < key ? k : key;
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)key, off, 64), hi = __shfl_xor((uint32_t)(key >> 32), off, 64);
    const uint64_t k = ((uint64_t)hi << 32) | lo;
    key = k < key ? k : key;
  }
  if (on && j == 0) {
    best_mode[i] = modes[(uint32_t)key];
    if (best_cost) best_cost[i] = (uint32_t)(key >> 32);
  }
}

extern "C" int uvghip_intra_select_best(const uint32_t *costs, int n, const int8_t *modes, int n_modes,
                                        int8_t *best_mode, uint32_t *best_cost, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (n <= 0) return 0;
  argmin_rows_kernel<<<(n * 8 + 255) / 256, 256, 0, uvghip_stream(stream)>>>(costs, n, n_modes, modes, best_mode, best_cost);
  UVGHIP_CHECK_LAUNCH();
}

// ----------------------------------------------------------------------- search kernel ----
// Tile-per-lane rough search.  A workgroup owns 64 tiles (T x T, T = 8, or 4 for 4x4 blocks) =
// 64 / tiles_per_block blocks; lane l of every wave is tile l.  Its waves (8; 4 for 4x4 blocks) walk the
// candidate list (wave w takes modes w, w + WAVES, ...), so the mode -- and with it every branch of the
// predictor -- is uniform across the wave.  A lane predicts its whole tile row by row in registers,
// subtracts the original (packed 16-bit pairs straight from LDS), accumulates SAD with v_sad_u16 and runs
// the Hadamard in its own registers (satd_tile_dev.h).  Tile costs of one block are added across 1/4/16
// neighbouring lanes with DPP; one lane keeps min(SATD, 2*SAD), writes it (if asked) and folds it into the
// block's best (cost << 7 | candidate index) key, an LDS atomicMin across the waves at the end.
//
// LDS per block: the reference rows as dword "pair rows" of RS = 2n+4 entries (raw top/left, smoothed
// top/left; 8-bit 16x16/32x32 blocks also A and B of the smoothing filter, see search_tile_angular_ab), the
// original block and its transpose in bands of T rows (horizontal modes are predicted in the reference's
// transposed work domain and compared with the transposed original -- SAD and the Hadamard magnitudes are
// transpose-invariant).  All strides are chosen against bank conflicts (make_search_layout).  Per
// workgroup: the packed candidate descriptors, the tap tables, the PDPC weight / projected-offset tables.
// Negative angles read the projected side reference left of main[0] (intra-generic.c:150-170); each wave
// builds that extended pair row for its current mode in a private LDS strip.
struct search_mode {
  int kind;        // 0 planar, 1 DC, 2 angular
  int row_main;    // reference row (0 top, 1 left, 2 smoothed top, 3 smoothed left) that is "main"
  int row_side;
  int sd, inv;     // intraPredAngle, invAngle
  int pdpc;        // 0 none, 1 planar/DC, 2 angular (projected side sample), 3 gradient (sd == 0)
  int scale;       // PDPC scale
  int coef;        // offset of the 32-entry tap table in sCoef (0 cubic, 32 smoothing)
  int transposed;  // work domain is the transposed block
  int noclamp;     // the 4-tap result cannot leave the sample range (smoothing taps are a convex combination; so is an integer phase)
};

// LDS form of a search_mode: two dwords, so a wave fetches its mode with one 8-byte read + two v_readfirstlane and
// unpacks with scalar bit-field extracts.
//   x: kind[1:0] row_main[3:2] row_side[5:4] pdpc[7:6] scale[10:8] (0..2, 0 without PDPC) coef>>5 [11] transposed[12] noclamp[13]
//   y: sd (low 16, signed) | inv << 16 (inv < 2^15)
__device__ inline uint2 pack_search_mode(const search_mode &S)
{
  uint2 p;
  p.x = (uint32_t)S.kind | ((uint32_t)S.row_main << 2) | ((uint32_t)S.row_side << 4) | ((uint32_t)S.pdpc << 6) |
        ((uint32_t)((S.pdpc ? S.scale : 0) & 7) << 8) | ((uint32_t)(S.coef >> 5) << 11) | ((uint32_t)S.transposed << 12) | ((uint32_t)S.noclamp << 13);
  p.y = ((uint32_t)S.sd & 0xffffu) | ((uint32_t)S.inv << 16);
  return p;
}
__device__ __forceinline__ search_mode unpack_search_mode(uint32_t x, uint32_t y)
{
  search_mode S;
  S.kind = x & 3; S.row_main = (x >> 2) & 3; S.row_side = (x >> 4) & 3; S.pdpc = (x >> 6) & 3;
  S.scale = (x >> 8) & 7;                         // only meaningful (and then 0..2) when pdpc != 0
  S.coef = ((x >> 11) & 1) << 5; S.transposed = (x >> 12) & 1; S.noclamp = (x >> 13) & 1;
  S.sd = (int)(int16_t)(y & 0xffffu); S.inv = (int)(y >> 16);
  return S;
}

__device__ inline search_mode make_search_mode(int mode, int n)
{
  const mode_info M = make_mode_info(mode, n, n, 0);
  const int lgn = ilog2_dev(n);
  search_mode S;
  S.kind = mode < 2 ? mode : 2;
  const int f = M.filtered ? 2 : 0;
  if (mode < 2 || M.vertical) { S.row_main = f; S.row_side = f + 1; }
  else { S.row_main = f + 1; S.row_side = f; }
  S.sd = M.sample_disp; S.inv = M.inv_disp;
  S.pdpc = !M.pdpc ? 0 : (mode < 2 ? 1 : (M.sample_disp != 0 ? 2 : 3));
  S.scale = (mode < 2 || M.sample_disp == 0) ? (2 * lgn - 2) >> 2 : M.scale;
  S.coef = M.use_cubic ? 0 : 32;
  S.transposed = mode >= 2 && !M.vertical;
  S.noclamp = mode >= 2 && (!M.use_cubic || !M.frac);
  return S;
}

// ---- tile predictors of the search kernel --------------------------------------------------
// Each predicts rows yd0..yd0+T-1, columns xd0..xd0+T-1 of the work domain, leaves
// original - prediction in d (packed pairs) and adds the tile's SAD to `sad`.  One function per
// predictor family, every one straight-line over the T rows (column-dependent PDPC weights are
// hoisted and set to zero where PDPC does not apply instead of branching), with the LDS loads
// of row r+1 issued before the arithmetic of row r.

// Reference rows live in LDS as "pair rows": dword i of a row holds (sample i, sample i+1).  A 4-tap
// window starting at any sample is then a run of aligned dwords that are directly the packed operands
// of v_dot2_i32_i16 -- no sub-dword-aligned wide loads (those cost tens of LDS cycles on gfx950) and
// no shuffling.  Single samples are the low halves.
__device__ __forceinline__ int pr_sample(const uint32_t *row, int i) { return reinterpret_cast<const uint16_t *>(row)[2 * i]; }

// one row of the lane's original tile: packed 16-bit pairs straight from LDS (the block and its transpose are
// staged as uint16 for both bit depths, so the row is one aligned b128 / b64 read issued with the row's taps)
template <int T>
__device__ __forceinline__ void load_orig_row(const uint16_t *otile, int n, int r, uint32_t (&o)[T / 2])
{
  if constexpr (T == 8) {
    const uint4 v = *reinterpret_cast<const uint4 *>(otile + r * n);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
    const uint2 v = *reinterpret_cast<const uint2 *>(otile + r * n);
    o[0] = v.x; o[1] = v.y;
  }
}
__device__ __forceinline__ uint32_t pack_lo16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }

template <int T>
__device__ __forceinline__ void finish_row(const uint32_t (&pp)[T / 2], const uint32_t (&o)[T / 2], uint32_t (&drow)[T / 2], uint32_t &sad)
{
#pragma unroll
  for (int c = 0; c < T / 2; ++c) {
    sad = __builtin_amdgcn_sad_u16(o[c], pp[c], sad);
    drow[c] = pk_sub(o[c], pp[c]);
  }
}
template <int T>
__device__ __forceinline__ void finish_row_i32(const int (&out)[T], const uint32_t (&o)[T / 2], uint32_t (&drow)[T / 2], uint32_t &sad)
{
  uint32_t pp[T / 2];
#pragma unroll
  for (int c = 0; c < T / 2; ++c) pp[c] = (uint32_t)out[2 * c] | ((uint32_t)out[2 * c + 1] << 16);
  finish_row<T>(pp, o, drow, sad);
}

// PDPC column weights 32 >> ((2x) >> scale) for x < lim, else 0 (a zero weight leaves the sample as is)
template <int T>
__device__ __forceinline__ void pdpc_col_weights(int xd0, int scale, int lim, int (&wl)[T])
{
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const int x = xd0 + i;
    wl[i] = x < lim ? 32 >> min(31, (2 * x) >> scale) : 0;
  }
}

// One row of 4-tap inputs: pairs P[k] = (p[k], p[k+1]), k = 0..T+1, and the two coefficient pairs
// (f0,f1), (f2,f3) of the row's phase.
template <int T> struct ang_row { uint32_t P[T + 2]; uint32_t f01, f23; uint32_t o[T / 2]; };

template <int T>
__device__ __forceinline__ void ang_load(const search_mode &S, const uint32_t *rowp, const uint2 *sCoef, int xd0, int yd,
                                         const uint16_t *otile, int n, int r, ang_row<T> &R)
{
  load_orig_row<T>(otile, n, r, R.o);
  const int delta = __mul24(S.sd, yd + 1), di = delta >> 5, df = delta & 31;
  const uint2 cf = sCoef[S.coef + df];
  R.f01 = cf.x; R.f23 = cf.y;
  const uint32_t *q = rowp + di + xd0;
#pragma unroll
  for (int k = 0; k < T + 2; ++k) R.P[k] = q[k];
}
// The tap table holds 4*f, so with the rounding term 4*32 the accumulator is 4*(sum + 32) and the
// filtered sample (sum + 32) >> 6 (intra-generic.c:216-222, unclamped here) sits byte-aligned in bits 8..23:
// two v_dot2_i32_i16 per sample, and one v_perm packs a pair of samples.
__device__ __forceinline__ int dot2_round(uint32_t a, uint32_t b)
{
  int v;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(v) : "v"(a), "v"(b), "s"(128));   // VOP3P form: scalar accumulator, no v_mov
  return v;
}
template <int T>
__device__ __forceinline__ void ang_filter_x4(const ang_row<T> &R, int (&acc)[T])
{
  const pk_s16 f23 = __builtin_bit_cast(pk_s16, R.f23);
#pragma unroll
  for (int i = 0; i < T; ++i)
    acc[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pk_s16, R.P[i + 2]), f23, dot2_round(R.P[i], R.f01), false);
}
// (lo >> 8, hi >> 8) as packed 16-bit halves
__device__ __forceinline__ uint32_t pack_shr8(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x06050201u); }

// PDPC: 0 none, 2 projected side sample (intra-generic.c:262-277), 3 gradient of the pure
// horizontal/vertical modes (:279-293)
template <int T, int PDPC, bool CLAMP, bool PK8>
__device__ __forceinline__ void search_tile_angular(const search_mode &S, const uint32_t *mainr, const uint32_t *side,
                                                    const uint32_t *rowp, const uint2 *sCoef, const uint16_t *wrow,
                                                    const uint16_t *sorow, int n, int xd0, int yd0,
                                                    const uint16_t *otile, int maxv, uint32_t (&d)[T][T / 2], uint32_t &sad)
{
  uint32_t wpk[T / 2];       // PDPC column weights (<= 32) of the lane's columns as 16-bit pairs
  const uint16_t *sp[T];    // PDPC 2: address of the projected side sample of column i in row yd0 (dword stride per row)
  int tl = 0;
  if constexpr (PDPC == 2) {
    // weights and projected-sample offsets come from the tables built at staging: one wide read each
    uint32_t so[T / 2];
    if constexpr (T == 8) {
      const uint4 w = *reinterpret_cast<const uint4 *>(wrow + xd0), o = *reinterpret_cast<const uint4 *>(sorow + xd0);
      wpk[0] = w.x; wpk[1] = w.y; wpk[2] = w.z; wpk[3] = w.w;
      so[0] = o.x; so[1] = o.y; so[2] = o.z; so[3] = o.w;
    } else {
      const uint2 w = *reinterpret_cast<const uint2 *>(wrow + xd0), o = *reinterpret_cast<const uint2 *>(sorow + xd0);
      wpk[0] = w.x; wpk[1] = w.y;
      so[0] = o.x; so[1] = o.y;
    }
    const char *sbase = reinterpret_cast<const char *>(side + yd0);
#pragma unroll
    for (int i = 0; i < T; ++i) sp[i] = reinterpret_cast<const uint16_t *>(sbase + ((so[i >> 1] >> (16 * (i & 1))) & 0xffffu));
  } else if constexpr (PDPC == 3) {
    const int lim = min(3 << S.scale, n);
    int wl[T];
    pdpc_col_weights<T>(xd0, S.scale, lim, wl);
#pragma unroll
    for (int c = 0; c < T / 2; ++c) wpk[c] = (uint32_t)wl[2 * c] | ((uint32_t)wl[2 * c + 1] << 16);
    tl = pr_sample(mainr, 0);
    sp[0] = reinterpret_cast<const uint16_t *>(side + yd0 + 1);
  }
  // the gradient blend of the pure horizontal / vertical modes only ever multiplies by a weight: bytes suffice
  uint32_t wlp[T / 4];
  if constexpr (PDPC == 3) {
#pragma unroll
    for (int q = 0; q < T / 4; ++q) wlp[q] = __builtin_amdgcn_perm(wpk[2 * q + 1], wpk[2 * q], 0x06040200u);
  }
  auto wl_of = [&](int i) -> int { return (int)((wlp[i >> 2] >> (8 * (i & 3))) & 0xffu); };
  ang_row<T> A, B;
  int lA[T];
  auto side_load = [&](int r, int (&l)[T]) {
    if constexpr (PDPC == 3) l[0] = sp[0][2 * r];      // PDPC 2 reads its side samples where it blends them
  };
  const pk_s16 vmax = {(short)maxv, (short)maxv};
  // PDPC 2 is the register-hungriest variant (eight side-sample pointers and values on top of the 32 difference
  // registers): it loads each row when it needs it instead of one row ahead; with four waves per SIMD the wait is covered
  constexpr bool PREFETCH = PDPC != 2;
  if constexpr (PREFETCH) ang_load<T>(S, rowp, sCoef, xd0, yd0, otile, n, 0, A);
#pragma unroll
  for (int r = 0; r < T; ++r) {
    if constexpr (PREFETCH) { if (r + 1 < T) ang_load<T>(S, rowp, sCoef, xd0, yd0 + r + 1, otile, n, r + 1, B); }
    else ang_load<T>(S, rowp, sCoef, xd0, yd0 + r, otile, n, r, A);
    side_load(r, lA);
    int out[T];
    ang_filter_x4<T>(A, out);
    if constexpr (PDPC == 0) {
      uint32_t pp[T / 2];
#pragma unroll
      for (int c = 0; c < T / 2; ++c) {
        pk_s16 v = __builtin_bit_cast(pk_s16, pack_shr8(out[2 * c], out[2 * c + 1]));
        if constexpr (CLAMP) v = __builtin_elementwise_min(__builtin_elementwise_max(v, (pk_s16){0, 0}), vmax);
        pp[c] = __builtin_bit_cast(uint32_t, v);
      }
      finish_row<T>(pp, A.o, d[r], sad);
    } else {
      if constexpr (PDPC == 2) {
        // the whole blend  c + ((wl * (l - c) + 32) >> 6)  runs on packed pairs: |wl * (l - c)| <= 32 * 1023 fits int16.
        // 8-bit: the rounding term fits too (v_pk_mad_i16, shift).  10-bit: 32736 + 32 would overflow, so the shift is
        // split -- (P + 32) >> 6 == ((P >> 5) + 1) >> 1 for every integer P.
        uint32_t pp[T / 2];
#pragma unroll
        for (int c = 0; c < T / 2; ++c) {
          pk_s16 v = __builtin_bit_cast(pk_s16, pack_shr8(out[2 * c], out[2 * c + 1]));
          if constexpr (CLAMP) v = __builtin_elementwise_min(__builtin_elementwise_max(v, (pk_s16){0, 0}), vmax);
          const pk_s16 l = {(short)sp[2 * c][2 * r], (short)sp[2 * c + 1][2 * r]};   // projected side samples of the pair
          const pk_s16 w = __builtin_bit_cast(pk_s16, wpk[c]);
          pk_s16 t;
          if constexpr (PK8) t = ((l - v) * w + (pk_s16){32, 32}) >> (pk_s16){6, 6};
          else t = ((((l - v) * w) >> (pk_s16){5, 5}) + (pk_s16){1, 1}) >> (pk_s16){1, 1};
          pp[c] = __builtin_bit_cast(uint32_t, v + t);
        }
        finish_row<T>(pp, A.o, d[r], sad);
        continue;
      } else {
        const int g = lA[0] - tl;
#pragma unroll
        for (int i = 0; i < T; ++i) out[i] = clampi((CLAMP ? clampi(out[i] >> 8, 0, maxv) : out[i] >> 8) + ((__mul24(wl_of(i), g) + 32) >> 6), 0, maxv);
      }
      finish_row_i32<T>(out, A.o, d[r], sad);
    }
    if constexpr (PREFETCH) A = B;
  }
}

// Smoothing-filter modes on 8-bit samples, non-negative angles (the majority of the candidates of 16x16 and 32x32 blocks).
// The 4-tap smoothing kernel is (16 - h, 32 - h, 16 + h, h), h = fraction >> 1 (intra-generic.c:206-214), so
//     sum f_k p_k + 32 = A + h * B,   A = 16 (p0 + 2 p1 + p2) + 32,   B = p2 + p3 - p0 - p1,
// exactly.  A and B do not depend on the mode: they are staged once per block as two more pair rows per reference side
// (rowA, rowB: dword i = values i and i + 1), and a row of the tile is four aligned dword reads of each followed by one
// v_pk_mad_i16 and one v_pk_ashrrev_i16 per sample pair -- 8 VALU and 8 window dwords per row instead of 20 and 10
// (+ the tap fetch).  Ranges: A <= 16352, |h B| <= 15 * 510, so everything stays inside int16; the result is a convex
// combination of samples (no clamp).  PDPC as in search_tile_angular (0 none, 2 projected side sample).
template <int T, int PDPC>
__device__ __forceinline__ void search_tile_angular_ab(const search_mode &S, const uint32_t *side, const uint32_t *rowA,
                                                       const uint32_t *rowB, const uint16_t *wrow, const uint16_t *sorow,
                                                       int n, int xd0, int yd0, const uint16_t *otile,
                                                       uint32_t (&d)[T][T / 2], uint32_t &sad)
{
  static_assert(T == 8, "used for 16x16 / 32x32 blocks only");
  uint32_t wpk[T / 2];
  const uint16_t *sp[T];
  if constexpr (PDPC == 2) {
    const uint4 w = *reinterpret_cast<const uint4 *>(wrow + xd0), o = *reinterpret_cast<const uint4 *>(sorow + xd0);
    wpk[0] = w.x; wpk[1] = w.y; wpk[2] = w.z; wpk[3] = w.w;
    const uint32_t so[4] = {o.x, o.y, o.z, o.w};
    const char *sbase = reinterpret_cast<const char *>(side + yd0);
#pragma unroll
    for (int i = 0; i < T; ++i) sp[i] = reinterpret_cast<const uint16_t *>(sbase + ((so[i >> 1] >> (16 * (i & 1))) & 0xffffu));
  }
  struct ab_row { uint32_t a[T / 2], b[T / 2], hh, o[T / 2]; };
  auto load = [&](int r, ab_row &R) {
    load_orig_row<T>(otile, n, r, R.o);
    const int delta = __mul24(S.sd, yd0 + r + 1), k0 = (delta >> 5) + xd0, h = (delta & 31) >> 1;
    R.hh = (uint32_t)h | ((uint32_t)h << 16);
#pragma unroll
    for (int c = 0; c < T / 2; ++c) { R.a[c] = rowA[k0 + 2 * c]; R.b[c] = rowB[k0 + 2 * c]; }
  };
  ab_row A, B;
  constexpr bool PREFETCH = PDPC != 2;
  if constexpr (PREFETCH) load(0, A);
#pragma unroll
  for (int r = 0; r < T; ++r) {
    if constexpr (PREFETCH) { if (r + 1 < T) load(r + 1, B); }
    else load(r, A);
    uint32_t pp[T / 2];
#pragma unroll
    for (int c = 0; c < T / 2; ++c) {
      const pk_s16 acc = __builtin_bit_cast(pk_s16, A.b[c]) * __builtin_bit_cast(pk_s16, A.hh) + __builtin_bit_cast(pk_s16, A.a[c]);
      pk_s16 v = acc >> (pk_s16){6, 6};
      if constexpr (PDPC == 2) {
        const pk_s16 l = {(short)sp[2 * c][2 * r], (short)sp[2 * c + 1][2 * r]};
        const pk_s16 w = __builtin_bit_cast(pk_s16, wpk[c]);
        v = v + (((l - v) * w + (pk_s16){32, 32}) >> (pk_s16){6, 6});
      }
      pp[c] = __builtin_bit_cast(uint32_t, v);
    }
    finish_row<T>(pp, A.o, d[r], sad);
    if constexpr (PREFETCH) A = B;
  }
}

// planar (intra-generic.c:306-361) or DC, both with the planar/DC PDPC (:414-437).
// ((hor << lg) + (ver << lg) + (1 << 2lg)) >> (2lg + 1) == (hor + ver + n) >> (lg + 1); hor and ver are
// linear in x and y, so they advance by one addition per sample.
template <int T, bool PLANAR>
__device__ __forceinline__ void search_tile_nonangular(const search_mode &S, const uint32_t *top, const uint32_t *left, int dc,
                                                       int n, int lgn, int xd0, int yd0, const uint16_t *otile,
                                                       uint32_t (&d)[T][T / 2], uint32_t &sad)
{
  int t[T], ver[T];
  uint32_t wlp[T / 4];       // column weights as bytes (see search_tile_angular)
  const uint32_t *topx = top + xd0;      // one base register; the columns are immediate offsets
#pragma unroll
  for (int i = 0; i < T; ++i) t[i] = pr_sample(topx, i + 1);
  {
    int wl[T];
    pdpc_col_weights<T>(xd0, S.scale, S.pdpc ? n : 0, wl);
#pragma unroll
    for (int q = 0; q < T / 4; ++q)
      wlp[q] = (uint32_t)wl[4 * q] | ((uint32_t)wl[4 * q + 1] << 8) | ((uint32_t)wl[4 * q + 2] << 16) | ((uint32_t)wl[4 * q + 3] << 24);
  }
  auto wl_of = [&](int i) -> int { return (int)((wlp[i >> 2] >> (8 * (i & 3))) & 0xffu); };
  int tr = 0, bl = 0;
  if constexpr (PLANAR) {
    tr = pr_sample(top, n + 1);
    bl = pr_sample(left, n + 1);
#pragma unroll
    for (int i = 0; i < T; ++i) ver[i] = (t[i] << lgn) + __mul24(yd0, bl - t[i]);   // + (bl - t) per row below
  }
  int lA = pr_sample(left, yd0 + 1), lB = 0;
  uint32_t oA[T / 2], oB[T / 2];
  load_orig_row<T>(otile, n, 0, oA);
#pragma unroll
  for (int r = 0; r < T; ++r) {
    const int yd = yd0 + r;
    if (r + 1 < T) { lB = pr_sample(left, yd + 2); load_orig_row<T>(otile, n, r + 1, oB); }
    int out[T];
    if constexpr (PLANAR) {
      const int dh = tr - lA;
      int hor = (lA << lgn) + __mul24(xd0, dh) + n;
#pragma unroll
      for (int i = 0; i < T; ++i) {
        hor += dh; ver[i] += bl - t[i];
        out[i] = (hor + ver[i]) >> (lgn + 1);
      }
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i) out[i] = dc;
    }
    const int wt = S.pdpc ? 32 >> min(31, (yd << 1) >> S.scale) : 0;
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const int c = out[i];
      out[i] = c + ((__mul24(wl_of(i), lA - c) + __mul24(wt, t[i] - c) + 32) >> 6);
    }
    finish_row_i32<T>(out, oA, d[r], sad);
    lA = lB;
#pragma unroll
    for (int c = 0; c < T / 2; ++c) oA[c] = oB[c];
  }
}

struct search_layout {
  int RS;        // samples per reference row (u16 scratch) = dwords per pair row
  int BRS;       // dwords per block: four pair rows, odd so that the wave's lanes spread over the banks
  int OS;        // uint16 elements per block: original + transpose + pad
  int AB;        // 8-bit, n >= 16: four more pair rows per block (A and B of the smoothing filter for top and left)
  int BAND;      // uint16 elements per band of T rows of the original (T * n + pad)
  int OT;        // uint16 offset of the transposed copy inside the block's original area
  int PS;        // dwords per private extended row (odd)
  int off_orig, off_ref, off_priv, off_dc, off_coef, off_mode, off_wtab, off_sotab;   // bytes
  size_t total;
};
// measured on MI355X with tools/dev/sweep_lds_strides.sh (SQ_LDS_BANK_CONFLICT per launch)
#ifndef UVGHIP_BRS16_PAD
#define UVGHIP_BRS16_PAD 11
#endif
#ifndef UVGHIP_PS16
#define UVGHIP_PS16 37
#endif
#ifndef UVGHIP_BRS32_PAD
#define UVGHIP_BRS32_PAD 2
#endif
#ifndef UVGHIP_PS32
#define UVGHIP_PS32 82
#endif
__host__ __device__ inline search_layout make_search_layout(int n, int bpg, int n_modes, int waves, int pxsz)
{
  search_layout L;
  L.RS = 2 * n + 4;
  // Strides chosen against LDS bank conflicts (64 banks x 4 B; measured with SQ_LDS_BANK_CONFLICT: at 32x32 half of
  // the LDS-active cycles were conflicts).  The 64 lanes of a wave are (block, tile column, tile row): window reads
  // start at block * BRS + 8 * column + row-dependent phase, so BRS / PS are picked per size by a model of the reads
  // of all 65 angular modes (tools/dev/lds_bank_model.py); original rows are read as b128 at band * BAND + row * n +
  // 8 * column, where a band stride of 0 mod 64 dwords (32x32: 8 rows of 16 dwords) puts all four tile rows on the
  // same banks -- the pad makes it 4 mod 16 quads, so (column, row) tile the 64 banks exactly.
  L.AB = pxsz == 1 && n >= 16;
  L.BRS = (L.AB ? 8 : 4) * L.RS + (n == 16 ? UVGHIP_BRS16_PAD : (n == 32 ? UVGHIP_BRS32_PAD : 1));
  L.PS = n == 16 ? UVGHIP_PS16 : (n == 32 ? UVGHIP_PS32 : 2 * n + 1);
  // Original rows are read as one b128 (4x4: b64) per lane, which the LDS serves 16 (32) lanes per cycle; the 16
  // lanes of such a group must land on 16 different bank quads, q = block * OSq + band * BANDq + row * rowq + column:
  //   32x32 (group = the 16 tiles of one block):        BANDq = 36 = 4 (mod 16)  -> 4 * band + column; the block stride
  //          has to be 0 mod 64 dwords on top of that (tools/dev/lds_probe.hip sweep: 4.5 clk per b128 instead of 8.4)
  //   16x16 (group = 4 blocks x 4 tiles):               BANDq = 18 = 2, OSq = 76 = 12 (mod 16) -> 12 * block + 2 * band + column
  //   8x8   (group = 16 blocks):                        OSq = 17 = 1 (mod 16)
  //   4x4   (b64, group = 32 blocks, dword pairs):      OS = 22 dwords = 2 * 11 (mod 64)
  const int t = n >= 8 ? 8 : 4;
  L.BAND = t * n + (n == 32 ? 32 : (n == 16 ? 16 : 0));
  L.OT = (n / t) * L.BAND;
  L.OS = 2 * L.OT + (n == 32 ? 0 : (n == 16 ? 32 : (n == 4 ? 12 : 8)));
  size_t o = 0;
  L.off_orig = (int)o; o += (size_t)bpg * L.OS * 2; o = (o + 15) & ~(size_t)15;
  L.off_ref = (int)o;  o += (size_t)bpg * L.BRS * 4; o = (o + 15) & ~(size_t)15;
  // private strips; the same space holds the u16 staging image of the rows (4 * RS samples per block)
  size_t pv = (size_t)waves * bpg * L.PS * 4, scratch = (size_t)bpg * 4 * L.RS * 2;
  L.off_priv = (int)o; o += pv > scratch ? pv : scratch; o = (o + 15) & ~(size_t)15;
  L.off_dc = (int)o;   o += (size_t)bpg * 8;      // DC values, then the per-block best keys
  o = (o + 7) & ~(size_t)7;
  L.off_coef = (int)o; o += 64 * 8;
  o = (o + 7) & ~(size_t)7;
  L.off_mode = (int)o; o += (size_t)n_modes * 8;
  o = (o + 15) & ~(size_t)15;
  L.off_wtab = (int)o; o += (size_t)3 * n * 2;          // PDPC column weights, u16 [scale 0..2][x < n]
  o = (o + 15) & ~(size_t)15;
  L.off_sotab = (int)o; o += (size_t)n_modes * n * 2;   // PDPC projected-side-sample byte offsets, u16 [candidate][x < n]
  L.total = o;
  return L;
}

#ifndef UVGHIP_SEARCH_WAVES
#define UVGHIP_SEARCH_WAVES 8
#endif
#ifndef UVGHIP_SEARCH_BPL4
#define UVGHIP_SEARCH_BPL4 1      // 4x4 blocks per lane (2 was measured slower: 109 vs 84 us per 1080p launch)
#endif
// BPL = blocks per lane (only with one tile per block): the lane evaluates BPL blocks per mode, which amortises the
// per-mode overhead (descriptor fetch, branching, bookkeeping) that dominates for 4x4 blocks.
// NFIX: block size known at compile time (4x4 and 8x8: one tile per block, so the tile origin is (0, 0) and the per-row
// angle arithmetic -- integer offset, fraction, tap pair -- becomes scalar), 0 = run-time size (16x16 / 32x32).
template <typename PX, int T, int WAVES, int BPL, int NFIX>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 8 ? 4 : (WAVES == 6 ? 3 : 1))
intra_search_kernel(const PX *__restrict__ rec, int rec_stride, const PX *__restrict__ orig, int orig_stride,
                    int n_arg, const uvghip_intra_blk_t *__restrict__ blks, int n_blks,
                    const int8_t *__restrict__ modes, int n_modes, uint32_t *__restrict__ costs,
                    int8_t *__restrict__ best_mode, uint32_t *__restrict__ best_cost, int pic_rows)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = NFIX ? NFIX : n_arg;
  const int lgn = NFIX ? (NFIX == 4 ? 2 : (NFIX == 8 ? 3 : (NFIX == 16 ? 4 : 5))) : ilog2_dev(n);
  const int lg_tx = lgn - (T == 8 ? 3 : 2);       // log2(tiles per block row)
  const int lg_tiles = 2 * lg_tx, tiles = 1 << lg_tiles;
  const int bpg = (64 * BPL) >> lg_tiles;
  const search_layout L = make_search_layout(n, bpg, n_modes, WAVES, (int)sizeof(PX));
  uint16_t *sOrig = reinterpret_cast<uint16_t *>(smem_raw + L.off_orig);
  uint32_t *sRef = reinterpret_cast<uint32_t *>(smem_raw + L.off_ref);
  uint32_t *sPriv = reinterpret_cast<uint32_t *>(smem_raw + L.off_priv);
  uint16_t *sScratch = reinterpret_cast<uint16_t *>(smem_raw + L.off_priv);
  int *sDC = reinterpret_cast<int *>(smem_raw + L.off_dc);
  uint32_t *sBest = reinterpret_cast<uint32_t *>(sDC + bpg);     // per block: min over modes of cost << 7 | candidate index
  uint2 *sCoef = reinterpret_cast<uint2 *>(smem_raw + L.off_coef);
  uint2 *sMode = reinterpret_cast<uint2 *>(smem_raw + L.off_mode);
  uint16_t *sWtab = reinterpret_cast<uint16_t *>(smem_raw + L.off_wtab);
  uint16_t *sSoTab = reinterpret_cast<uint16_t *>(smem_raw + L.off_sotab);

  const int blk0 = blockIdx.x * bpg;
  const int here = min(bpg, n_blks - blk0);
  if (here <= 0) return;
  const int nn = n * n;

  // ---- stage: tpb = WAVES * tiles threads per block; all global loads of a thread are in flight together ----
  {
    constexpr int NT = WAVES * 64;
    // staging uses the largest power-of-two share of the workgroup (8 or 4 waves): tpb threads per block
    const int lg_tpb = lg_tiles + (WAVES == 8 ? 3 : 2) - (BPL == 2 ? 1 : 0), tpb = 1 << lg_tpb;
    const int myb = threadIdx.x >> lg_tpb, mytid = threadIdx.x & (tpb - 1);
    const bool on = myb < here;     // also false for the waves beyond the staging share (myb >= bpg)
    uint16_t *base = sScratch + (size_t)myb * 4 * L.RS;     // u16 image: top | left | ftop | fleft
    if (on) {
      const uvghip_intra_blk_t b = blks[blk0 + myb];
      build_ref_rows_batched<PX, 6>(rec, rec_stride, b.x, b.y, b.avail_top, b.avail_left, base, base + L.RS, L.RS, mytid, tpb, pic_rows);
      // original block in 4-sample segments: segment sg = (row, 4 columns)
      uint16_t *so = sOrig + (size_t)myb * L.OS, *sot = so + L.OT;
      auto band_off = [&](int row) { return (row / T) * L.BAND + (row % T) * n; };   // element offset of a row of the block
      const int nseg = nn >> 2, lg_spr = lgn - 2;     // segments, log2(segments per row)
      int v[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sg = mytid + k * tpb;
        if (sg < nseg) {
          const int yy = sg >> lg_spr, xx = (sg & ((1 << lg_spr) - 1)) * 4;
          load4(orig + (size_t)(b.y + yy) * orig_stride + b.x + xx, v[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sg = mytid + k * tpb;
        if (sg < nseg) {
          const int yy = sg >> lg_spr, xx = (sg & ((1 << lg_spr) - 1)) * 4;
          *reinterpret_cast<uint2 *>(so + band_off(yy) + xx) = make_uint2((uint32_t)v[k][0] | ((uint32_t)v[k][1] << 16), (uint32_t)v[k][2] | ((uint32_t)v[k][3] << 16));
#pragma unroll
          for (int i = 0; i < 4; ++i) sot[band_off(xx + i) + yy] = (uint16_t)v[k][i];
        }
      }
    }
    for (int m = threadIdx.x; m < n_modes; m += NT) sMode[m] = pack_search_mode(make_search_mode(modes[m], n));
    // PDPC tables (intra-generic.c:262-277): what every lane used to derive per mode from (column, scale, invAngle)
    //   wtab[s][x]  = 32 >> ((2x) >> s) for x < min(3 << s, n), else 0 (a zero weight leaves the sample as is)
    //   sotab[m][x] = byte offset, inside the side pair row, of the projected side sample of column x in row 0
    //                 (side[(inv_sum >> 9) + 1]); 0, a harmless in-range offset, where the column has no PDPC
    //                 (filled after the barrier, from the packed modes)
    for (int i = threadIdx.x; i < 3 * n; i += NT) {
      const int sc = i / n, x = i - sc * n;
      sWtab[i] = (uint16_t)(x < min(3 << sc, n) ? 32 >> min(31, (2 * x) >> sc) : 0);
    }
    for (int i = threadIdx.x; i < bpg; i += NT) sBest[i] = 0xffffffffu;
    if (threadIdx.x < 64) {
      const int df = threadIdx.x & 31;
      int f0, f1, f2, f3;
      if (threadIdx.x < 32) { f0 = kCubic[df][0]; f1 = kCubic[df][1]; f2 = kCubic[df][2]; f3 = kCubic[df][3]; }
      else { f0 = 16 - (df >> 1); f1 = 32 - (df >> 1); f2 = 16 + (df >> 1); f3 = df >> 1; }   // intra-generic.c:206-214
      f0 *= 4; f1 *= 4; f2 *= 4; f3 *= 4;        // see ang_filter_x4
      sCoef[threadIdx.x] = make_uint2((uint32_t)(f0 & 0xffff) | ((uint32_t)f1 << 16), (uint32_t)(f2 & 0xffff) | ((uint32_t)f3 << 16));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_modes * n; i += NT) {
      const int m = i >> lgn, x = i & (n - 1);
      const uint2 pm = sMode[m];
      const search_mode Sm = unpack_search_mode(pm.x, pm.y);
      const bool on_col = Sm.pdpc == 2 && x < min(3 << Sm.scale, n);
      sSoTab[i] = (uint16_t)(on_col ? 4 * (((256 + (x + 1) * Sm.inv) >> 9) + 1) : 0);
    }
    if (on) {
      // 4x4 blocks never use the smoothed references (intra.c:690-726: no reference filtering for 4x4)
      if constexpr (NFIX != 4) filter_ref_rows(base, base + L.RS, base + 2 * L.RS, base + 3 * L.RS, n, n, L.RS, mytid, tpb);
      if (mytid == 0) sDC[myb] = dc_value(base, base + L.RS, n, n);
    }
    __syncthreads();
    if (on) {
      // u16 image -> pair rows
      uint32_t *pr = sRef + (size_t)myb * L.BRS;
#pragma unroll
      for (int row = 0; row < (NFIX == 4 ? 2 : 4); ++row)
        for (int i = mytid; i < L.RS; i += tpb) {
          const int e = row * L.RS + i;
          pr[e] = (uint32_t)base[e] | (i + 1 < L.RS ? (uint32_t)base[e + 1] << 16 : 0u);
        }
      if (L.AB) {
        // rows 4..7: A(top), B(top), A(left), B(left) of the unfiltered references (see search_tile_angular_ab)
        for (int e = mytid; e < 2 * L.RS; e += tpb) {
          const int side_i = e >= L.RS, i = e - side_i * L.RS;
          const uint16_t *b = base + side_i * L.RS;
          auto at = [&](int j) { return j < L.RS ? (int)b[j] : 0; };
          const int a0 = 16 * (at(i) + 2 * at(i + 1) + at(i + 2)) + 32, a1 = 16 * (at(i + 1) + 2 * at(i + 2) + at(i + 3)) + 32;
          const int b0 = at(i + 2) + at(i + 3) - at(i) - at(i + 1), b1 = at(i + 3) + at(i + 4) - at(i + 1) - at(i + 2);
          pr[(4 + 2 * side_i) * L.RS + i] = (uint32_t)(a0 & 0xffff) | ((uint32_t)a1 << 16);
          pr[(5 + 2 * side_i) * L.RS + i] = (uint32_t)(b0 & 0xffff) | ((uint32_t)b1 << 16);
        }
      }
    }
    __syncthreads();   // also: the scratch image is dead from here on, its space becomes the private strips
  }

  // ---- search: lane = tile, wave = mode ----
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = lane & (tiles - 1);
  constexpr int LB_STEP_NUM = 64;                       // block q of a lane is lb + q * (64 >> lg_tiles)
  const int maxv = px_traits<PX>::maxv;
  const int dshift = px_traits<PX>::depth - 8;

  uint32_t my_best[BPL];
#pragma unroll
  for (int q = 0; q < BPL; ++q) my_best[q] = 0xffffffffu;
  int strip_main = -1;      // which reference row the upper half of this wave's strips currently holds (wave-uniform)
  {
    for (int m = wave; m < n_modes; m += WAVES) {
      const uint2 pm = sMode[m];
      const search_mode S = unpack_search_mode(__builtin_amdgcn_readfirstlane(pm.x), __builtin_amdgcn_readfirstlane(pm.y));
      // the main-row half of a strip does not depend on the mode, only on which row is "main": it is copied when that
      // changes (a wave's negative-angle candidates come as a run of horizontal and a run of vertical modes), the
      // projected half is rebuilt for every mode
      const bool copy_main = S.kind == 2 && S.sd < 0 && S.row_main != strip_main;
      // Everything per-lane is re-derived from one opaque copy of the lane's (block, tile) each iteration: left to itself
      // LICM hoists these pointers and a dozen per-column values (xd0 | i, 2 * (xd0 | i), ...) out of the mode loop and the
      // register allocator then spills them to scratch (measured: 18 MB of scratch writes per launch)
#pragma unroll
      for (int q = 0; q < BPL; ++q) {
      int lane_i = lane;
      if constexpr (T == 8) asm volatile("" : "+v"(lane_i));     // the 4x4 kernel has registers to spare: let LICM hoist
      const int lb_i = (lane_i >> lg_tiles) + q * (LB_STEP_NUM >> lg_tiles), tile_i = lane_i & (tiles - 1);
      const int bb_i = lb_i < here ? lb_i : 0;
      const int xd0 = (tile_i & ((1 << lg_tx) - 1)) * T, yd0 = (tile_i >> lg_tx) * T;
      const uint32_t *ref = sRef + __mul24(bb_i, L.BRS);
      const uint16_t *ob = sOrig + __mul24(bb_i, L.OS) + __mul24(yd0 / T, L.BAND) + xd0;     // yd0 is a multiple of T
      uint32_t *priv = sPriv + __mul24(wave * bpg + bb_i, L.PS);
      const uint32_t *mainr = ref + S.row_main * L.RS, *side = ref + S.row_side * L.RS;
      const bool neg = S.kind == 2 && S.sd < 0;
      if (neg) {
        // extended main pair row of this (block, mode), ext[-j] = side[min((j*inv + 256) >> 9, n)]
        // (intra-generic.c:156-159; j = 0 gives the corner = main[0]):
        //   priv[n - j] = (ext[-j], ext[-j+1]), j = 1..need;  priv[n + i] = main pair i, i = 0..n.
        // The block's `tiles` lanes share the work.
        if (lg_tiles == 0) {
          // one tile per block (n == T): the lane builds its own strip, fully unrolled.  j and inv are uniform,
          // so the projected indices are scalar arithmetic and each sample costs one LDS read.
          int sv[T + 1];
#pragma unroll
          for (int j = 0; j <= T; ++j) sv[j] = pr_sample(side, min((j * S.inv + 256) >> 9, T));
#pragma unroll
          for (int j = 1; j <= T; ++j) priv[T - j] = (uint32_t)sv[j] | ((uint32_t)sv[j - 1] << 16);
          if (copy_main) {
#pragma unroll
            for (int i = 0; i <= T; ++i) priv[T + i] = mainr[i];
          }
        } else {
          const int need = min(n, (__mul24(-S.sd, n) + 31) >> 5);   // deepest row reaches ext[-need]
          for (int e = n - need + tile; e < n; e += tiles) {
            const int j = n - e;
            const int a0 = pr_sample(side, min((__mul24(j, S.inv) + 256) >> 9, n));
            const int a1 = pr_sample(side, min((__mul24(j - 1, S.inv) + 256) >> 9, n));
            priv[e] = (uint32_t)a0 | ((uint32_t)a1 << 16);
          }
          if (copy_main)
            for (int e = n + tile; e < 2 * n + 1; e += tiles) priv[e] = mainr[e - n];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      uint32_t d[T][T / 2];
      uint32_t sad = 0;
      const uint16_t *ot = ob + (S.transposed ? L.OT : 0);
      if (S.kind == 2) {
        const uint32_t *rowp = neg ? priv + n : mainr;
        const uint16_t *wrow = sWtab + S.scale * n, *sorow = sSoTab + m * n;     // wave-uniform rows; the lane adds its xd0
        if (T == 8 && sizeof(PX) == 1 && L.AB && S.coef == 32 && !neg) {
          // smoothing taps, non-negative angle: A + h * B on the precomputed rows (S.row_main is an unfiltered row here)
          if constexpr (T == 8 && sizeof(PX) == 1) {
            const uint32_t *rowA = ref + (4 + 2 * S.row_main) * L.RS, *rowB = rowA + L.RS;
            if (S.pdpc == 2) search_tile_angular_ab<T, 2>(S, side, rowA, rowB, wrow, sorow, n, xd0, yd0, ot, d, sad);
            else search_tile_angular_ab<T, 0>(S, side, rowA, rowB, wrow, sorow, n, xd0, yd0, ot, d, sad);
          }
        } else if (S.pdpc == 0) {
          if (S.noclamp) search_tile_angular<T, 0, false, sizeof(PX) == 1>(S, mainr, side, rowp, sCoef, wrow, sorow, n, xd0, yd0, ot, maxv, d, sad);
          else search_tile_angular<T, 0, true, sizeof(PX) == 1>(S, mainr, side, rowp, sCoef, wrow, sorow, n, xd0, yd0, ot, maxv, d, sad);
        } else if (S.pdpc == 2) {
          // 8-bit 16x16 / 32x32: the smoothing modes took the A + h * B path above, which leaves only the two integer-slope
          // modes for the unclamped variant -- they go through the clamped one (same result, 64 ops more for two modes)
          // and the register-hungriest variant is not instantiated at all (it was the one that spilled)
          constexpr bool kNoclampVariant = NFIX < 16;   // (10-bit 16x16 / 32x32: their smoothing modes take the clamped variant too)
          if (kNoclampVariant && S.noclamp) search_tile_angular<T, 2, false, sizeof(PX) == 1>(S, mainr, side, rowp, sCoef, wrow, sorow, n, xd0, yd0, ot, maxv, d, sad);
          else search_tile_angular<T, 2, true, sizeof(PX) == 1>(S, mainr, side, rowp, sCoef, wrow, sorow, n, xd0, yd0, ot, maxv, d, sad);
        } else search_tile_angular<T, 3, false, sizeof(PX) == 1>(S, mainr, side, rowp, sCoef, wrow, sorow, n, xd0, yd0, ot, maxv, d, sad);   // pure H/V: integer phase
      } else if (S.kind == 0) search_tile_nonangular<T, true>(S, mainr, side, 0, n, lgn, xd0, yd0, ot, d, sad);
      else search_tile_nonangular<T, false>(S, mainr, side, sDC[bb_i], n, lgn, xd0, yd0, ot, d, sad);
      uint32_t satd;
      if constexpr (T == 8) satd = satd8_tile_lane(d); else satd = satd4_tile_lane(d);
      if (lg_tiles >= 2) {
        satd += dpp_xor1(satd); sad += dpp_xor1(sad);
        satd += dpp_xor2(satd); sad += dpp_xor2(sad);
        if (lg_tiles == 4) {
          satd += dpp_mirror8(satd); sad += dpp_mirror8(sad);
          satd += dpp_mirror16(satd); sad += dpp_mirror16(sad);
        }
      }
      if (neg) __builtin_amdgcn_wave_barrier();   // strip is rewritten by the next negative mode
      if (lb_i < here && tile == 0) {
        // search_intra.c:158: min(SATD, 2*SAD) with the NxN strategy functions' depth shifts
        // (satd_4x4 is unshifted, picture-generic.c:170; everything else >> depth-8)
        const uint32_t c_satd = satd >> (T == 4 ? 0 : dshift);
        const uint32_t c_sad = sad >> dshift;
        const uint32_t c = min(c_satd, 2 * c_sad);
        if (costs) costs[(size_t)(blk0 + lb_i) * n_modes + m] = c;
        my_best[q] = min(my_best[q], (c << 7) | (uint32_t)m);       // c < 2^25: at most 2 * 1024 samples * 255 (after the depth shift)
      }
      }   // q
      if (copy_main) strip_main = S.row_main;
    }
  }
  // fused arg-min (the strict "<" scan of search_intra.c:1089-1101: ties keep the earlier candidate):
  // every wave contributes the best of its share of the candidates
  if (best_mode) {
#pragma unroll
    for (int q = 0; q < BPL; ++q) {
      const int lb = (lane >> lg_tiles) + q * (LB_STEP_NUM >> lg_tiles);
      if (lb < here && tile == 0) atomicMin(&sBest[lb], my_best[q]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < here; i += WAVES * 64) {
      const uint32_t k = sBest[i];
      best_mode[blk0 + i] = modes[k & 127];
      if (best_cost) best_cost[blk0 + i] = k >> 7;
    }
  }
}

static int launch_intra_search(int bitdepth, const void *rec, int rec_stride, const void *orig, int orig_stride,
                               int size, const uvghip_intra_blk_t *blks, int n, const int8_t *modes, int n_modes,
                               uint32_t *costs, int8_t *best_mode, uint32_t *best_cost, void *stream, const char *who, int pic_rows = 0)
{
  UVGHIP_REQUIRE_READY();
  if (!(size == 4 || size == 8 || size == 16 || size == 32) || n_modes < 1 || n_modes > 128 || pic_rows < 0)
    return uvghip_set_error(hipErrorInvalidValue, who);
  if (n <= 0) return 0;
  const int tiles = size == 4 ? 1 : (size / 8) * (size / 8);
  const int bpl = size == 4 ? UVGHIP_SEARCH_BPL4 : 1;
  const int bpg = 64 * bpl / tiles;
  const int grid = (n + bpg - 1) / bpg;
  hipStream_t st = uvghip_stream(stream);
  // 8x8 tiles: 8 waves per workgroup (two workgroups per CU = 4 waves per SIMD); 4x4: 4 waves, many workgroups
#define LAUNCH(PX, T, W, B, NF) do { const search_layout L = make_search_layout(size, bpg, n_modes, W, (int)sizeof(PX)); \
    /* allow more than the default 64 KiB of dynamic LDS: once per instantiation AND device (the attribute is per device; \
       strategy pointers are entered concurrently from all worker threads, threadqueue.c:275) */ \
    { static std::mutex lds_m; static uint64_t lds_done = 0; int dev_ = 0; UVGHIP_TRY(hipGetDevice(&dev_)); \
      std::lock_guard<std::mutex> lk_(lds_m); \
      if (!((lds_done >> (dev_ & 63)) & 1)) { \
        UVGHIP_TRY(hipFuncSetAttribute((const void *)intra_search_kernel<PX, T, W, B, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        lds_done |= 1ull << (dev_ & 63); } } \
    intra_search_kernel<PX, T, W, B, NF><<<grid, W * 64, L.total, st>>>((const PX *)rec, rec_stride, (const PX *)orig, orig_stride, size, blks, n, modes, n_modes, costs, best_mode, best_cost, pic_rows); } while (0)
  if (bitdepth == 8) {
    if (size == 4) LAUNCH(uint8_t, 4, 4, UVGHIP_SEARCH_BPL4, 4);
    else if (size == 8) LAUNCH(uint8_t, 8, UVGHIP_SEARCH_WAVES, 1, 8);
    else if (size == 16) LAUNCH(uint8_t, 8, UVGHIP_SEARCH_WAVES, 1, 16);
    else LAUNCH(uint8_t, 8, UVGHIP_SEARCH_WAVES, 1, 32);
  } else {
    if (size == 4) LAUNCH(uint16_t, 4, 4, UVGHIP_SEARCH_BPL4, 4);
    else if (size == 8) LAUNCH(uint16_t, 8, UVGHIP_SEARCH_WAVES, 1, 8);
    else if (size == 16) LAUNCH(uint16_t, 8, UVGHIP_SEARCH_WAVES, 1, 16);
    else LAUNCH(uint16_t, 8, UVGHIP_SEARCH_WAVES, 1, 32);
  }
#undef LAUNCH
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_intra_search_batch(int bitdepth, const void *rec, int rec_stride, const void *orig, int orig_stride,
                                         int size, const uvghip_intra_blk_t *blks, int n, const int8_t *modes,
                                         int n_modes, uint32_t *costs, void *stream)
{
  if (!costs) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return launch_intra_search(bitdepth, rec, rec_stride, orig, orig_stride, size, blks, n, modes, n_modes, costs, nullptr, nullptr,
                             stream, __func__);
}

extern "C" int uvghip_intra_search_best_batch(int bitdepth, const void *rec, int rec_stride, const void *orig, int orig_stride,
                                              int size, const uvghip_intra_blk_t *blks, int n, const int8_t *modes,
                                              int n_modes, int8_t *best_mode, uint32_t *best_cost, uint32_t *costs,
                                              void *stream)
{
  if (!best_mode) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return launch_intra_search(bitdepth, rec, rec_stride, orig, orig_stride, size, blks, n, modes, n_modes, costs, best_mode,
                             best_cost, stream, __func__);
}

extern "C" int uvghip_intra_search_best_stacked_batch(int bitdepth, const void *rec, int rec_stride, const void *orig, int orig_stride,
                                                      int size, const uvghip_intra_blk_t *blks, int n, const int8_t *modes,
                                                      int n_modes, int8_t *best_mode, uint32_t *best_cost, uint32_t *costs,
                                                      int pic_rows, void *stream)
{
  if (!best_mode) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return launch_intra_search(bitdepth, rec, rec_stride, orig, orig_stride, size, blks, n, modes, n_modes, costs, best_mode,
                             best_cost, stream, __func__, pic_rows);
}

// =================================================== drop-in strategy layer ====
// angular_pred_func / intra_pred_planar_func / pdpc_planar_dc_func
// (src/strategies/strategies-intra.h:47-79).  The reference hands these
// functions ready-made reference arrays, so the per-call path uploads the two
// arrays, runs a one-block launch of a kernel that only predicts, and
// downloads the block.
namespace {

template <typename PX>
__global__ void __launch_bounds__(256)
pred_from_rows_kernel(const PX *__restrict__ ref_top, const PX *__restrict__ ref_left, int refn, int kind, int mode,
                      int is_chroma, int w, int h, int mrl, int isp, PX *__restrict__ dst)
{
  // kind 0: angular (mode = already wide-angle-corrected mode), 1: planar, 2: pdpc on dst (mode 0/1)
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t *top = smem, *left = smem + refn;
  for (int i = threadIdx.x; i < refn; i += blockDim.x) { top[i] = ref_top[i]; left[i] = ref_left[i]; }
  __syncthreads();
  const ref_rows R{top, left, top, left};
  const int maxv = px_traits<PX>::maxv;
  mode_info M;
  if (kind == 0) {
    // rebuild the angular parameters from the corrected mode (intra-generic.c:118-136,206-246)
    const int lw = ilog2_dev(w), lh = ilog2_dev(h);
    const int pm = mode, vertical = pm >= 34;
    const int mode_disp = vertical ? pm - 50 : 18 - pm, amd = abs(mode_disp);
    const int sd = (mode_disp < 0 ? -1 : 1) * kSampleDisp[amd];
    const int d50 = abs(pm - 50), d18 = abs(pm - 18), dist = d50 < d18 ? d50 : d18;
    M.mode = 2; M.pred_mode = (int16_t)pm; M.filtered = 0; M.vertical = (int8_t)vertical;
    M.sample_disp = (int16_t)sd; M.inv_disp = kInvDisp[amd];
    int scale = (vertical ? lh : lw) - kPreScale[amd]; if (scale > 2) scale = 2;
    M.scale = (int8_t)scale; M.frac = (abs(sd) & 31) != 0;
    M.use_cubic = !(dist > kDistThres[(lw + lh) >> 1] && M.frac);
    int pdpc = (w >= 4 && h >= 4);
    if (sd != 0 && pm > 1 && pm < 67) { if (mode_disp < 0) pdpc = 0; else if (mode_disp > 0) pdpc = pdpc && scale >= 0; }
    M.pdpc = (int8_t)pdpc;
  } else {
    M = make_mode_info(kind == 1 ? 0 : mode, w, h, is_chroma);
    M.filtered = 0;
    if (kind == 1) M.pdpc = 0;      // uvg_intra_pred_planar alone does no PDPC
  }
  const bool transposed = kind == 0 && !M.vertical;
  const int wd = transposed ? h : w, hd = transposed ? w : h;
  const int segs = w * h / 4;
  for (int s = threadIdx.x; s < segs; s += blockDim.x) {
    const int yd = s / (wd / 4), xd0 = (s - yd * (wd / 4)) * 4;
    int v[4];
    if (kind == 2) {
      // PDPC only: dst holds the un-combined prediction (intra-generic.c:429-435)
      const int lw = ilog2_dev(w), lh = ilog2_dev(h), scale = (lw + lh - 2) >> 2;
      const int wt = 32 >> min(31, (yd << 1) >> scale), l = left[yd + 1];
      for (int i = 0; i < 4; ++i) {
        const int x = xd0 + i, wl = 32 >> min(31, (x << 1) >> scale), c = dst[yd * w + x];
        dst[yd * w + x] = (PX)(c + ((wl * (l - c) + wt * ((int)top[x + 1] - c) + 32) >> 6));
      }
      continue;
    }
    predict_row<4>(M, R, 0, is_chroma, wd, hd, yd, xd0, maxv, v);
    for (int i = 0; i < 4; ++i) {
      const int px = transposed ? yd : xd0 + i, py = transposed ? xd0 + i : yd;
      dst[py * w + px] = (PX)v[i];
    }
  }
}

template <typename PX>
void percall_rows(int kind, int mode, int is_chroma, int w, int h, const PX *ref_top, const PX *ref_left, PX *dst,
                  int mrl, int isp)
{
  const int refn = ref_len(w, h);
  const size_t rb = (size_t)refn * sizeof(PX), db = (size_t)w * h * sizeof(PX);
  percall_ctx *c = percall_get(2 * rb + db + 1024);
  const size_t ot = c->take(rb), ol = c->take(rb), od = c->take(db);
  memcpy(c->hp<PX>(ot), ref_top, rb);
  memcpy(c->hp<PX>(ol), ref_left, rb);
  if (kind == 2) memcpy(c->hp<PX>(od), dst, db);
  c->upload(0, c->used);
  pred_from_rows_kernel<PX><<<1, 256, (size_t)2 * refn * 2, c->stream>>>(c->dp<PX>(ot), c->dp<PX>(ol), refn, kind, mode, is_chroma, w, h, mrl, isp, c->dp<PX>(od));
  if (hipGetLastError() != hipSuccess) c->fail("intra launch");
  c->download(od, db);
  c->sync();
  memcpy(dst, c->hp<PX>(od), db);
}

template <typename PX>
void angular_pred_hip(const ref_cu_loc *cu_loc, const int_fast8_t intra_mode, const int_fast8_t channel_type,
                      const PX *in_ref_above, const PX *in_ref_left, PX *dst, const uint8_t multi_ref_idx,
                      const uint8_t isp_mode, const int cu_dim)
{
  const int w = channel_type == 0 ? cu_loc->width : cu_loc->chroma_width;
  const int h = channel_type == 0 ? cu_loc->height : cu_loc->chroma_height;
  if (multi_ref_idx || isp_mode || w < 4 || h < 4) {
    fprintf(stderr, "uvg266hip: angular_pred with MRL/ISP/sub-4 blocks is not supported by the hip strategy; "
                    "run with --no-mrl --no-isp or override UVG266_OVERRIDE_angular_pred=generic\n");
    abort();
  }
  percall_rows<PX>(0, intra_mode, channel_type != 0, w, h, in_ref_above, in_ref_left, dst, 0, 0);
}
template <typename PX>
void intra_pred_planar_hip(const ref_cu_loc *cu_loc, int color, const PX *ref_top, const PX *ref_left, PX *dst)
{
  const int w = color == 0 ? cu_loc->width : cu_loc->chroma_width, h = color == 0 ? cu_loc->height : cu_loc->chroma_height;
  percall_rows<PX>(1, 0, color != 0, w, h, ref_top, ref_left, dst, 0, 0);
}
// uvg_intra_ref (intra.h:48-51): { uvg_pixel left[INTRA_REF_LENGTH]; uvg_pixel top[INTRA_REF_LENGTH]; }, INTRA_REF_LENGTH = 358
template <typename PX>
void pdpc_planar_dc_hip(const int mode, const ref_cu_loc *cu_loc, const int color, const PX *used_ref, PX *dst)
{
  const int w = color == 0 ? cu_loc->width : cu_loc->chroma_width, h = color == 0 ? cu_loc->height : cu_loc->chroma_height;
  percall_rows<PX>(2, mode, color != 0, w, h, used_ref + 358, used_ref, dst, 0, 0);
}


// intra_pred_filtered_dc_func (strategies-intra.h; intra-generic.c:371-402): HEVC-style smoothed DC, registered
// upstream without a caller -- bound for table completeness.  One wave: lanes sum the two reference sides (shuffle
// reduction), then every lane writes samples; the DC sum honours multi_ref_idx, the boundary filter does not.
template <typename PX>
__global__ void __launch_bounds__(64)
filtered_dc_kernel(const PX *__restrict__ top, const PX *__restrict__ left, int log2w, int mrl, PX *__restrict__ dst)
{
  const int n = 1 << log2w, l = threadIdx.x;
  int part = l < n ? (int)top[l + 1 + mrl] + (int)left[l + 1 + mrl] : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
  const int dc = (int)(PX)((__builtin_amdgcn_readfirstlane(part) + n) >> (log2w + 1));
  for (int i = l; i < n * n; i += 64) {
    const int y = i >> log2w, x = i & (n - 1);
    int v = dc;
    if (i == 0) v = ((int)left[1] + 2 * dc + (int)top[1] + 2) >> 2;
    else if (y == 0) v = ((int)top[x + 1] + 3 * dc + 2) >> 2;
    else if (x == 0) v = ((int)left[y + 1] + 3 * dc + 2) >> 2;
    dst[i] = (PX)v;
  }
}
template <typename PX>
void intra_pred_filtered_dc_hip(const int_fast8_t log2_width, const PX *ref_top, const PX *ref_left, PX *out_block,
                                const uint8_t multi_ref_idx)
{
  const int n = 1 << log2_width, len = n + 1 + multi_ref_idx;
  percall_ctx *c = percall_get((size_t)(2 * len + n * n) * sizeof(PX) + 1024);
  if (log2_width < 2 || log2_width > 5) c->fail("intra_pred_filtered_dc: log2_width outside 2..5");
  const size_t ot = c->stage_block(ref_top, (size_t)len, len, 1, sizeof(PX));
  const size_t ol = c->stage_block(ref_left, (size_t)len, len, 1, sizeof(PX));
  c->upload(0, c->used);
  const size_t oo = c->take((size_t)n * n * sizeof(PX));
  filtered_dc_kernel<PX><<<1, 64, 0, c->stream>>>(c->dp<PX>(ot), c->dp<PX>(ol), log2_width, multi_ref_idx, c->dp<PX>(oo));
  if (hipGetLastError() != hipSuccess) c->fail("filtered dc launch");
  c->download(oo, (size_t)n * n * sizeof(PX));
  c->sync();
  memcpy(out_block, c->hp<PX>(oo), (size_t)n * n * sizeof(PX));
}

}  // namespace

int uvghip_register_mip(void *opaque, uint8_t bitdepth);   // mip.hip

extern "C" int uvg_strategy_register_intra_hip(void *opaque, uint8_t bitdepth)
{
  if (!uvghip_ready() && uvghip_init(0) != 0) return 0;
  int ok = 1;
  if (bitdepth == 8) {
    ok &= uvghip_do_register(opaque, "angular_pred", (void *)&angular_pred_hip<uint8_t>);
    ok &= uvghip_do_register(opaque, "intra_pred_planar", (void *)&intra_pred_planar_hip<uint8_t>);
    ok &= uvghip_do_register(opaque, "pdpc_planar_dc", (void *)&pdpc_planar_dc_hip<uint8_t>);
    ok &= uvghip_do_register(opaque, "intra_pred_filtered_dc", (void *)&intra_pred_filtered_dc_hip<uint8_t>);
  } else {
    ok &= uvghip_do_register(opaque, "angular_pred", (void *)&angular_pred_hip<uint16_t>);
    ok &= uvghip_do_register(opaque, "intra_pred_planar", (void *)&intra_pred_planar_hip<uint16_t>);
    ok &= uvghip_do_register(opaque, "pdpc_planar_dc", (void *)&pdpc_planar_dc_hip<uint16_t>);
    ok &= uvghip_do_register(opaque, "intra_pred_filtered_dc", (void *)&intra_pred_filtered_dc_hip<uint16_t>);
  }
  ok &= uvghip_register_mip(opaque, bitdepth);
  return ok;
}
