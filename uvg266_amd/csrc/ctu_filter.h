// The in-loop filters of ONE CTU inside the persistent P / B kernel (ctu_search_pb.hip): what encoder_state_worker_encode_lcu_search does
// after uvg_search_lcu (src/encoderstate.c:841-853) -- uvg_filter_deblock_lcu, uvg_sao_search_lcu, encoder_sao_reconstruct -- so that a
// picture becomes final CTU by CTU and pictures that refer to it can be in flight behind it on the reference's own schedule
// (encoderstate.c:1084-1116: CTU (x, y) of a picture waits for CTU (x + 2, y + 1) of its reference; fracmv_within_tile,
// search_inter.c:94-149, keeps the vectors inside what is final then).
//
// NOT the reference's schedule of side effects.  The reference filters in place and saves the unfiltered / pre-SAO lines the later
// CTUs need (hor_buf_search, ver_buf_before_sao, ...).  Here the search's reconstruction `rec` stays unfiltered for good, and a CTU
// *pulls* what it can finish from it:
//   tile    T = rec[64 cx - 16, 64 cx + 64) x [64 cy - 16, 64 cy + 64) in LDS (needs the CTU, its left, upper and upper-left neighbour:
//           all searched before this CTU started), all vertical edges, then all horizontal edges -- VVC sizes the filters so that edges
//           of one direction never touch each other's samples (filter.c:587-644), so a tile gives the picture's values;
//   D(x, y) = [64 cx - 8, 64 cx + 56) x [64 cy - 8, 64 cy + 56) of the tile is final (the edges on the CTU's right / lower boundary
//           reach 7 samples back: DEBLOCK_DELAY_PX 8, global.h:240) -> the deblocked picture `dbk`; out to the picture's edge in its
//           last column / row;
//   S(x, y) = the CTU's own block in the tile, with the horizontal edges of its last 8 columns left out, IS the block
//           uvg_sao_search_lcu sees (deblocked by the CTU's own edges only: uvghip_deblock_frame_sao_snapshot) -> statistics -> the
//           decision, chained through the two SAO models along the row and from the first CTU of the row above (sao_decide_dev.h);
//   F(x, y) = [64 cx - 10, 64 cx + 54) x [64 cy - 10, 64 cy + 54) (SAO_DELAY_PX 10: one more sample for the edge classes' neighbours,
//           encoderstate.c:316-364) of dbk + the decisions of the up to four CTUs it overlaps -> the output picture.
// Flags (agent scope, release / acquire): sao_done[k] after D is written and the decision is out; final_done[k] after F is written AND
// the left, upper and upper-right CTU are final -- so a set flag means: final is everything up and left of the CTU's lower right corner
// minus 10, and of the corners of the CTUs (x + j, y - j) up the diagonal -- the shape a vector of a picture in flight may reach into.
#pragma once
#include "deblock_dev.h"
#include "sao_decide_dev.h"

namespace ctuf {

struct filt_pic {
  void *dbk_y, *dbk_u, *dbk_v;            // the deblocked picture (workspace)
  void *out_y, *out_u, *out_v;            // the output picture: what uvg_encoder_encode returns, the later pictures' reference
  int32_t dbk_stride, dbk_stride_c, out_stride, out_stride_c;
  int32_t *sao_info;                      // [ctu][2][17]: sao_info_t luma, chroma
  uint16_t *sao_models;                   // [ctu][6]: the two SAO models after the CTU's SAO syntax
  double lambda;
  int32_t sao_type, slice_type, qp, is_b;
};

struct filt_ctu {                         // one CTU's job
  const void *rec_y, *rec_u, *rec_v;      // the search's reconstruction (unfiltered)
  const void *src_y, *src_u, *src_v;
  int32_t rec_stride, rec_stride_c, src_stride, src_stride_c;
  const uvghip_scu_t *scu;
  int32_t scu_stride;
  int32_t W, H, cx, cy, wc, hc;
  int32_t *sao_done, *final_done;         // this picture's flags [ctus]
};

__device__ __forceinline__ int sgn3(int v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int eo_cat(int a, int b, int c) { return (0x43021 >> (4 * (2 + sgn3(c - a) + sgn3(c - b)))) & 7; }      // sao_shared_generics.h:42-50
__device__ static const int8_t kEo[4][4] = {{-1, 0, 1, 0}, {0, -1, 0, 1}, {-1, -1, 1, 1}, {1, -1, -1, 1}};                         // ax, ay, bx, by (sao.h:71-76)

enum { TP = 80, TPC = 40, REPL = 16, ACC = 104 };

template <typename PX> struct filt_lds {
  PX tile[TP * TP];                       // luma tile; then the two chroma tiles (2 x 40 x 40)
  int32_t acc[REPL][ACC];                 // statistics, REPL copies to thin out the atomics' collisions: [0, 40) edge (class * 10 + sum / 5 + count by category), [40, 104) band
  int32_t E[3][40], B[3][64];
  saod::cand cands[2];
  saod::sao_info L, C;
};

__device__ inline void wait_set(const int32_t *flag)          // one lane
{
  int naps = 1;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(8);
    if (naps < 8) naps <<= 1;
  }
}

// rec -> tile: rows [ty0, ty1) x columns [tx0, tx1) of a plane (bounds multiples of 4)
template <typename PX> __device__ inline void load_tile(PX *tile, int pitch, const PX *rec, int stride, int tx0, int ty0, int tx1, int ty1)
{
  const int w4 = (tx1 - tx0) >> 2, n = w4 * (ty1 - ty0);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / w4, c = (i - r * w4) * 4;
    int v[4];
    load4(rec + (size_t)(ty0 + r) * stride + tx0 + c, v);
    PX *t = tile + r * pitch + c;
    t[0] = (PX)v[0]; t[1] = (PX)v[1]; t[2] = (PX)v[2]; t[3] = (PX)v[3];
  }
}

template <typename PX> __device__ inline void store_region(const PX *tile, int pitch, int tx0, int ty0, PX *dst, int stride, int x0, int y0, int x1, int y1)
{
  const int w = x1 - x0, n = w * (y1 - y0);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / w, c = i - r * w;
    dst[(size_t)(y0 + r) * stride + x0 + c] = tile[(y0 + r - ty0) * pitch + x0 + c - tx0];
  }
}

// SAO statistics of the block [x0, x0 + bw) x [y0, y0 + bh) of the tile against the source (calc_sao_edge_dir x 4 + calc_sao_bands:
// sao-generic.c:51-81, sao.c:268-285) -> E[40], B[64] in uvghip_sao_stats_batch's layout
template <typename PX> __device__ inline void stats_block(filt_lds<PX> *F, const PX *tile, int pitch, int tx0, int ty0, int x0, int y0, int bw, int bh, const PX *src, int sstride,
                                                          int32_t *E, int32_t *B)
{
  constexpr int bshift = px_traits<PX>::depth - 5;
  for (int i = threadIdx.x; i < REPL * ACC; i += blockDim.x) (&F->acc[0][0])[i] = 0;
  __syncthreads();
  int32_t *acc = F->acc[threadIdx.x & (REPL - 1)];
  const int n = bw * bh;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int y = i / bw, x = i - y * bw;
    const PX *t = tile + (y0 + y - ty0) * pitch + (x0 + x - tx0);
    const int c = t[0], d = (int)src[(size_t)(y0 + y) * sstride + x0 + x] - c;
    atomicAdd(&acc[40 + (c >> bshift)], d);
    atomicAdd(&acc[72 + (c >> bshift)], 1);
    if (y >= 1 && y < bh - 1 && x >= 1 && x < bw - 1) {
      const int l = t[-1], r = t[1], u = t[-pitch], dn = t[pitch], ul = t[-pitch - 1], ur = t[-pitch + 1], dl = t[pitch - 1], dr = t[pitch + 1];
      const int c0 = eo_cat(l, r, c), c1 = eo_cat(u, dn, c), c2 = eo_cat(ul, dr, c), c3 = eo_cat(ur, dl, c);
      atomicAdd(&acc[0 + c0], d);  atomicAdd(&acc[5 + c0], 1);
      atomicAdd(&acc[10 + c1], d); atomicAdd(&acc[15 + c1], 1);
      atomicAdd(&acc[20 + c2], d); atomicAdd(&acc[25 + c2], 1);
      atomicAdd(&acc[30 + c3], d); atomicAdd(&acc[35 + c3], 1);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ACC; e += blockDim.x) {
    int s = 0;
    for (int r = 0; r < REPL; ++r) s += F->acc[r][e];
    if (e < 40) E[e] = s; else B[e - 40] = s;
  }
  __syncthreads();
}

// uvg_sao_reconstruct's per-sample rule (sao.c:302-361, sao-generic.c:84-124) on the region [x0, x1) x [y0, y1) of a plane: every sample
// with the decision of the CTU it lies in; edge classes leave the picture's outermost rows / columns alone.
template <typename PX> __device__ inline void sao_region(const PX *dbk, int dstride, PX *out, int ostride, int pw, int ph, int x0, int y0, int x1, int y1, int shift, int wc,
                                                         const int32_t *info, int comp)
{
  constexpr int maxv = px_traits<PX>::maxv;
  constexpr int bshift = px_traits<PX>::depth - 5;
  if (x1 <= x0 || y1 <= y0) return;
  const int size = 1 << shift;
  for (int qy = y0 >> shift; qy <= (y1 - 1) >> shift; ++qy)
    for (int qx = x0 >> shift; qx <= (x1 - 1) >> shift; ++qx) {
      const int32_t *s = info + ((size_t)(qy * wc + qx) * 2 + (comp ? 1 : 0)) * 17;
      const int type = s[0], cls = s[1], band_pos = s[5 + (comp == 2)];
      const int32_t *off = s + 7 + (comp == 2 ? 5 : 0);
      const int ax0 = max(x0, qx * size), ay0 = max(y0, qy * size), ax1 = min(x1, (qx + 1) * size), ay1 = min(y1, (qy + 1) * size);
      const int w = ax1 - ax0, n = w * (ay1 - ay0);
      const int ax = type == 2 ? kEo[cls][0] : 0, ay = type == 2 ? kEo[cls][1] : 0, bx = type == 2 ? kEo[cls][2] : 0, by = type == 2 ? kEo[cls][3] : 0;
      const int o0 = off[0], o1 = off[1], o2 = off[2], o3 = off[3], o4 = off[4];
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / w, x = ax0 + (i - r * w), y = ay0 + r;
        const PX *p = dbk + (size_t)y * dstride + x;
        const int c = p[0];
        int v = c;
        if (type == 1) {
          const int band = (c >> bshift) - band_pos;
          if (band >= 0 && band <= 3) v = clampi(c + (band == 0 ? o1 : band == 1 ? o2 : band == 2 ? o3 : o4), 0, maxv);
        } else if (type == 2) {
          if (x + ax >= 0 && x + bx >= 0 && x + ax < pw && x + bx < pw && y + ay >= 0 && y + by >= 0 && y + ay < ph && y + by < ph) {
            const int cat = eo_cat(p[(ptrdiff_t)ay * dstride + ax], p[(ptrdiff_t)by * dstride + bx], c);
            v = clampi(c + (cat == 0 ? o0 : cat == 1 ? o1 : cat == 2 ? o2 : cat == 3 ? o3 : o4), 0, maxv);
          }
        }
        out[(size_t)y * ostride + x] = (PX)v;
      }
    }
}

// The job.  smem: the workgroup's dynamic LDS image (free between two CTUs), at least sizeof(filt_lds<PX>).
#if !defined(CTUF_CALL)
#define CTUF_CALL __attribute__((noinline))          // (inside the P / B search kernel: a call, not a copy of the stage in the kernel's body)
#endif
template <typename PX>
__device__ CTUF_CALL void filter_ctu(unsigned char *smem, const filt_pic &P, const filt_ctu &J)
{
  filt_lds<PX> *F = reinterpret_cast<filt_lds<PX> *>(smem);
  const int W = J.W, H = J.H, cx = J.cx, cy = J.cy, wc = J.wc, hc = J.hc, k = cy * wc + cx;
  const bool last_col = cx == wc - 1, last_row = cy == hc - 1;
  dbk_cfg cfg;
  cfg.beta_offset_div2 = 0; cfg.tc_offset_div2 = 0; cfg.slice_is_b = P.is_b; cfg.frame_qp = P.qp; cfg.has_qp_map = 0; cfg.snapshot = 0;
  // the tile and the regions, in luma samples
  const int tx0 = max(0, 64 * cx - 16), ty0 = max(0, 64 * cy - 16), tx1 = min(W, 64 * cx + 64), ty1 = min(H, 64 * cy + 64);
  const int dx0 = cx ? 64 * cx - 8 : 0, dy0 = cy ? 64 * cy - 8 : 0, dx1 = last_col ? W : 64 * cx + 56, dy1 = last_row ? H : 64 * cy + 56;
  const int bx0 = 64 * cx, by0 = 64 * cy, bw = tx1 - bx0, bh = ty1 - by0;
  __syncthreads();
  // ---- luma ----
  {
    PX *tile = F->tile;
    load_tile<PX>(tile, TP, (const PX *)J.rec_y, J.rec_stride, tx0, ty0, tx1, ty1);
    __syncthreads();
    PX *base = tile - ((ptrdiff_t)ty0 * TP + tx0);          // base[y * TP + x]: the tile's sample of picture position (x, y)
    {
      const int ex0 = max(4, dx0), nex = (tx1 - ex0) >> 2, ney = (ty1 - ty0) >> 2;
      for (int i = threadIdx.x; i < nex * ney; i += blockDim.x) {
        const int r = i / nex, ex = ex0 + (i - r * nex) * 4, y = ty0 + r * 4;
        if (J.scu[(size_t)(y >> 2) * J.scu_stride + (ex >> 2)].luma_edges & 1) luma_segment<PX>(base, TP, J.scu, J.scu_stride, ex, y, false, cfg, false);
      }
    }
    __syncthreads();
    {
      const int fy0 = max(4, dy0), nfy = (ty1 - fy0) >> 2, nfx = (dx1 - dx0) >> 2;
      for (int i = threadIdx.x; i < nfx * nfy; i += blockDim.x) {
        const int r = i / nfx, x = dx0 + (i - r * nfx) * 4, fy = fy0 + r * 4;
        if (J.scu[(size_t)(fy >> 2) * J.scu_stride + (x >> 2)].luma_edges & 2) luma_segment<PX>(base, TP, J.scu, J.scu_stride, x, fy, true, cfg, false);
      }
    }
    __syncthreads();
    store_region<PX>(tile, TP, tx0, ty0, (PX *)P.dbk_y, P.dbk_stride, dx0, dy0, dx1, dy1);
    if (P.sao_type) stats_block<PX>(F, tile, TP, tx0, ty0, bx0, by0, bw, bh, (const PX *)J.src_y, J.src_stride, F->E[0], F->B[0]);
    __syncthreads();
  }
  // ---- chroma: both planes side by side in the tile's memory ----
  {
    PX *tu = F->tile, *tv = F->tile + TPC * TPC;
    const int cx0 = tx0 >> 1, cy0 = ty0 >> 1, cx1 = tx1 >> 1, cy1 = ty1 >> 1;
    load_tile<PX>(tu, TPC, (const PX *)J.rec_u, J.rec_stride_c, cx0, cy0, cx1, cy1);
    load_tile<PX>(tv, TPC, (const PX *)J.rec_v, J.rec_stride_c, cx0, cy0, cx1, cy1);
    __syncthreads();
    PX *bu = tu - ((ptrdiff_t)cy0 * TPC + cx0), *bv = tv - ((ptrdiff_t)cy0 * TPC + cx0);
    {
      // vertical edges on the 8-sample chroma grid (every fourth 4x4 unit column), from the CTU's left boundary on; a thread per unit row
      const int ex0 = cx ? 64 * cx : 16, nex = ex0 < tx1 ? (tx1 - ex0 + 15) >> 4 : 0, ney = (ty1 - ty0) >> 2;
      for (int i = threadIdx.x; i < nex * ney; i += blockDim.x) {
        const int r = i / nex, ex = ex0 + (i - r * nex) * 16, y = ty0 + r * 4;
        const uvghip_scu_t &c = J.scu[(size_t)(y >> 2) * J.scu_stride + (ex >> 2)];
        if ((c.luma_edges & 1) && (c.chroma_edges & 1)) chroma_segment<PX>(bu, bv, TPC, J.scu, J.scu_stride, ex >> 1, y >> 1, false, cfg, false);
      }
    }
    __syncthreads();
    {
      const int fy0 = cy ? 64 * cy : 16, nfy = fy0 < ty1 ? (ty1 - fy0 + 15) >> 4 : 0, nfx = (dx1 - dx0) >> 2;
      for (int i = threadIdx.x; i < nfx * nfy; i += blockDim.x) {
        const int r = i / nfx, x = dx0 + (i - r * nfx) * 4, fy = fy0 + r * 16;
        const uvghip_scu_t &c = J.scu[(size_t)(fy >> 2) * J.scu_stride + (x >> 2)];
        if ((c.luma_edges & 2) && (c.chroma_edges & 2)) chroma_segment<PX>(bu, bv, TPC, J.scu, J.scu_stride, x >> 1, fy >> 1, true, cfg, false);
      }
    }
    __syncthreads();
    store_region<PX>(tu, TPC, cx0, cy0, (PX *)P.dbk_u, P.dbk_stride_c, dx0 >> 1, dy0 >> 1, dx1 >> 1, dy1 >> 1);
    store_region<PX>(tv, TPC, cx0, cy0, (PX *)P.dbk_v, P.dbk_stride_c, dx0 >> 1, dy0 >> 1, dx1 >> 1, dy1 >> 1);
    if (P.sao_type) {
      stats_block<PX>(F, tu, TPC, cx0, cy0, bx0 >> 1, by0 >> 1, bw >> 1, bh >> 1, (const PX *)J.src_u, J.src_stride_c, F->E[1], F->B[1]);
      stats_block<PX>(F, tv, TPC, cx0, cy0, bx0 >> 1, by0 >> 1, bw >> 1, bh >> 1, (const PX *)J.src_v, J.src_stride_c, F->E[2], F->B[2]);
    }
    __syncthreads();
  }
  // ---- the decision ----
  const int32_t *left_f = cx ? &J.sao_done[k - 1] : nullptr, *top_f = cy ? &J.sao_done[k - wc] : nullptr;
  if (P.sao_type) {
    if (threadIdx.x < 2) {
      const int g = threadIdx.x;
      const int32_t *E[2] = {g ? F->E[1] : F->E[0], F->E[2]}, *B[2] = {g ? F->B[1] : F->B[0], F->B[2]};
      const int omax = (1 << ((px_traits<PX>::depth < 10 ? px_traits<PX>::depth : 10) - 5)) - 1;              // SAO_ABS_OFFSET_MAX (global.h:295)
      saod::sao_candidates_one(E, B, g ? 2 : 1, omax, F->cands[g]);
    }
  }
  if (threadIdx.x == 0) {
    if (left_f) wait_set(left_f);
    if (top_f) wait_set(top_f);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (P.sao_type && threadIdx.x == 0) {
    saod::models2 m;
    const int from = cx > 0 ? k - 1 : (cy > 0 ? (cy - 1) * wc : -1);
    if (from < 0) saod::sao_models_init(m, P.slice_type, P.qp);
    else {
      const uint16_t *mi = P.sao_models + (size_t)from * 6;
      m.s0[0] = mi[0]; m.s1[0] = mi[1]; m.rate[0] = (uint8_t)mi[2]; m.s0[1] = mi[3]; m.s1[1] = mi[4]; m.rate[1] = (uint8_t)mi[5];
    }
    const saod::sao_info *all = reinterpret_cast<const saod::sao_info *>(P.sao_info);
    const saod::sao_info *top_l = cy ? &all[2 * (k - wc)] : nullptr, *left_l = cx ? &all[2 * (k - 1)] : nullptr;
    saod::sao_decide_one(m, P.lambda, P.sao_type, F->cands[0], F->cands[1], F->E[0], F->B[0], F->E[1], F->B[1], F->E[2], F->B[2], F->L, F->C, top_l, top_l ? top_l + 1 : nullptr,
                         left_l, left_l ? left_l + 1 : nullptr);
    saod::sao_info *mine = reinterpret_cast<saod::sao_info *>(P.sao_info) + 2 * (size_t)k;
    mine[0] = F->L; mine[1] = F->C;
    uint16_t *mo = P.sao_models + (size_t)k * 6;
    mo[0] = m.s0[0]; mo[1] = m.s1[0]; mo[2] = m.rate[0]; mo[3] = m.s0[1]; mo[4] = m.s1[1]; mo[5] = m.rate[1];
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&J.sao_done[k], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // ---- the output picture ----
  {
    const int delay = P.sao_type ? 10 : 8;
    const int fx0 = cx ? 64 * cx - delay : 0, fy0 = cy ? 64 * cy - delay : 0, fx1 = last_col ? W : 64 * cx + 64 - delay, fy1 = last_row ? H : 64 * cy + 64 - delay;
    if (P.sao_type) {
      sao_region<PX>((const PX *)P.dbk_y, P.dbk_stride, (PX *)P.out_y, P.out_stride, W, H, fx0, fy0, fx1, fy1, 6, wc, P.sao_info, 0);
      sao_region<PX>((const PX *)P.dbk_u, P.dbk_stride_c, (PX *)P.out_u, P.out_stride_c, W >> 1, H >> 1, fx0 >> 1, fy0 >> 1, fx1 >> 1, fy1 >> 1, 5, wc, P.sao_info, 1);
      sao_region<PX>((const PX *)P.dbk_v, P.dbk_stride_c, (PX *)P.out_v, P.out_stride_c, W >> 1, H >> 1, fx0 >> 1, fy0 >> 1, fx1 >> 1, fy1 >> 1, 5, wc, P.sao_info, 2);
    } else {
      for (int c = 0; c < 3; ++c) {
        const int sh = c != 0, x0 = fx0 >> sh, y0 = fy0 >> sh, w = (fx1 >> sh) - x0, n = w * ((fy1 >> sh) - y0);
        const PX *s = (const PX *)(c == 0 ? P.dbk_y : c == 1 ? P.dbk_u : P.dbk_v);
        PX *d = (PX *)(c == 0 ? P.out_y : c == 1 ? P.out_u : P.out_v);
        const int ss = c ? P.dbk_stride_c : P.dbk_stride, ds = c ? P.out_stride_c : P.out_stride;
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const int r = i / w, x = x0 + (i - r * w); d[(size_t)(y0 + r) * ds + x] = s[(size_t)(y0 + r) * ss + x]; }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (cx) wait_set(&J.final_done[k - 1]);
    if (cy) wait_set(&J.final_done[k - wc]);
    if (cy && cx + 1 < wc) wait_set(&J.final_done[k - wc + 1]);
    __hip_atomic_store(&J.final_done[k], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}

}  // namespace ctuf
