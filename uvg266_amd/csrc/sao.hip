// "sao" strategy group on gfx950: sample-adaptive-offset statistics and
// reconstruction over rectangles (CTUs) of device-resident planes.
// Bit-exact with
//   calc_sao_edge_dir       src/strategies/generic/sao-generic.c:51-81
//   sao_edge_ddistortion    src/strategies/generic/sao_shared_generics.h:53-88
//   sao_band_ddistortion    src/strategies/generic/sao_shared_generics.h:90-127
//   sao_reconstruct_color   src/strategies/generic/sao-generic.c:84-124
//   calc_sao_bands / uvg_calc_sao_offset_array / uvg_sao_reconstruct   src/sao.c:180-201,268-285,302-361
//
// Statistics kernel: one workgroup per rectangle.  Each thread walks its
// pixels once, classifying them for all four edge classes from a 3x3
// neighbourhood, and adds (count, difference) to its own column of an LDS
// table of 52 packed accumulators (20 edge class/category pairs + 32 bands):
// no contended atomics, no select chains; the columns are summed at the end
// (details at the kernel).  HBM traffic: orig + rec read once, 104 integers
// written per rectangle.
#include <climits>
#include "uvghip_common.h"
#include "percall.h"
#include "ref_abi.h"

__device__ __forceinline__ int sgn3(int v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int eo_cat(int a, int b, int c)
{
  // {1,2,0,3,4}[2 + sign(c-a) + sign(c-b)]  (sao_shared_generics.h:42-50)
  const int idx = 2 + sgn3(c - a) + sgn3(c - b);
  return (0x43021 >> (4 * idx)) & 7;
}
__device__ static const int8_t kEoOfs[4][4] = {{-1, 0, 1, 0}, {0, -1, 0, 1}, {-1, -1, 1, 1}, {1, -1, -1, 1}};  // ax,ay,bx,by (sao.h:71-76)

// Statistics: one workgroup per rectangle, one thread per 4-sample segment at a time.  Every thread owns a
// private column of an LDS table [52 entries][256 threads]: entries 0..19 = (edge class, category),
// 20..51 = band; each holds (count << 16 | sum of (diff + BIAS)).  A sample is five conflict-free
// ds_add_u32 on the thread's own column (bank = thread id) -- no select chains, no contended atomics.
// The 16-bit fields hold a chunk of at most 8192 samples per workgroup (32 per thread; a 64x64 CTU is one
// chunk); larger rectangles are walked in row chunks whose column sums accumulate in 32-bit registers.
template <typename PX>
__global__ void __launch_bounds__(256)
sao_stats_kernel(const PX *__restrict__ orig, int ostride, const PX *__restrict__ rec, int rstride,
                 const uvghip_rect_t *__restrict__ rects, int32_t *__restrict__ edge_out, int32_t *__restrict__ band_out)
{
  constexpr int BIAS = 1 << px_traits<PX>::depth;
  constexpr int bshift = px_traits<PX>::depth - 5;
  __shared__ uint32_t sT[52][256];
  const uvghip_rect_t R = rects[blockIdx.x];
  const int tid = threadIdx.x;
  const int segs_per_row = (R.w + 3) >> 2;
  const int rows_chunk = max(1, 2048 / segs_per_row);          // <= 8192 samples per chunk
  int tot_sum = 0, tot_cnt = 0;                                // running totals of entry tid / 4 (quarter tid & 3)
  for (int row0 = 0; row0 < R.h; row0 += rows_chunk) {
#pragma unroll
  for (int e = 0; e < 52; ++e) sT[e][tid] = 0;
  const int nseg = segs_per_row * min(rows_chunk, R.h - row0);
  for (int sg = tid; sg < nseg; sg += 256) {
    const int yc = sg / segs_per_row, x0 = (sg - yc * segs_per_row) * 4, y = row0 + yc;
    const PX *rp = rec + (size_t)(R.y + y) * rstride + R.x + x0;
    const PX *op = orig + (size_t)(R.y + y) * ostride + R.x + x0;
    const int nx = min(4, R.w - x0);
    const bool rows_in = y >= 1 && y < R.h - 1;                 // interior rows only (sao-generic.c:67-68)
    int c[4], o[4], u[6], m[6], d[6];                           // m = this row at x0-1 .. x0+4
    if (nx == 4) { load4(rp, c); load4(op, o); }
    else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { c[i] = i < nx ? rp[i] : 0; o[i] = i < nx ? op[i] : 0; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) m[1 + i] = c[i];
    const bool has_l = x0 >= 1, has_r = x0 + 4 < R.w;           // the samples left/right of the segment, inside the rectangle
    m[0] = has_l ? rp[-1] : 0; m[5] = has_r ? rp[4] : 0;
    if (rows_in) {
      if (nx == 4) { int t[4]; load4(rp - rstride, t); u[1] = t[0]; u[2] = t[1]; u[3] = t[2]; u[4] = t[3];
                     load4(rp + rstride, t); d[1] = t[0]; d[2] = t[1]; d[3] = t[2]; d[4] = t[3]; }
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { u[1 + i] = i < nx ? rp[i - rstride] : 0; d[1 + i] = i < nx ? rp[i + rstride] : 0; }
      }
      u[0] = has_l ? rp[-rstride - 1] : 0; u[5] = has_r ? rp[-rstride + 4] : 0;
      d[0] = has_l ? rp[rstride - 1] : 0;  d[5] = has_r ? rp[rstride + 4] : 0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nx) {
        const int x = x0 + i;
        const uint32_t add = (1u << 16) + (uint32_t)(o[i] - c[i] + BIAS);
        atomicAdd(&sT[20 + (c[i] >> bshift)][tid], add);
        if (rows_in && x >= 1 && x < R.w - 1) {
          const int cc = c[i];
          atomicAdd(&sT[0 + eo_cat(m[i], m[i + 2], cc)][tid], add);         // class 0: left / right
          atomicAdd(&sT[5 + eo_cat(u[i + 1], d[i + 1], cc)][tid], add);     // class 1: up / down
          atomicAdd(&sT[10 + eo_cat(u[i], d[i + 2], cc)][tid], add);        // class 2: up-left / down-right
          atomicAdd(&sT[15 + eo_cat(u[i + 2], d[i], cc)][tid], add);        // class 3: up-right / down-left
        }
      }
    }
  }
  __syncthreads();
  // column sums: 52 entries x 256 threads; thread t reduces entry t / 4 over a quarter of the columns
  if (tid < 208) {
    const int e = tid >> 2, q = tid & 3;
    int sum = 0, cnt = 0;
    for (int k = 0; k < 64; ++k) {
      const uint32_t v = sT[e][q * 64 + ((k + tid) & 63)];     // staggered start: the wave's lanes hit different banks
      sum += (int)(v & 0xffffu); cnt += (int)(v >> 16);
    }
    tot_sum += sum - cnt * BIAS; tot_cnt += cnt;
  }
  __syncthreads();
  }   // chunks
  {
    const int e = tid >> 2, q = tid & 3;
    int s = tot_sum, cnt = tot_cnt;                              // the four quarter partials meet through the wave
    s += __shfl_xor(s, 1, 64); cnt += __shfl_xor(cnt, 1, 64);
    s += __shfl_xor(s, 2, 64); cnt += __shfl_xor(cnt, 2, 64);
    if (q == 0 && e < 52) {
      if (e < 20) {
        const int cl = e / 5, k = e - cl * 5;
        edge_out[(size_t)blockIdx.x * 40 + cl * 10 + k] = s;
        edge_out[(size_t)blockIdx.x * 40 + cl * 10 + 5 + k] = cnt;
      } else {
        band_out[(size_t)blockIdx.x * 64 + (e - 20)] = s;
        band_out[(size_t)blockIdx.x * 64 + 32 + (e - 20)] = cnt;
      }
    }
  }
}

extern "C" int uvghip_sao_stats_batch(int bitdepth, const void *orig, int orig_stride, const void *rec, int rec_stride,
                                      const uvghip_rect_t *rects, int n, int32_t *edge_stats, int32_t *band_stats,
                                      void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) sao_stats_kernel<uint8_t><<<n, 256, 0, st>>>((const uint8_t *)orig, orig_stride, (const uint8_t *)rec, rec_stride, rects, edge_stats, band_stats);
  else sao_stats_kernel<uint16_t><<<n, 256, 0, st>>>((const uint16_t *)orig, orig_stride, (const uint16_t *)rec, rec_stride, rects, edge_stats, band_stats);
  UVGHIP_CHECK_LAUNCH();
}

// Apply one parameter set per rectangle.  Edge classes skip the picture's outermost
// rows/columns (sao.c:321-348); band offsets are applied through the value test of
// uvg_calc_sao_offset_array (sao.c:180-201) instead of a 2^depth LUT.  Samples SAO leaves alone (type 0
// rectangles, the skipped border rows/columns) are copied from rec, so `out` is completely defined -- the
// reference gets the same by filtering a copy of the picture in place.
// One thread per 4-sample segment: the segment and its two neighbour segments (class-dependent) are three
// unaligned 4-sample loads; offsets come from a 5-entry LDS table.
template <typename PX>
__global__ void __launch_bounds__(256)
sao_apply_kernel(const PX *__restrict__ rec, int rstride, PX *__restrict__ out, int ostride, int pic_w, int pic_h,
                 const uvghip_rect_t *__restrict__ rects, const uvghip_sao_param_t *__restrict__ params)
{
  __shared__ int sOff[8];
  const uvghip_rect_t R = rects[blockIdx.x];
  const uvghip_sao_param_t P = params[blockIdx.x];
  // straight from global memory: indexing the register copy P dynamically would make the compiler promote P to an
  // LDS alloca whose addressing reads the AQL dispatch packet (host memory) -- measured +13 us per launch
  if (threadIdx.x < 5) sOff[threadIdx.x] = params[blockIdx.x].offsets[threadIdx.x];
  __syncthreads();
  constexpr int maxv = px_traits<PX>::maxv;
  constexpr int bshift = px_traits<PX>::depth - 5;
  // [ix0, ix1) x [iy0, iy1): the part of the rectangle SAO modifies, relative to the rectangle
  int ix0 = 0, iy0 = 0, ix1 = R.w, iy1 = R.h;
  int ax = 0, ay = 0, bx = 0, by = 0;
  if (P.type == 2) {
    ax = kEoOfs[P.eo_class][0]; ay = kEoOfs[P.eo_class][1]; bx = kEoOfs[P.eo_class][2]; by = kEoOfs[P.eo_class][3];
    if (R.x + R.w + ax > pic_w || R.x + R.w + bx > pic_w) ix1 -= 1;
    if (R.x + ax < 0 || R.x + bx < 0) ix0 += 1;
    // (the rectangle's row inside its own picture: `rec` may be a stack of pictures of pic_h rows each, one under the other)
    const int ry = R.y % pic_h;
    if (ry + R.h + ay > pic_h || ry + R.h + by > pic_h) iy1 -= 1;
    if (ry + ay < 0 || ry + by < 0) iy0 += 1;
  } else if (P.type == 0) ix1 = 0;
  if (R.w <= 0 || R.h <= 0) return;
  const int segs_per_row = (R.w + 3) >> 2, nseg = segs_per_row * R.h;
  const ptrdiff_t oa = (ptrdiff_t)ay * rstride + ax, ob = (ptrdiff_t)by * rstride + bx;
  for (int sg = threadIdx.x; sg < nseg; sg += 256) {
    const int y = sg / segs_per_row, x = (sg - y * segs_per_row) * 4;
    const PX *rp = rec + (size_t)(R.y + y) * rstride + R.x + x;
    PX *q = out + (size_t)(R.y + y) * ostride + R.x + x;
    const int nx = min(4, R.w - x);
    const bool row_in = y >= iy0 && y < iy1;
    const bool whole = nx == 4 && row_in && x >= ix0 && x + 4 <= ix1;     // every sample of the segment is filtered
    int c[4], a[4], b[4], v[4];
    if (whole) {
      load4(rp, c);
      if (P.type == 2) { load4(rp + oa, a); load4(rp + ob, b); }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool in = i < nx;
        const bool filt = in && row_in && x + i >= ix0 && x + i < ix1;
        c[i] = in ? rp[i] : 0;
        a[i] = filt && P.type == 2 ? rp[oa + i] : 0;
        b[i] = filt && P.type == 2 ? rp[ob + i] : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool filt = whole || (row_in && x + i >= ix0 && x + i < ix1);
      if (!filt) v[i] = c[i];
      else if (P.type == 1) {
        const int band = (c[i] >> bshift) - P.band_position;
        v[i] = (band >= 0 && band <= 3) ? clampi(c[i] + sOff[band + 1], 0, maxv) : c[i];
      } else v[i] = clampi(c[i] + sOff[eo_cat(a[i], b[i], c[i])], 0, maxv);
    }
    if (nx == 4) {
      if constexpr (sizeof(PX) == 1) *reinterpret_cast<u32_unaligned *>(q) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
      else { *reinterpret_cast<u32_unaligned *>(q) = (uint32_t)v[0] | ((uint32_t)v[1] << 16); *reinterpret_cast<u32_unaligned *>(q + 2) = (uint32_t)v[2] | ((uint32_t)v[3] << 16); }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) if (i < nx) q[i] = (PX)v[i];
    }
  }
}

extern "C" int uvghip_sao_apply_batch(int bitdepth, const void *rec, int rec_stride, void *out, int out_stride,
                                      int pic_w, int pic_h, const uvghip_rect_t *rects,
                                      const uvghip_sao_param_t *params, int n, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) sao_apply_kernel<uint8_t><<<n, 256, 0, st>>>((const uint8_t *)rec, rec_stride, (uint8_t *)out, out_stride, pic_w, pic_h, rects, params);
  else sao_apply_kernel<uint16_t><<<n, 256, 0, st>>>((const uint16_t *)rec, rec_stride, (uint16_t *)out, out_stride, pic_w, pic_h, rects, params);
  UVGHIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------- edge offsets from statistics ----
// sao.c:380-439 without the entropy-coder rate (passed in by the host, or zero).  One thread per rectangle.
__global__ void __launch_bounds__(256)
sao_edge_offsets_kernel(const int32_t *__restrict__ edge_stats, const int32_t *__restrict__ rate_cost, int n,
                        uvghip_sao_param_t *__restrict__ params, int32_t *__restrict__ ddist)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t *st = edge_stats + (size_t)i * 40;          // [class][sum|cnt][cat]
  int best_dd = INT_MAX, best_class = 0;
  int best_off[5] = {0, 0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) {
    int off[5] = {0, 0, 0, 0, 0};
    int dd = 0;
#pragma unroll
    for (int cat = 1; cat <= 4; ++cat) {
      const int sum = st[c * 10 + cat], cnt = st[c * 10 + 5 + cat];
      int o = 0;
      if (cnt != 0) o = clampi((sum + (cnt >> 1)) / cnt, -7, 7);   // C division: truncates toward zero
      if (cat <= 2 && o < 0) o = 0;
      if (cat >= 3 && o > 0) o = 0;
      off[cat] = o;
      dd += cnt * o * o - 2 * o * sum;
    }
    if (rate_cost) dd += rate_cost[(size_t)i * 4 + c];
    if (dd < best_dd) {
      best_dd = dd; best_class = c;
#pragma unroll
      for (int k = 0; k < 5; ++k) best_off[k] = off[k];
    }
  }
  uvghip_sao_param_t P;
  P.type = 2; P.eo_class = best_class; P.band_position = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) P.offsets[k] = best_off[k];
  params[i] = P;
  if (ddist) ddist[i] = best_dd;
}

extern "C" int uvghip_sao_edge_offsets_batch(const int32_t *edge_stats, const int32_t *rate_cost, int n,
                                             uvghip_sao_param_t *params_out, int32_t *ddist_out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (n <= 0) return 0;
  sao_edge_offsets_kernel<<<(n + 255) / 256, 256, 0, uvghip_stream(stream)>>>(edge_stats, rate_cost, n, params_out, ddist_out);
  UVGHIP_CHECK_LAUNCH();
}

// =================================================== drop-in strategy layer ====
// strategies-sao.h:49-66.  encoder_control_t* / encoder_state_t* arguments are
// only read for the bit depth in the reference (sao.c:183, sao_shared_generics.h:100),
// which the registrar fixes, so they are ignored here.
namespace {

template <typename PX>
struct staged_pair { percall_ctx *c; size_t oo, orr, orect; };

template <typename PX>
staged_pair<PX> stage_packed(const PX *orig, const PX *rec, int bw, int bh, size_t extra)
{
  const size_t nb = (size_t)bw * bh * sizeof(PX);
  percall_ctx *c = percall_get(2 * nb + extra + 2048);
  staged_pair<PX> s{c, c->take(nb), c->take(nb), c->take(sizeof(uvghip_rect_t))};
  memcpy(c->hp<PX>(s.oo), orig, nb);
  memcpy(c->hp<PX>(s.orr), rec, nb);
  *c->hp<uvghip_rect_t>(s.orect) = uvghip_rect_t{0, 0, bw, bh};
  return s;
}

template <typename PX>
void stats_packed(const PX *orig, const PX *rec, int bw, int bh, int32_t (&edge)[40], int32_t (&band)[64])
{
  staged_pair<PX> s = stage_packed<PX>(orig, rec, bw, bh, 512);
  percall_ctx *c = s.c;
  const size_t oe = c->take(160), ob = c->take(256);
  c->upload(0, oe);
  c->must(uvghip_sao_stats_batch(px_traits<PX>::depth, c->dp<PX>(s.oo), bw, c->dp<PX>(s.orr), bw, c->dp<uvghip_rect_t>(s.orect), 1,
                                 c->dp<int32_t>(oe), c->dp<int32_t>(ob), c->stream), "sao stats");
  c->download(oe, 160 + 256 + 96);
  c->sync();
  memcpy(edge, c->hp<int32_t>(oe), 160);
  memcpy(band, c->hp<int32_t>(ob), 256);
}

template <typename PX>
void calc_sao_edge_dir_hip(const PX *orig_data, const PX *rec_data, int eo_class, int block_width, int block_height,
                           int cat_sum_cnt[2][5])
{
  int32_t edge[40], band[64];
  stats_packed<PX>(orig_data, rec_data, block_width, block_height, edge, band);
  for (int s = 0; s < 2; ++s)
    for (int k = 0; k < 5; ++k) cat_sum_cnt[s][k] += edge[eo_class * 10 + s * 5 + k];   // accumulates, like the reference
}

// delta-distortion = sum_cat cnt*o^2 - 2*o*sum: an exact function of the statistics
template <typename PX>
int sao_edge_ddistortion_hip(const PX *orig_data, const PX *rec_data, int block_width, int block_height, int eo_class,
                             int offsets[5])
{
  int32_t edge[40], band[64];
  stats_packed<PX>(orig_data, rec_data, block_width, block_height, edge, band);
  int sum = 0;
  for (int k = 0; k < 5; ++k) sum += edge[eo_class * 10 + 5 + k] * offsets[k] * offsets[k] - 2 * offsets[k] * edge[eo_class * 10 + k];
  return sum;
}
template <typename PX>
int sao_band_ddistortion_hip(const void * /*state*/, const PX *orig_data, const PX *rec_data, int block_width,
                             int block_height, int band_pos, const int sao_bands[4])
{
  int32_t edge[40], band[64];
  stats_packed<PX>(orig_data, rec_data, block_width, block_height, edge, band);
  int sum = 0;
  for (int k = 0; k < 4; ++k) {
    const int b = band_pos + k;
    if (b < 0 || b > 31) continue;
    sum += band[32 + b] * sao_bands[k] * sao_bands[k] - 2 * sao_bands[k] * band[b];
  }
  return sum;
}

// sao_reconstruct_color: rec_data may be read one sample outside the block for edge
// classes (the caller's buffer has a 1-sample border, sao.c:302-361), so a bordered copy is staged.
template <typename PX>
void sao_reconstruct_color_hip(const void * /*encoder*/, const PX *rec_data, PX *new_rec_data, const ref_sao_info *sao,
                               int stride, int new_stride, int block_width, int block_height, int color_i)
{
  if (sao->type == 0 || block_width <= 0 || block_height <= 0) return;
  const int bw = block_width + 2, bh = block_height + 2;
  const size_t nb = (size_t)bw * bh * sizeof(PX);
  percall_ctx *c = percall_get(2 * nb + 2048);
  const size_t oi = c->take(nb), orect = c->take(sizeof(uvghip_rect_t)), op = c->take(sizeof(uvghip_sao_param_t)), oo = c->take(nb);
  PX *h = c->hp<PX>(oi);
  const bool edge = sao->type == 2;
  for (int y = 0; y < bh; ++y)
    for (int x = 0; x < bw; ++x) {
      const bool in_x = x >= 1 && x <= block_width, in_y = y >= 1 && y <= block_height;
      // touch only the border samples the edge class really reads (class 0: left/right, 1: up/down, 2/3: corners too)
      const bool need = (in_x && in_y) || (edge && ((in_y && sao->eo_class != 1) || (in_x && sao->eo_class != 0) ||
                                                    (!in_x && !in_y && sao->eo_class >= 2)));
      h[y * bw + x] = need ? rec_data[(ptrdiff_t)(y - 1) * stride + (x - 1)] : (PX)0;
    }
  *c->hp<uvghip_rect_t>(orect) = uvghip_rect_t{1, 1, block_width, block_height};
  uvghip_sao_param_t P;
  const int v = color_i == 2;
  P.type = sao->type; P.eo_class = sao->eo_class; P.band_position = sao->band_position[v ? 1 : 0];
  for (int k = 0; k < 5; ++k) P.offsets[k] = sao->offsets[k + (v ? 5 : 0)];
  *c->hp<uvghip_sao_param_t>(op) = P;
  c->upload(0, oo);
  // picture size chosen so that no border row/column is dropped: the caller already did that (sao.c:321-348)
  c->must(uvghip_sao_apply_batch(px_traits<PX>::depth, c->dp<PX>(oi), bw, c->dp<PX>(oo), bw, bw + 2, bh + 2,
                                 c->dp<uvghip_rect_t>(orect), c->dp<uvghip_sao_param_t>(op), 1, c->stream), "sao apply");
  c->download(oo, nb);
  c->sync();
  const PX *o = c->hp<PX>(oo);
  for (int y = 0; y < block_height; ++y)
    memcpy(new_rec_data + (size_t)y * new_stride, o + (size_t)(y + 1) * bw + 1, (size_t)block_width * sizeof(PX));
}

template <typename PX>
int register_sao(void *opaque)
{
  int ok = 1;
  ok &= uvghip_do_register(opaque, "calc_sao_edge_dir", (void *)&calc_sao_edge_dir_hip<PX>);
  ok &= uvghip_do_register(opaque, "sao_edge_ddistortion", (void *)&sao_edge_ddistortion_hip<PX>);
  ok &= uvghip_do_register(opaque, "sao_band_ddistortion", (void *)&sao_band_ddistortion_hip<PX>);
  ok &= uvghip_do_register(opaque, "sao_reconstruct_color", (void *)&sao_reconstruct_color_hip<PX>);
  return ok;
}

}  // namespace

extern "C" int uvg_strategy_register_sao_hip(void *opaque, uint8_t bitdepth)
{
  if (!uvghip_ready() && uvghip_init(0) != 0) return 0;
  return bitdepth == 8 ? register_sao<uint8_t>(opaque) : register_sao<uint16_t>(opaque);
}
