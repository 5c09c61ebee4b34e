// Matrix-based intra prediction (MIP) on gfx950.  Bit-exact with
//   mip_predict_generic, uvg_mip_boundary_downsampling_1D, uvg_mip_reduced_pred, uvg_mip_pred_upsampling_1D
//   (src/strategies/generic/intra-generic.c:441-727).
// One wave per block: reference rows in LDS, the <= 8 reduced boundary values and the 16 / 64 reduced predictions
// by lanes, then every output sample from the closed form of the reference's two 1-D interpolation passes
// (horizontal first, each pass rounding on its own):
//   pass(before, behind, pos) = ((before << lg) - pos * before + pos * behind + (1 << (lg - 1))) >> lg,  pos = 1..factor.
// The weights are the normative tables of H.266 8.4.5.2.4 (vvc_mip_tables.h, generated -- tools/gen_mip_tables.py).
#include "uvghip_common.h"
#include "percall.h"
#include "vvc_mip_tables.h"

namespace {

__device__ __forceinline__ int ilog2i(int v) { return 31 - __clz(v); }
__device__ __forceinline__ int mip_interp(int before, int behind, int pos, int lg)
{
  return ((before << lg) - pos * before + pos * behind + (1 << (lg - 1))) >> lg;
}

// top / left: LDS rows with the corner at index 0 (samples from index 1), w / h entries used.
template <typename PX>
__device__ __forceinline__ void mip_block(const uint16_t *top, const uint16_t *left, int w, int h, int mode, bool transpose,
                                          int *sIn, int *sRed, PX *out, int out_fill, int lane)
{
  const int size_id = (w == 4 && h == 4) ? 0 : ((w == 4 || h == 4 || (w == 8 && h == 8)) ? 1 : 2);
  const int rb = size_id == 0 ? 2 : 4, rp = size_id < 2 ? 4 : 8, in_size = 2 * rb;
  const int ups_h = w / rp, ups_v = h / rp;
  // reduced boundary: entries 0..rb-1 from the top row, rb..2rb-1 from the left column (transposed: the other way round)
  if (lane < in_size) {
    const bool from_top = (lane < rb) != transpose;
    const int idx = lane < rb ? lane : lane - rb;
    const uint16_t *src = from_top ? top : left;
    const int len = from_top ? w : h, f = len / rb;
    int v;
    if (f > 1) { int s = 0; for (int k = 0; k < f; ++k) s += src[1 + idx * f + k]; const int lg = ilog2i(f); v = (s + (1 << (lg - 1))) >> lg; }
    else v = src[1 + idx];
    sIn[lane] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int in_off = sIn[0];
  const int half = 1 << (px_traits<PX>::depth - 1);
  int in[8], sum = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int v = 0;
    if (i < in_size) v = i == 0 ? (size_id < 2 ? half - in_off : 0) : sIn[i] - in_off;
    in[i] = v; sum += v;
  }
  const uint8_t *M = size_id == 0 ? VVC_MIP0 + (size_t)mode * 16 * 4 : (size_id == 1 ? VVC_MIP1 + (size_t)mode * 16 * 8 : VVC_MIP2 + (size_t)mode * 64 * 8);
  if (lane < rp * rp) {
    int acc = 32 - 32 * sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < in_size) acc += in[i] * M[i * rp * rp + lane];     // input-major table
    const int v = clampi((acc >> 6) + in_off, 0, px_traits<PX>::maxv);
    const int y = lane / rp, x = lane - y * rp;
    sRed[transpose ? x * rp + y : lane] = v;             // transposed: result (y, x) is output (x, y)
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int lgh = ups_h > 1 ? ilog2i(ups_h) : 0, lgv = ups_v > 1 ? ilog2i(ups_v) : 0;
  // horizontally up-sampled value of reduced row yr at column x
  auto H = [&](int yr, int x) -> int {
    if (ups_h == 1) return sRed[yr * rp + x];
    const int u = x >> lgh, pos = (x & (ups_h - 1)) + 1;
    const int before = u == 0 ? (int)left[1 + yr * ups_v + ups_v - 1] : sRed[yr * rp + u - 1];
    return mip_interp(before, sRed[yr * rp + u], pos, lgh);
  };
  for (int p = lane; p < w * h; p += 64) {
    const int y = p / w, x = p - y * w;
    int v;
    if (ups_v == 1) v = H(y, x);
    else {
      const int vr = y >> lgv, pos = (y & (ups_v - 1)) + 1;
      const int before = vr == 0 ? (int)top[1 + x] : H(vr - 1, x);
      v = mip_interp(before, H(vr, x), pos, lgv);
    }
    out[p] = (PX)v;
  }
  for (int p = w * h + lane; p < out_fill; p += 64) out[p] = 0;   // the reference's dst is 32*32 with zeros behind the block
}

// Reference rows of one block from a plane, as uvg_intra_build_reference leaves them for MRL 0 (see intra.hip).
template <typename PX>
__device__ __forceinline__ void mip_build_rows(const PX *rec, int stride, int x, int y, int w, int h, int avail_top, int avail_left,
                                               uint16_t *top, uint16_t *left, int lane)
{
  const int dc = 1 << (px_traits<PX>::depth - 1);
  if (avail_left < 1) avail_left = 1;
  if (avail_top < 1) avail_top = 1;
  for (int i = lane; i < 64; i += 64) {
    int lv = dc, tv = dc;
    if (i < h) { if (x > 0) lv = rec[(size_t)(y + min(i, avail_left - 1)) * stride + x - 1]; else if (y > 0) lv = rec[(size_t)(y - 1) * stride + x]; }
    if (i < w) { if (y > 0) tv = rec[(size_t)(y - 1) * stride + x + min(i, avail_top - 1)]; else if (x > 0) tv = rec[(size_t)y * stride + x - 1]; }
    left[1 + i] = (uint16_t)lv; top[1 + i] = (uint16_t)tv;
  }
  if (lane == 0) top[0] = left[0] = (uint16_t)dc;        // the corner is not read by MIP
}

template <typename PX>
__global__ void __launch_bounds__(256)
mip_pred_kernel(const PX *__restrict__ rec, int stride, int w, int h, const uvghip_intra_blk_t *__restrict__ blks, int n,
                const uint8_t *__restrict__ mode_transp, PX *__restrict__ out)
{
  __shared__ uint16_t sTop[4][72], sLeft[4][72];
  __shared__ int sIn[4][8], sRed[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  const uvghip_intra_blk_t b = blks[i];
  mip_build_rows<PX>(rec, stride, b.x, b.y, w, h, b.avail_top, b.avail_left, sTop[wave], sLeft[wave], lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int mt = mode_transp[i];
  mip_block<PX>(sTop[wave], sLeft[wave], w, h, mt & 0x7f, (mt & 0x80) != 0, sIn[wave], sRed[wave], out + (size_t)i * w * h, 0, lane);
}

template <typename PX>
__global__ void __launch_bounds__(64)
mip_from_rows_kernel(const PX *__restrict__ ref_top, const PX *__restrict__ ref_left, int w, int h, int mode, int transp,
                     PX *__restrict__ dst, int dst_fill)
{
  __shared__ uint16_t sTop[72], sLeft[72];
  __shared__ int sIn[8], sRed[64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 65; i += 64) { sTop[i] = i <= w ? ref_top[i] : 0; sLeft[i] = i <= h ? ref_left[i] : 0; }
  __syncthreads();
  mip_block<PX>(sTop, sLeft, w, h, mode, transp != 0, sIn, sRed, dst, dst_fill, lane);
}

bool mip_shape_ok(int w, int h, int mode)
{
  auto ok = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
  if (!ok(w) || !ok(h)) return false;
  const int size_id = (w == 4 && h == 4) ? 0 : ((w == 4 || h == 4 || (w == 8 && h == 8)) ? 1 : 2);
  return mode >= 0 && mode < (size_id == 0 ? 16 : size_id == 1 ? 8 : 6);
}

// mip_pred_func (strategies-intra.h:79-86): (const uvg_intra_references *refs, uint16_t w, uint16_t h, uvg_pixel *dst,
// int mip_mode, bool mip_transp).  uvg_intra_references begins with uvg_intra_ref ref = { left[358], top[358] } (intra.h:46-57).
template <typename PX>
void mip_predict_hip(const void *refs, const uint16_t w, const uint16_t h, PX *dst, const int mip_mode, const bool mip_transp)
{
  if (!mip_shape_ok(w, h, mip_mode) || w > 32 || h > 32) {
    fprintf(stderr, "uvg266hip: mip_predict %dx%d mode %d is outside the supported range\n", w, h, mip_mode);
    abort();
  }
  const PX *left = static_cast<const PX *>(refs), *top = left + 358;
  const size_t rb = 65 * sizeof(PX), db = 32 * 32 * sizeof(PX);
  percall_ctx *c = percall_get(2 * rb + db + 1024);
  const size_t ot = c->take(rb), ol = c->take(rb), od = c->take(db);
  memcpy(c->hp<PX>(ot), top, rb);
  memcpy(c->hp<PX>(ol), left, rb);
  c->upload(0, c->used);
  mip_from_rows_kernel<PX><<<1, 64, 0, c->stream>>>(c->dp<PX>(ot), c->dp<PX>(ol), w, h, mip_mode, mip_transp ? 1 : 0, c->dp<PX>(od), 32 * 32);
  if (hipGetLastError() != hipSuccess) c->fail("mip launch");
  c->download(od, db);
  c->sync();
  memcpy(dst, c->hp<PX>(od), db);        // the reference writes all 32*32 entries of dst (zeros behind the block)
}

}  // namespace

extern "C" int uvghip_mip_pred_batch(int bitdepth, const void *rec, int rec_stride, int width, int height,
                                     const uvghip_intra_blk_t *blks, int n, const uint8_t *mode_transp, void *preds_out,
                                     void *stream)
{
  UVGHIP_REQUIRE_READY();
  if ((bitdepth != 8 && bitdepth != 10) || !mip_shape_ok(width, height, 0)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
  const int grid = (n + 3) / 4;
  if (bitdepth == 8) mip_pred_kernel<uint8_t><<<grid, 256, 0, st>>>((const uint8_t *)rec, rec_stride, width, height, blks, n, mode_transp, (uint8_t *)preds_out);
  else mip_pred_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t *)rec, rec_stride, width, height, blks, n, mode_transp, (uint16_t *)preds_out);
  UVGHIP_CHECK_LAUNCH();
}

// called by uvg_strategy_register_intra_hip (intra.hip)
int uvghip_register_mip(void *opaque, uint8_t bitdepth)
{
  return bitdepth == 8 ? uvghip_do_register(opaque, "mip_predict", (void *)&mip_predict_hip<uint8_t>)
                       : uvghip_do_register(opaque, "mip_predict", (void *)&mip_predict_hip<uint16_t>);
}
