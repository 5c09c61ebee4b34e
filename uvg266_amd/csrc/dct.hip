// "dct" strategy group on gfx950: DCT-2 / DST-7 / DCT-8, 4..32 points,
// square and non-square, forward and inverse, with the reference's zero-out
// rules.  Bit-exact with src/strategies/generic/dct-generic.c:
//   dct_NxN / idct_NxN            :396-750  (fwd truncates to int16, inv clips)
//   mts_dct_generic / mts_idct    :2560-2678 (two 1-D passes, transposed hand-off)
//   per-kernel skip_line2 rules   :1030-2370 (see pass_params() below)
//
// Design: the reference's butterflies are exact integer factorisations of
// the N-point matrix products (no intermediate rounding), so each 1-D pass
// is computed as a K-contiguous int16 dot product with v_dot2c_i32_i16.
// One workgroup (256 threads) holds 1024 coefficients in LDS: one 32x32
// block, four 16x16, ... sixty-four 4x4.  HBM traffic per block is the
// algorithmic minimum: w*h int16 read once, w*h int16 written once; the
// intermediate and both kernels matrices never leave LDS.
#include "uvghip_common.h"
#include "percall.h"
#include "ref_abi.h"

#include "transform_dev.h"

__global__ void __launch_bounds__(256)
transform_kernel(tr_params P, int inverse, const int16_t *__restrict__ in, int16_t *__restrict__ out, int n, int bpg)
{
  // A: input (K-contiguous per line), T: hand-off between the passes, M1/M2: kernel matrices
  __shared__ __attribute__((aligned(16))) int16_t sA[TR_LINEBUF_ELEMS];
  __shared__ __attribute__((aligned(16))) int16_t sT[TR_LINEBUF_ELEMS];
  __shared__ __attribute__((aligned(16))) int16_t sM1[TR_MATRIX_ELEMS];
  __shared__ __attribute__((aligned(16))) int16_t sM2[TR_MATRIX_ELEMS];

  const int w = P.w, h = P.h, wh = w * h;
  const int blk0 = blockIdx.x * bpg;
  const int here = min(bpg, n - blk0);
  if (here <= 0) return;
  const int16_t *gin = in + (size_t)blk0 * wh;
  int16_t *gout = out + (size_t)blk0 * wh;

  const tr_pass p1 = inverse ? P.i1 : P.f1, p2 = inverse ? P.i2 : P.f2;
  const int pa1 = tr_pitch(p1.K), a1_blk = p1.R * pa1;
  const int pa2 = tr_pitch(p2.K), a2_blk = p2.R * pa2;

  // ---- stage input and matrices ------------------------------------------
  if (!inverse) {
    // lines = rows of the block (R = h, K = w)
    for (int e = threadIdx.x; e < here * wh; e += blockDim.x) {
      const int b = e / wh, rem = e - b * wh, y = rem / w, x = rem - y * w;
      sA[b * a1_blk + y * pa1 + x] = gin[e];
    }
    tr_stage_matrix(sM1, P.type_hor, w, false);
    tr_stage_matrix(sM2, P.type_ver, h, false);
  } else {
    // first pass is vertical: lines = columns i of the block (R = w, K = h), A[i][k] = in[k*w + i]
    for (int e = threadIdx.x; e < here * wh; e += blockDim.x) {
      const int b = e / wh, rem = e - b * wh, k = rem / w, i = rem - k * w;
      sA[b * a1_blk + i * pa1 + k] = gin[e];
    }
    tr_stage_matrix(sM1, P.type_ver, h, true);
    tr_stage_matrix(sM2, P.type_hor, w, true);
  }
  __syncthreads();

  // pass 1 hands lines to pass 2 K-contiguously: T[b][c][r]
  auto to_lds = [&](int b, int r, int c, int v) { sT[b * a2_blk + c * pa2 + r] = (int16_t)v; };
  if (!inverse) {
    tr_run_pass<false>(p1, sA, a1_blk, sM1, true, here, to_lds);     // horizontal: r = row, c = hor freq
    __syncthreads();
    // vertical: r = hor freq i, c = ver freq j -> out[j*w + i]
    tr_run_pass<false>(p2, sT, a2_blk, sM2, true, here, [&](int b, int r, int c, int v) { gout[b * wh + c * w + r] = (int16_t)v; });
  } else {
    tr_run_pass<true>(p1, sA, a1_blk, sM1, true, here, to_lds);      // vertical: r = column, c = spatial row
    __syncthreads();
    // horizontal: r = spatial row, c = spatial column -> out[r*w + c]
    tr_run_pass<true>(p2, sT, a2_blk, sM2, false, here, [&](int b, int r, int c, int v) { gout[b * wh + r * w + c] = (int16_t)v; });
  }
}

// ---- thin blocks: a dimension of 1 or 2 (ISP sub-partitions, 2xN chroma from multi-type trees) ----
// mts_dct_generic / mts_idct_generic handle them with the 2-point DCT-2 (dct_table[DCT2][0], dct-generic.c:1030-1090) and,
// for a dimension of 1, a single 1-D pass with its own shift (:2608-2612, :2669-2673).  Rare and tiny: one 64-thread
// workgroup per block, plain loops, the intermediate in LDS.
__device__ __forceinline__ int thin_coef(int type, int n, int row, int col)
{
  if (n == 2) return row == 1 && col == 1 ? -64 : 64;                 // uvg_g_DCT2P2 = {64, 64, 64, -64}
  return tr_matrix_dev(type, n)[row * n + col];
}

// forward 1-D over `line` rows of n samples: dst[j * line + i] (the reference's transposed hand-off); inverse: dst[i * n + j]
__device__ inline void thin_pass(bool inverse, int type, int n, const int16_t *src, int16_t *dst, int shift, int line, int skip_line,
                                 int skip_line2)
{
  const int add = shift > 0 ? 1 << (shift - 1) : 0;
  const int reduced = line - skip_line;
  const int cut_f = (!inverse && type != TR_DCT2 && n >= 8) ? n - skip_line2 : n;      // which kernels honour skip_line2: transform_dev.h
  const int kmax_i = (inverse && type != TR_DCT2 && n == 8) ? n - skip_line2 : n;
  for (int e = threadIdx.x; e < n * line; e += blockDim.x) {
    int v = 0;
    if (!inverse) {
      const int j = e / line, i = e - j * line;
      if (j < cut_f && i < reduced) {
        int acc = 0;
        for (int k = 0; k < n; ++k) acc += thin_coef(type, n, j, k) * src[i * n + k];
        v = (int)(int16_t)((acc + add) >> shift);
      }
      dst[j * line + i] = (int16_t)v;
    } else {
      const int i = e / n, j = e - i * n;
      if (i < reduced) {
        int acc = 0;
        for (int k = 0; k < kmax_i; ++k) acc += src[k * line + i] * thin_coef(type, n, k, j);
        v = clampi((acc + add) >> shift, -32768, 32767);
      }
      dst[i * n + j] = (int16_t)v;
    }
  }
}

__global__ void __launch_bounds__(64)
transform_thin_kernel(int bitdepth, int inverse, int type_hor, int type_ver, int w, int h, int skip_w, int skip_h,
                      const int16_t *__restrict__ in, int16_t *__restrict__ out)
{
  __shared__ int16_t sIn[64], sTmp[64], sOut[64];
  const int wh = w * h;
  const int16_t *gin = in + (size_t)blockIdx.x * wh;
  int16_t *gout = out + (size_t)blockIdx.x * wh;
  for (int e = threadIdx.x; e < wh; e += blockDim.x) sIn[e] = gin[e];
  __syncthreads();
  const int lw = 31 - __clz(w), lh = 31 - __clz(h);
  if (!inverse) {
    const int s1 = lw - 1 + bitdepth - 8, s2 = lh - 1 + 7;
    if (h == 1) thin_pass(false, type_hor, w, sIn, sOut, s1, 1, 0, skip_w);
    else if (w == 1) thin_pass(false, type_ver, h, sIn, sOut, lh - 1 + 1 + bitdepth + 6 - 15, 1, 0, skip_h);
    else {
      thin_pass(false, type_hor, w, sIn, sTmp, s1, h, 0, skip_w);
      __syncthreads();
      thin_pass(false, type_ver, h, sTmp, sOut, s2, w, skip_w, skip_h);
    }
  } else {
    const int s1 = 7, s2 = 20 - bitdepth;
    if (h == 1) thin_pass(true, type_hor, w, sIn, sOut, s2 + 1, 1, 0, skip_w);
    else if (w == 1) thin_pass(true, type_ver, h, sIn, sOut, s2 + 1, 1, 0, skip_h);
    else {
      thin_pass(true, type_ver, h, sIn, sTmp, s1, w, skip_w, skip_h);
      __syncthreads();
      thin_pass(true, type_hor, w, sTmp, sOut, s2, h, 0, skip_w);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < wh; e += blockDim.x) gout[e] = sOut[e];
}

int uvghip_launch_tr_lane(const tr_params &P, bool inverse, const int16_t *in, int16_t *out, int n, hipStream_t st);   // quant.hip
int uvghip_launch_tr_wave(const tr_params &P, bool inverse, const int16_t *in, int16_t *out, int n, hipStream_t st);

extern "C" int uvghip_transform_batch(int bitdepth, int inverse, int type_hor, int type_ver, int width, int height,
                                      int skip_width, int skip_height, const int16_t *in, int16_t *out, int n,
                                      void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (bitdepth != 8 && bitdepth != 10) return uvghip_set_error(hipErrorInvalidValue, __func__);
  {
    // thin blocks (a dimension of 1 or 2, the other 1..32): ISP / 2xN chroma.  Only DCT-2 exists for 1- and 2-point lines.
    auto pow2_le32 = [](int v) { return v >= 1 && v <= 32 && !(v & (v - 1)); };
    if (pow2_le32(width) && pow2_le32(height) && (width <= 2 || height <= 2)) {
      if ((width <= 2 && type_hor != TR_DCT2) || (height <= 2 && type_ver != TR_DCT2) || type_hor < 0 || type_hor > 2 || type_ver < 0 ||
          type_ver > 2 || skip_width < 0 || skip_width >= width || skip_height < 0 || skip_height >= height || width * height > 64)
        return uvghip_set_error(hipErrorInvalidValue, __func__);
      if (n <= 0) return 0;
      transform_thin_kernel<<<n, 64, 0, uvghip_stream(stream)>>>(bitdepth, inverse != 0, type_hor, type_ver, width, height, skip_width,
                                                                 skip_height, in, out);
      UVGHIP_CHECK_LAUNCH();
    }
  }
  if (!tr_valid_dim(width) || !tr_valid_dim(height) || type_hor < 0 || type_hor > 2 || type_ver < 0 || type_ver > 2 ||
      skip_width < 0 || skip_width >= width || skip_height < 0 || skip_height >= height ||
      (skip_width & 3) || (skip_height & 3))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  const tr_params P = tr_make_params(bitdepth, type_hor, type_ver, width, height, skip_width, skip_height);
  // the TU kernels' passes (quant.hip) read and write whole dwords / quads: 16-byte aligned, non-aliasing buffers only
  if (width == height && skip_width == 0 && skip_height == 0 && in != out && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
    if (width <= 8) return uvghip_launch_tr_lane(P, inverse != 0, in, out, n, uvghip_stream(stream));
    return uvghip_launch_tr_wave(P, inverse != 0, in, out, n, uvghip_stream(stream));
  }
  const int bpg = 1024 / (width * height);
  const int grid = (n + bpg - 1) / bpg;
  transform_kernel<<<grid, 256, 0, uvghip_stream(stream)>>>(P, inverse != 0, in, out, n, bpg);
  UVGHIP_CHECK_LAUNCH();
}

// uvg_get_tr_type (dct-generic.c:2501-2557) + skip rules (:2582-2600) on plain arguments.
extern "C" int uvghip_mts_select(int width, int height, int color, int cu_type, int isp_mode, int lfnst_idx,
                                 int cr_lfnst_idx, int tr_idx, int mts_type, int *type_hor, int *type_ver,
                                 int *skip_width, int *skip_height)
{
  int hor = TR_DCT2, ver = TR_DCT2;
  const bool intra = cu_type == REF_CU_INTRA, inter = cu_type == REF_CU_INTER;
  if (color == 0) {
    const bool explicit_mts = mts_type == 3 || (intra ? mts_type == 1 : (mts_type == 2 && inter));
    const bool implicit_mts = intra && (mts_type == 4 || mts_type == 2);
    const bool is_isp = intra && isp_mode;
    if (!(is_isp && lfnst_idx)) {
      if (implicit_mts || (is_isp && explicit_mts)) {
        if (width >= 4 && width <= 16) hor = TR_DST7;
        if (height >= 4 && height <= 16) ver = TR_DST7;
      } else if (explicit_mts && tr_idx > 1) {
        static const int subset[4][2] = {{TR_DST7, TR_DST7}, {TR_DCT8, TR_DST7}, {TR_DST7, TR_DCT8}, {TR_DCT8, TR_DCT8}};
        hor = subset[tr_idx - 2][0]; ver = subset[tr_idx - 2][1];
      }
    }
  }
  int sw = (hor != TR_DCT2 && width == 32) ? 16 : (width > 32 ? width - 32 : 0);
  int sh = (ver != TR_DCT2 && height == 32) ? 16 : (height > 32 ? height - 32 : 0);
  const bool lf = (lfnst_idx && color == 0) || (cr_lfnst_idx && color != 0);
  if (lf) {
    if ((width == 4 && height > 4) || (width > 4 && height == 4)) { sw = width - 4; sh = height - 4; }
    else if (width >= 8 && height >= 8) { sw = width - 8; sh = height - 8; }
  }
  // The square DCT-2 fast path (dct_NxN) has no zero-out at all (:2567-2571).
  if (hor == TR_DCT2 && ver == TR_DCT2 && !lfnst_idx && !cr_lfnst_idx && width == height) sw = sh = 0;
  *type_hor = hor; *type_ver = ver; *skip_width = sw; *skip_height = sh;
  return 0;
}

// =================================================== drop-in strategy layer ====
// dct_func (strategies-dct.h:44): (int8_t bitdepth, const int16_t *input, int16_t *output)
// mts dct  (strategies-dct.h:46-63)
namespace {

void percall_transform(int bitdepth, int inverse, int th, int tv, int w, int h, int sw, int sh,
                       const int16_t *input, int16_t *output)
{
  const size_t bytes = (size_t)w * h * 2;
  percall_ctx *c = percall_get(2 * bytes + 1024);
  const size_t oi = c->take(bytes), oo = c->take(bytes);
  memcpy(c->hp<int16_t>(oi), input, bytes);
  c->upload(oi, bytes);
  c->must(uvghip_transform_batch(bitdepth, inverse, th, tv, w, h, sw, sh, c->dp<int16_t>(oi), c->dp<int16_t>(oo), 1,
                                 c->stream), "transform");
  c->download(oo, bytes);
  c->sync();
  memcpy(output, c->hp<int16_t>(oo), bytes);
}

template <int N, int INV> void dct_nxn_hip(int8_t bitdepth, const int16_t *input, int16_t *output)
{
  percall_transform(bitdepth, INV, TR_DCT2, TR_DCT2, N, N, 0, 0, input, output);
}

template <int INV>
void mts_hip(const int8_t bitdepth, const int /*color_t*/ color, const ref_cu_info *tu, const int8_t width,
             const int8_t height, const int16_t *input, int16_t *output, const int8_t mts_type)
{
  int th, tv, sw, sh;
  const int isp = tu->type == REF_CU_INTRA ? tu->intra.isp_mode : 0;
  uvghip_mts_select(width, height, color, tu->type, isp, tu->lfnst_idx, tu->cr_lfnst_idx, tu->tr_idx, mts_type,
                    &th, &tv, &sw, &sh);
  percall_transform(bitdepth, INV, th, tv, width, height, sw, sh, input, output);
}

}  // namespace

// fast_forward/inverse_dst_4x4 (dct-generic.c:359-393,752-770; dead upstream: their only callers are commented
// out, strategies-dct.c:106-110,140-144) are the HEVC 4-point DST-VII, whose integer matrix and stage shifts are
// exactly the VVC DST-7 4x4 transform pair -- registered for table completeness as that kernel.
template <int INV> void dst_4x4_hip(int8_t bitdepth, const int16_t *input, int16_t *output)
{
  percall_transform(bitdepth, INV, TR_DST7, TR_DST7, 4, 4, 0, 0, input, output);
}
extern "C" int uvg_strategy_register_dct_hip(void *opaque, uint8_t bitdepth)
{
  if (!uvghip_ready() && uvghip_init(0) != 0) return 0;
  int ok = 1;
#define REG(type, fn) ok &= uvghip_do_register(opaque, type, (void *)(fn))
  REG("dct_4x4", (&dct_nxn_hip<4, 0>));     REG("dct_8x8", (&dct_nxn_hip<8, 0>));
  REG("dct_16x16", (&dct_nxn_hip<16, 0>));  REG("dct_32x32", (&dct_nxn_hip<32, 0>));
  REG("idct_4x4", (&dct_nxn_hip<4, 1>));    REG("idct_8x8", (&dct_nxn_hip<8, 1>));
  REG("idct_16x16", (&dct_nxn_hip<16, 1>)); REG("idct_32x32", (&dct_nxn_hip<32, 1>));
  REG("fast_forward_dst_4x4", (&dst_4x4_hip<0>));
  REG("fast_inverse_dst_4x4", (&dst_4x4_hip<1>));
  REG("mts_dct", (&mts_hip<0>));
  REG("mts_idct", (&mts_hip<1>));
#undef REG
  return ok;
}
