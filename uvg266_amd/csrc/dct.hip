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
#include "vvc_tables.h"

typedef short v2s __attribute__((ext_vector_type(2)));

enum { TR_DCT2 = 0, TR_DCT8 = 1, TR_DST7 = 2 };   // src/uvg266.h:235-237

struct tr_pass {
  int R, C, K;          // lines, outputs per line, taps
  int rmax, cmax, kmax; // lines processed / outputs kept / taps summed (rest -> 0)
  int shift;
};
struct tr_params {
  int w, h, inverse;
  int type_hor, type_ver;
  tr_pass p1, p2;
};

__device__ __forceinline__ const int16_t *tr_matrix_dev(int type, int n)
{
  if (type == TR_DCT2) return n == 4 ? VVC_DCT2_4 : n == 8 ? VVC_DCT2_8 : n == 16 ? VVC_DCT2_16 : VVC_DCT2_32;
  if (type == TR_DCT8) return n == 4 ? VVC_DCT8_4 : n == 8 ? VVC_DCT8_8 : n == 16 ? VVC_DCT8_16 : VVC_DCT8_32;
  return n == 4 ? VVC_DST7_4 : n == 8 ? VVC_DST7_8 : n == 16 ? VVC_DST7_16 : VVC_DST7_32;
}

// LDS pitches: K+2 int16 per row keeps rows 4-byte aligned and walks the
// banks (row stride = K/2+1 dwords, odd) so lanes reading different rows at
// the same k do not collide.
__device__ __forceinline__ int pitch_of(int k) { return k + 2; }

// One 1-D pass over all blocks held by the workgroup.
//   acc(r,c) = sum_{k<kmax} A[b][r][k] * B[c][k]
// r_fast: consecutive lanes take consecutive r (B row broadcast) else consecutive c.
// dst index = b*dst_blk + r*sr + c*sc.
template <bool INVERSE, typename DST>
__device__ __forceinline__ void run_pass(const tr_pass &p, const int16_t *A, int a_blk, const int16_t *B,
                                         DST *dst, int dst_blk, int sr, int sc, bool r_fast, int nblk_here)
{
  const int pa = pitch_of(p.K), pb = pitch_of(p.K);
  const int per_blk = p.R * p.C, total = per_blk * nblk_here;
  const int add = p.shift > 0 ? 1 << (p.shift - 1) : 0;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int b = e / per_blk, rem = e - b * per_blk;
    int r, c;
    if (r_fast) { c = rem / p.R; r = rem - c * p.R; } else { r = rem / p.C; c = rem - r * p.C; }
    int v = 0;
    if (r < p.rmax && c < p.cmax) {
      const int *a2 = reinterpret_cast<const int *>(A + b * a_blk + r * pa);
      const int *b2 = reinterpret_cast<const int *>(B + c * pb);
      int acc = 0;
#pragma unroll 4
      for (int k = 0; k < p.kmax / 2; ++k)
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a2[k]), __builtin_bit_cast(v2s, b2[k]), acc, false);
      v = (acc + add) >> p.shift;
      if (INVERSE) v = clampi(v, -32768, 32767);   // inverse clips (dct-generic.c:438)
    }
    dst[b * dst_blk + r * sr + c * sc] = (DST)(int16_t)v;   // forward truncates (dct-generic.c:411)
  }
}

__global__ void __launch_bounds__(256)
transform_kernel(tr_params P, const int16_t *__restrict__ in, int16_t *__restrict__ out, int n, int bpg)
{
  // A: input (K-contiguous per line), T: hand-off between the passes, M1/M2: kernel matrices
  __shared__ __attribute__((aligned(16))) int16_t sA[1024 + 2 * 256 + 64];
  __shared__ __attribute__((aligned(16))) int16_t sT[1024 + 2 * 256 + 64];
  __shared__ __attribute__((aligned(16))) int16_t sM1[32 * 34];
  __shared__ __attribute__((aligned(16))) int16_t sM2[32 * 34];

  const int w = P.w, h = P.h, wh = w * h;
  const int blk0 = blockIdx.x * bpg;
  const int here = min(bpg, n - blk0);
  if (here <= 0) return;
  const int16_t *gin = in + (size_t)blk0 * wh;
  int16_t *gout = out + (size_t)blk0 * wh;

  const tr_pass &p1 = P.p1, &p2 = P.p2;
  const int pa1 = pitch_of(p1.K), a1_blk = p1.R * pa1;
  const int pa2 = pitch_of(p2.K), a2_blk = p2.R * pa2;

  // ---- stage input and matrices ------------------------------------------
  if (!P.inverse) {
    // lines = rows of the block (R = h, K = w)
    for (int e = threadIdx.x; e < here * wh; e += blockDim.x) {
      const int b = e / wh, rem = e - b * wh, y = rem / w, x = rem - y * w;
      sA[b * a1_blk + y * pa1 + x] = gin[e];
    }
    const int16_t *Th = tr_matrix_dev(P.type_hor, w), *Tv = tr_matrix_dev(P.type_ver, h);
    for (int e = threadIdx.x; e < w * w; e += blockDim.x) sM1[(e / w) * pitch_of(w) + (e % w)] = Th[e];
    for (int e = threadIdx.x; e < h * h; e += blockDim.x) sM2[(e / h) * pitch_of(h) + (e % h)] = Tv[e];
  } else {
    // first pass is vertical: lines = columns i of the block (R = w, K = h), A[i][k] = in[k*w + i]
    for (int e = threadIdx.x; e < here * wh; e += blockDim.x) {
      const int b = e / wh, rem = e - b * wh, k = rem / w, i = rem - k * w;
      sA[b * a1_blk + i * pa1 + k] = gin[e];
    }
    const int16_t *Th = tr_matrix_dev(P.type_hor, w), *Tv = tr_matrix_dev(P.type_ver, h);
    // B[c][k] = T[k][c]
    for (int e = threadIdx.x; e < h * h; e += blockDim.x) sM1[(e % h) * pitch_of(h) + (e / h)] = Tv[e];
    for (int e = threadIdx.x; e < w * w; e += blockDim.x) sM2[(e % w) * pitch_of(w) + (e / w)] = Th[e];
  }
  __syncthreads();

  if (!P.inverse) {
    // pass 1 (horizontal): r = row y, c = freq j -> T[j][y]  (lines of pass 2 = j, K = h)
    run_pass<false>(p1, sA, a1_blk, sM1, sT, a2_blk, 1, pa2, true, here);
    __syncthreads();
    // pass 2 (vertical): r = hor freq i, c = ver freq j -> out[j*w + i]
    run_pass<false>(p2, sT, a2_blk, sM2, gout, wh, 1, w, true, here);
  } else {
    // pass 1 (vertical): r = column i, c = spatial row j -> T[j][i]  (lines of pass 2 = j, K = w)
    run_pass<true>(p1, sA, a1_blk, sM1, sT, a2_blk, 1, pa2, true, here);
    __syncthreads();
    // pass 2 (horizontal): r = spatial row, c = spatial column -> out[r*w + c]
    run_pass<true>(p2, sT, a2_blk, sM2, gout, wh, w, 1, false, here);
  }
}

// Which 1-D kernels honour skip_line2 in the reference (dct-generic.c):
//   forward: DST7/DCT8 with n >= 8 zero rows >= cutoff (:1651,:1768,:2014,:2139,:2264,:2335);
//            every DCT2 kernel and the 4-point DST7/DCT8 ignore it.
//   inverse: only the 8-point DST7/DCT8 stop their sums at cutoff (:2280,:2351).
static bool fwd_cut(int type, int n) { return type != TR_DCT2 && n >= 8; }
static bool inv_cut(int type, int n) { return type != TR_DCT2 && n == 8; }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static tr_params make_params(int bitdepth, int inverse, int th, int tv, int w, int h, int sw, int sh)
{
  tr_params P;
  P.w = w; P.h = h; P.inverse = inverse; P.type_hor = th; P.type_ver = tv;
  if (!inverse) {
    P.p1 = tr_pass{h, w, w, h, fwd_cut(th, w) ? w - sw : w, w, ilog2(w) - 1 + bitdepth - 8};
    P.p2 = tr_pass{w, h, h, w - sw, fwd_cut(tv, h) ? h - sh : h, h, ilog2(h) + 6};
  } else {
    P.p1 = tr_pass{w, h, h, w - sw, h, inv_cut(tv, h) ? h - sh : h, 7};
    P.p2 = tr_pass{h, w, w, h, w, inv_cut(th, w) ? w - sw : w, 20 - bitdepth};
  }
  return P;
}

static bool valid_dim(int v) { return v == 4 || v == 8 || v == 16 || v == 32; }

extern "C" int uvghip_transform_batch(int bitdepth, int inverse, int type_hor, int type_ver, int width, int height,
                                      int skip_width, int skip_height, const int16_t *in, int16_t *out, int n,
                                      void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!valid_dim(width) || !valid_dim(height) || type_hor < 0 || type_hor > 2 || type_ver < 0 || type_ver > 2 ||
      skip_width < 0 || skip_width >= width || skip_height < 0 || skip_height >= height ||
      (skip_width & 3) || (skip_height & 3))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  const tr_params P = make_params(bitdepth, inverse != 0, type_hor, type_ver, width, height, skip_width, skip_height);
  const int bpg = 1024 / (width * height);
  const int grid = (n + bpg - 1) / bpg;
  transform_kernel<<<grid, 256, 0, uvghip_stream(stream)>>>(P, in, out, n, bpg);
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

// Not registered: fast_forward/inverse_dst_4x4 (dead upstream: their only
// callers are commented out, strategies-dct.c:106-110,140-144).
extern "C" int uvg_strategy_register_dct_hip(void *opaque, uint8_t bitdepth)
{
  if (!uvghip_ready() && uvghip_init(0) != 0) return 0;
  int ok = 1;
#define REG(type, fn) ok &= uvghip_do_register(opaque, type, (void *)(fn))
  REG("dct_4x4", (&dct_nxn_hip<4, 0>));     REG("dct_8x8", (&dct_nxn_hip<8, 0>));
  REG("dct_16x16", (&dct_nxn_hip<16, 0>));  REG("dct_32x32", (&dct_nxn_hip<32, 0>));
  REG("idct_4x4", (&dct_nxn_hip<4, 1>));    REG("idct_8x8", (&dct_nxn_hip<8, 1>));
  REG("idct_16x16", (&dct_nxn_hip<16, 1>)); REG("idct_32x32", (&dct_nxn_hip<32, 1>));
  REG("mts_dct", (&mts_hip<0>));
  REG("mts_idct", (&mts_hip<1>));
#undef REG
  return ok;
}
