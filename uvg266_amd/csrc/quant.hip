// "quant" strategy group on gfx950: quant / dequant / coeff_abs_sum /
// fast_coeff_cost and the fused TU round trip (quantize_residual).
// Bit-exact with src/strategies/generic/quant-generic.c:
//   uvg_quant_generic    :51-121  (scaling list off, lfnst 0; sign hiding :123-233 not implemented --
//                                  signhide=0 in every preset of the north-star configs)
//   uvg_dequant_generic  :618-669 (no scaling list / dep-quant)
//   uvg_quantize_residual_generic :460-612, plain-quant branch (:532-536)
//   coeff_abs_sum :671, fast_coeff_cost :688
// The reference reads qp, bit depth, slice type and flags from
// encoder_state_t; here they are plain arguments (the host-side shim that
// extracts them is shown in INTEGRATION.md).
//
// Fused TU kernel data flow (one workgroup = 1024 coefficients = 1..64 TUs):
//   HBM: orig + pred pixels (read once) -> residual in LDS -> 2 forward
//   passes (dot2) -> quant in registers -> levels to HBM (coeff_out) and
//   dequantised straight into the inverse line buffer in LDS -> 2 inverse
//   passes -> + pred, clip -> recon pixels to HBM.  Nothing intermediate
//   touches HBM; algorithmic bytes per TU = w*h*(2*px + 2 + px).
#include "uvghip_common.h"
#include "percall.h"
#include "transform_dev.h"

struct quant_params {
  int scale, add; int q_bits;      // level = (|c| * scale + add) >> q_bits   (int64 product)
  int iscale, iadd, ishift;        // coef  = clip16((q * iscale + iadd) >> ishift)
  int dq_on_load;                  // TU_INV kernels: the input holds LEVELS, dequantise them as they are loaded
};

static quant_params make_quant_params(int bitdepth, int width, int height, int qp_scaled, int transform_skip,
                                      int slice_is_intra)
{
  static const int16_t qs[2][6] = {{26214, 23302, 20560, 18396, 16384, 14564}, {18396, 16384, 14564, 13107, 11651, 10280}};
  static const int16_t iqs[2][6] = {{40, 45, 51, 57, 64, 72}, {57, 64, 72, 80, 90, 102}};
  const int lw = tr_ilog2(width), lh = tr_ilog2(height);
  if (transform_skip && qp_scaled < 4 + 6 * 2) qp_scaled = 4 + 6 * 2;     // MIN_QP_PRIME_TS (global.h:147)
  const int sqrt2 = !transform_skip && ((lw + lh) & 1);
  quant_params q;
  const int tshift_q = 15 - bitdepth - ((lw + lh) >> 1) - sqrt2;         // quant-generic.c:74
  q.q_bits = 14 + qp_scaled / 6 + (transform_skip ? 0 : tshift_q);
  q.add = (slice_is_intra ? 171 : 85) << (q.q_bits - 9);
  q.scale = qs[sqrt2][qp_scaled % 6];
  const int tshift_d = 15 - bitdepth - ((lw + lh) >> 1);                  // quant-generic.c:629
  q.ishift = 20 - 14 - (transform_skip ? 0 : tshift_d - sqrt2);
  q.iscale = iqs[sqrt2][qp_scaled % 6] << (qp_scaled / 6);
  q.iadd = 1 << (q.ishift - 1);
  q.dq_on_load = 0;
  return q;
}

__device__ __forceinline__ int quant_one(int c, const quant_params &q)
{
  const long long a = c < 0 ? -(long long)c : (long long)c;
  int level = (int)((a * q.scale + q.add) >> q.q_bits);
  if (c < 0) level = -level;
  return clampi(level, -32768, 32767);
}
// Same value with 32-bit arithmetic, valid while |c| <= 32768 and q_bits <= 31 (every TU of at least 4x4):
// |c| * scale <= 32768 * 26214 and add <= 171 << 22, so the sum stays below 2^32.
__device__ __forceinline__ int quant_one32(int c, const quant_params &q)
{
  const uint32_t a = (uint32_t)abs(c);
  int level = (int)((__umul24(a, (uint32_t)q.scale) + (uint32_t)q.add) >> q.q_bits);
  if (c < 0) level = -level;
  return clampi(level, -32768, 32767);
}
__device__ __forceinline__ int dequant_one(int l, const quant_params &q)
{
  return clampi((l * q.iscale + q.iadd) >> q.ishift, -32768, 32767);
}

__global__ void __launch_bounds__(256)
quant_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out, size_t total, quant_params q, int inverse)
{
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= total) return;
  if (i0 + 4 <= total) {
    const short4 v = *reinterpret_cast<const short4 *>(in + i0);
    short4 r;
    if (inverse) { r.x = dequant_one(v.x, q); r.y = dequant_one(v.y, q); r.z = dequant_one(v.z, q); r.w = dequant_one(v.w, q); }
    else { r.x = quant_one(v.x, q); r.y = quant_one(v.y, q); r.z = quant_one(v.z, q); r.w = quant_one(v.w, q); }
    *reinterpret_cast<short4 *>(out + i0) = r;
  } else {
    for (size_t i = i0; i < total; ++i) out[i] = inverse ? dequant_one(in[i], q) : quant_one(in[i], q);
  }
}

static int check_qargs(int bitdepth, int width, int height, int qp_scaled)
{
  // at least 8 coefficients: below that the dequantiser's shift (20 - 14 - transform_shift) is not positive and its rounding
  // term 1 << (shift - 1) is undefined (the reference has the same latent problem, quant-generic.c:655)
  if ((bitdepth != 8 && bitdepth != 10) || width < 1 || height < 1 || width > 64 || height > 64 || width * height < 8 ||
      (width & (width - 1)) || (height & (height - 1)) || qp_scaled < 0 || qp_scaled > 63 + 12)
    return uvghip_set_error(hipErrorInvalidValue, "quant arguments");
  return 0;
}

extern "C" int uvghip_quant_batch(int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n,
                                  int qp_scaled, int transform_skip, int slice_is_intra, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (int rc = check_qargs(bitdepth, width, height, qp_scaled)) return rc;
  if (n <= 0) return 0;
  const quant_params q = make_quant_params(bitdepth, width, height, qp_scaled, transform_skip, slice_is_intra);
  const size_t total = (size_t)n * width * height;
  quant_kernel<<<(unsigned)((total / 4 + 255) / 256 + 1), 256, 0, uvghip_stream(stream)>>>(coef, q_coef, total, q, 0);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_dequant_batch(int bitdepth, const int16_t *q_coef, int16_t *coef, int width, int height, int n,
                                    int qp_scaled, int transform_skip, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (int rc = check_qargs(bitdepth, width, height, qp_scaled)) return rc;
  if (n <= 0) return 0;
  const quant_params q = make_quant_params(bitdepth, width, height, qp_scaled, transform_skip, 1);
  const size_t total = (size_t)n * width * height;
  quant_kernel<<<(unsigned)((total / 4 + 255) / 256 + 1), 256, 0, uvghip_stream(stream)>>>(q_coef, coef, total, q, 1);
  UVGHIP_CHECK_LAUNCH();
}

// uvg_quant with lfnst_idx != 0 quantises only the first 8 (4x4 / 8x8 TUs) or 16 coefficients of the scan and zeroes the
// rest (quant-generic.c:101-120).  Quantisation is element-wise, so masking the quantised block gives the same levels:
// the first 16 scan positions are the top-left 4x4 group in its up-right diagonal order.
__global__ void __launch_bounds__(256) lfnst_keep_kernel(int16_t *__restrict__ q, int width, int wh, size_t total, int keep)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int pos = (int)(i % wh), x = pos % width, y = pos / width;
  bool kept = false;
  if (x < 4 && y < 4) {
    // rank of (x, y) in the 4x4 up-right diagonal order: diagonals d = x + y, each walked from x = 0 (y = d) upwards
    const int d = x + y;
    const int before = d <= 3 ? d * (d + 1) / 2 : 16 - (7 - d) * (8 - d) / 2;
    const int in_diag = d <= 3 ? x : x - (d - 3);
    kept = before + in_diag < keep;
  }
  if (!kept) q[i] = 0;
}

extern "C" int uvghip_quant_lfnst_batch(int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n,
                                        int qp_scaled, int transform_skip, int slice_is_intra, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (int rc = check_qargs(bitdepth, width, height, qp_scaled)) return rc;
  if (width < 4 || height < 4) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  quant_params q = make_quant_params(bitdepth, width, height, qp_scaled, transform_skip, slice_is_intra);
  // this branch reads the scaling-list array even with scaling lists off (quant_coeff[n], :113), and the flat encoder lists are
  // built from uvg_g_quant_scales[0] without the sqrt(2) row ("TODO: the sqrt adjusted lists", scalinglist.c:415-417), while
  // q_bits keeps the block-size adjustment: reproduced as is
  {
    static const int16_t qs0[6] = {26214, 23302, 20560, 18396, 16384, 14564};
    int qp = qp_scaled;
    if (transform_skip && qp < 4 + 6 * 2) qp = 4 + 6 * 2;
    q.scale = qs0[qp % 6];
  }
  const size_t total = (size_t)n * width * height;
  hipStream_t st = uvghip_stream(stream);
  quant_kernel<<<(unsigned)((total / 4 + 255) / 256 + 1), 256, 0, st>>>(coef, q_coef, total, q, 0);
  const int keep = ((width == 4 && height == 4) || (width == 8 && height == 8)) ? 8 : 16;
  lfnst_keep_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(q_coef, width, width * height, total, keep);
  UVGHIP_CHECK_LAUNCH();
}

// ---- uvg_quant with sign-data hiding (cfg.signhide_enable; quant-generic.c:51-232) ----
// A workgroup of 64 threads takes 1024 / (w h) blocks = 1024 positions = 64 coefficient groups.  Phase 1: every position's level
// (the sqrt(2)-aware scale; the lfnst form quantises the first 8 / 16 scan positions with the flat scale) and delta_u (always
// the flat scaling-list scale, :133/:145) into LDS, the block's level sum.  Phase 2: one thread per coefficient group runs the
// hiding decision (:154-228) -- groups do not interact; "last_cg" (the first group from the end that holds a level, whose scan
// starts at its last non-zero position) is found with a per-block reduction.
__global__ void __launch_bounds__(64)
quant_signhide_kernel(const int16_t *__restrict__ coef, int16_t *__restrict__ q_coef, int n, int l2w, int l2h, quant_params q, int flat_scale,
                      int lfnst_idx)
{
  __shared__ int16_t sQ[1024];
  __shared__ int sDelta[1024];
  __shared__ uint16_t sScan[1024];               // scan position -> raster position inside a block
  __shared__ uint8_t sScanCg[64];
  __shared__ unsigned sSum[64], sLastCg[64];
  const int tid = threadIdx.x;
  const int l2wh = l2w + l2h, wh = 1 << l2wh, width = 1 << l2w, height = 1 << l2h;
  const int bpw = 1024 >> l2wh;                  // blocks per workgroup
  const int blk0 = blockIdx.x * bpw, here = min(bpw, n - blk0);
  const int l2cgw = l2w - 2, cgw = 1 << l2cgw, cgh = height >> 2, ncg = wh >> 4;
  if (tid == 0) {
    int i = 0, x = 0, y = 0;
    while (i < ncg) {
      while (y >= 0) { if (x < cgw && y < cgh) sScanCg[i++] = (uint8_t)(y * cgw + x); y--; x++; }
      y = x; x = 0;
    }
  }
  if (tid < bpw) { sSum[tid] = 0; sLastCg[tid] = 0; }
  __syncthreads();
  constexpr unsigned long long kDiag4 = 0xFBE7AD369C258140ull;
  for (int e = tid; e < wh; e += 64) {
    const int g = sScanCg[e >> 4], k = (int)((kDiag4 >> (4 * (e & 15))) & 15);
    sScan[e] = (uint16_t)(((((g >> l2cgw) << 2) + (k >> 2)) << l2w) + ((g & (cgw - 1)) << 2) + (k & 3));
  }
  __syncthreads();
  const int maxn = ((width == 4 && height == 4) || (width == 8 && height == 8)) ? 8 : 16;
  // ---- phase 1 ----
  for (int e = tid; e < here * wh; e += 64) {
    const int b = e >> l2wh, sp = e & (wh - 1);                   // walk in SCAN order so that the lfnst cut is a range test
    const int pos = sScan[sp];
    const int c = coef[(size_t)(blk0 + b) * wh + pos];
    const long long a = c < 0 ? -(long long)c : (long long)c;
    int level = 0, delta = 0;
    if (!lfnst_idx || sp < maxn) {
      const int lv_flat = (int)((a * flat_scale + q.add) >> q.q_bits);
      level = lfnst_idx ? lv_flat : (int)((a * q.scale + q.add) >> q.q_bits);
      delta = (int)((a * flat_scale - ((long long)lv_flat << q.q_bits)) >> (q.q_bits - 8));
      if (level) atomicAdd(&sSum[b], (unsigned)level);
    }
    sQ[b * wh + pos] = (int16_t)clampi(c < 0 ? -level : level, -32768, 32767);
    sDelta[b * wh + pos] = delta;
  }
  __syncthreads();
  // ---- phase 2: one coefficient group per thread ----
  const int gid = tid;                                            // group index inside the workgroup: block gid / ncg, group gid % ncg
  const int gb = gid >> (l2wh - 4), subset = gid & (ncg - 1);
  const bool act = gb < here;
  const int16_t *Q = sQ + gb * wh;
  int first_nz = 16, last_nz = -1;
  if (act) {
    for (int k = 15; k >= 0; k--) if (Q[sScan[subset * 16 + k]]) { last_nz = k; break; }
    for (int k = 0; k < 16; k++) if (Q[sScan[subset * 16 + k]]) { first_nz = k; break; }
    if (last_nz >= 0) atomicMax(&sLastCg[gb], (unsigned)subset + 1);
  }
  __syncthreads();
  if (act && sSum[gb] >= 2 && last_nz - first_nz >= 4) {
    const int subpos = subset * 16;
    int abssum = 0;
    for (int k = first_nz; k <= last_nz; k++) abssum += Q[sScan[subpos + k]];
    const int signbit = Q[sScan[subpos + first_nz]] > 0 ? 0 : 1;
    if (signbit != (abssum & 1)) {
      const bool is_last_cg = sLastCg[gb] == (unsigned)subset + 1;
      int min_cost = 0x7fffffff, cur_cost = 0x7fffffff, min_pos = -1, final_change = 0, cur_change = 0;
      const int16_t *C = coef + (size_t)(blk0 + gb) * wh;
      for (int k = is_last_cg ? last_nz : 15; k >= 0; k--) {
        const int b = sScan[subpos + k];
        const int qv = Q[b], du = sDelta[gb * wh + b];
        if (qv != 0) {
          if (du > 0) { cur_cost = -du; cur_change = 1; }
          else if (k == first_nz && abs(qv) == 1) cur_cost = 0x7fffffff;
          else { cur_cost = du; cur_change = -1; }
        } else if (k < first_nz && ((C[b] >= 0) ? 0 : 1) != signbit) cur_cost = 0x7fffffff;
        else { cur_cost = -du; cur_change = 1; }
        if (cur_cost < min_cost) { min_cost = cur_cost; final_change = cur_change; min_pos = b; }
      }
      if (min_pos >= 0) {
        const int qv = Q[min_pos];
        if (qv == 32767 || qv == -32768) final_change = -1;
        sQ[gb * wh + min_pos] = (int16_t)(C[min_pos] >= 0 ? qv + final_change : qv - final_change);
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < here * wh; e += 64) q_coef[(size_t)blk0 * wh + e] = sQ[e];
}

extern "C" int uvghip_quant_signhide_batch(int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n,
                                           int qp_scaled, int transform_skip, int slice_is_intra, int lfnst_idx, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (int rc = check_qargs(bitdepth, width, height, qp_scaled)) return rc;
  auto pow2 = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32; };
  if (!pow2(width) || !pow2(height) || !coef || !q_coef || lfnst_idx < 0 || lfnst_idx > 2) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  const quant_params q = make_quant_params(bitdepth, width, height, qp_scaled, transform_skip, slice_is_intra);
  static const int16_t qs0[6] = {26214, 23302, 20560, 18396, 16384, 14564};
  int qp = qp_scaled;
  if (transform_skip && qp < 4 + 6 * 2) qp = 4 + 6 * 2;
  const int l2w = tr_ilog2(width), l2h = tr_ilog2(height);
  const int bpw = 1024 / (width * height);
  quant_signhide_kernel<<<(n + bpw - 1) / bpw, 64, 0, uvghip_stream(stream)>>>(coef, q_coef, n, l2w, l2h, q, qs0[qp % 6], lfnst_idx);
  UVGHIP_CHECK_LAUNCH();
}

// ---- per-block coefficient sums ------------------------------------------------
// out[b] = sum |c| (mode 0, coeff_abs_sum) or (sum weights[min(|c|,3)] + 128) >> 8 (mode 1, fast_coeff_cost)
__global__ void __launch_bounds__(256)
coeff_cost_kernel(const int16_t *__restrict__ c, int len, int n, uint32_t *__restrict__ out, int lpb, int mode,
                  unsigned long long weights)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int bpw = 64 / lpb;
  const int blk = wave * bpw + lane / lpb;
  const int l = lane & (lpb - 1);
  const bool active = blk < n;
  int acc = 0;
  if (active) {
    const int16_t *p = c + (size_t)blk * len;
    for (int i = l; i < len; i += lpb) {
      int a = abs((int)p[i]);
      if (mode) { a = a > 3 ? 3 : a; acc += (int)((weights >> (16 * a)) & 0xffff); }
      else acc += a;
    }
  }
  acc = group_sum(acc, lpb);
  if (active && l == 0) out[blk] = mode ? ((uint32_t)acc + 128u) >> 8 : (uint32_t)acc;
}

static int launch_coeff_cost(const int16_t *c, int len, int n, uint32_t *out, int mode, unsigned long long w, hipStream_t st)
{
  if (n <= 0) return 0;
  int lpb = 1; while (lpb < len && lpb < 64) lpb <<= 1;
  const int bpw = 64 / lpb, waves = (n + bpw - 1) / bpw;
  coeff_cost_kernel<<<(waves + 3) / 4, 256, 0, st>>>(c, len, n, out, lpb, mode, w);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_coeff_abs_sum_batch(const int16_t *coeffs, int length, int n, uint32_t *out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return launch_coeff_cost(coeffs, length, n, out, 0, 0, uvghip_stream(stream));
}
extern "C" int uvghip_fast_coeff_cost_batch(const int16_t *coeffs, int width, int height, int n, uint64_t weights,
                                            uint32_t *out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return launch_coeff_cost(coeffs, width * height, n, out, 1, weights, uvghip_stream(stream));
}

// ---- fused TU round trip -----------------------------------------------------------

// MODE 0: the whole round trip.  The staged path (uvghip_quantize_residual_batch: RDOQ, LFNST, transform skip) cuts it
// at the quantiser: MODE 1 = residual + forward transform, the coefficients (int16, as the reference's coeff_t buffer) go
// to coeff_out; MODE 2 = coeff_out holds DEQUANTISED coefficients: inverse transform + reconstruction.  trskip replaces
// the transform passes by the identity (uvg_transformskip / uvg_itransformskip, transform.c:222-247).
enum { TU_FULL = 0, TU_FWD = 1, TU_INV = 2 };
template <typename PX, int MODE = TU_FULL>
__global__ void __launch_bounds__(256)
tu_roundtrip_kernel(tr_params P, quant_params Q, const PX *__restrict__ orig, int orig_stride,
                    const PX *__restrict__ pred, int pred_stride, PX *__restrict__ rec, int rec_stride,
                    const uvghip_tu_t *__restrict__ tus, int n, int bpg, int16_t *__restrict__ coeff_out,
                    uint8_t *__restrict__ has_coeffs, int trskip = 0)
{
  __shared__ __attribute__((aligned(16))) int16_t sA[TR_LINEBUF_ELEMS];
  __shared__ __attribute__((aligned(16))) int16_t sT[TR_LINEBUF_ELEMS];
  __shared__ __attribute__((aligned(16))) int16_t sMf1[TR_MATRIX_ELEMS], sMf2[TR_MATRIX_ELEMS];
  __shared__ __attribute__((aligned(16))) int16_t sMi1[TR_MATRIX_ELEMS], sMi2[TR_MATRIX_ELEMS];
  __shared__ int16_t sPred[1024];
  __shared__ int sHas[64];
  __shared__ int sX[64], sY[64];

  const int w = P.w, h = P.h, wh = w * h;
  const int blk0 = blockIdx.x * bpg;
  const int here = min(bpg, n - blk0);
  if (here <= 0) return;

  for (int b = threadIdx.x; b < here; b += blockDim.x) { sX[b] = tus[blk0 + b].x; sY[b] = tus[blk0 + b].y; sHas[b] = 0; }
  if (MODE != TU_INV) { tr_stage_matrix(sMf1, P.type_hor, w, false); tr_stage_matrix(sMf2, P.type_ver, h, false); }
  if (MODE != TU_FWD) { tr_stage_matrix(sMi1, P.type_ver, h, true); tr_stage_matrix(sMi2, P.type_hor, w, true); }
  __syncthreads();
  int16_t *gco = coeff_out + (size_t)blk0 * wh;

  if constexpr (MODE != TU_FULL) {
    if (trskip) {                               // identity "transform": residual <-> coefficient, element by element
      for (int e = threadIdx.x; e < here * wh; e += blockDim.x) {
        const int b = e / wh, rem = e - b * wh, y = rem / w, x = rem - y * w;
        const int p = pred[(size_t)(sY[b] + y) * pred_stride + sX[b] + x];
        if constexpr (MODE == TU_FWD) {
          const int o = orig[(size_t)(sY[b] + y) * orig_stride + sX[b] + x];
          gco[e] = (int16_t)(o - p);
        } else {
          const int sres = (int)(int16_t)(gco[e] + p);
          rec[(size_t)(sY[b] + y) * rec_stride + sX[b] + x] = (PX)clampi(sres, 0, px_traits<PX>::maxv);
        }
      }
      return;
    }
  }
  if constexpr (MODE == TU_INV) {
    // dequantised coefficients -> the inverse line buffer A'[b][i][j] (lines = columns, K = h); prediction for the final add
    const int pai1_ = tr_pitch(P.i1.K), ai1_blk_ = P.i1.R * pai1_;
    for (int e = threadIdx.x; e < here * wh; e += blockDim.x) {
      const int b = e / wh, rem = e - b * wh, y = rem / w, x = rem - y * w;
      sPred[e] = (int16_t)pred[(size_t)(sY[b] + y) * pred_stride + sX[b] + x];
      sA[b * ai1_blk_ + x * pai1_ + y] = gco[e];
    }
    __syncthreads();
  }
  if constexpr (MODE != TU_INV) {

  // residual = orig - pred (picture-generic.c:1360), forward line layout A[b][y][x]
  const int paf1 = tr_pitch(P.f1.K), af1_blk = P.f1.R * paf1;
  for (int e = threadIdx.x; e < here * wh; e += blockDim.x) {
    const int b = e / wh, rem = e - b * wh, y = rem / w, x = rem - y * w;
    const int o = orig[(size_t)(sY[b] + y) * orig_stride + sX[b] + x];
    const int p = pred[(size_t)(sY[b] + y) * pred_stride + sX[b] + x];
    sPred[e] = (int16_t)p;
    sA[b * af1_blk + y * paf1 + x] = (int16_t)(o - p);
  }
  __syncthreads();

  const int paf2 = tr_pitch(P.f2.K), af2_blk = P.f2.R * paf2;
  tr_run_pass<false>(P.f1, sA, af1_blk, sMf1, true, here,
                     [&](int b, int r, int c, int v) { sT[b * af2_blk + c * paf2 + r] = (int16_t)v; });
  __syncthreads();

  // vertical pass -> coefficient (j,i); quantise; store the level; dequantise into the
  // inverse line buffer A'[b][i][j] (lines = columns, K = h)
  const int pai1 = tr_pitch(P.i1.K), ai1_blk = P.i1.R * pai1;
  tr_run_pass<false>(P.f2, sT, af2_blk, sMf2, true, here, [&](int b, int r, int c, int v) {
    if constexpr (MODE == TU_FWD) { gco[b * wh + c * w + r] = (int16_t)v; return; }
    const int level = quant_one(v, Q);
    gco[b * wh + c * w + r] = (int16_t)level;
    if (level) sHas[b] = 1;                         // benign race: every writer stores 1
    sA[b * ai1_blk + r * pai1 + c] = (int16_t)dequant_one(level, Q);
  });
  if constexpr (MODE == TU_FWD) return;
  __syncthreads();
  }   // MODE != TU_INV
  const int pai1 = tr_pitch(P.i1.K), ai1_blk = P.i1.R * pai1;

  const int pai2 = tr_pitch(P.i2.K), ai2_blk = P.i2.R * pai2;
  tr_run_pass<true>(P.i1, sA, ai1_blk, sMi1, true, here,
                    [&](int b, int r, int c, int v) { sT[b * ai2_blk + c * pai2 + r] = (int16_t)v; });
  __syncthreads();
  tr_run_pass<true>(P.i2, sT, ai2_blk, sMi2, false, here, [&](int b, int r, int c, int v) {
    // int16 wrap of (residual + pred) then clip, as quant-generic.c:594-595
    const int s = (int)(int16_t)(v + sPred[b * wh + r * w + c]);
    rec[(size_t)(sY[b] + r) * rec_stride + sX[b] + c] = (PX)clampi(s, 0, px_traits<PX>::maxv);
  });
  if (has_coeffs)
    for (int b = threadIdx.x; b < here; b += blockDim.x) has_coeffs[blk0 + b] = (uint8_t)sHas[b];
}

// ---- fused TU round trip, small square TUs: one lane = one TU, everything in registers -----------
// 4x4 and 8x8 TUs without zero-out.  The kernel matrices are read through uniform (scalar) loads, so
// every multiply-accumulate is a v_dot2_i32_i16 with an SGPR coefficient pair; between passes the
// int16 intermediates are re-packed along the next pass's tap dimension (a compile-time register
// permutation).  Blocks in raster order make the row loads/stores of a wave contiguous.
//   fwd1  t[y][c]  = int16((sum_k res[y][k] Th[c][k] + a1) >> s1)      dct-generic.c:411 (truncation)
//   fwd2  co[j][c] = int16((sum_y t[y][c] Tv[j][y] + a2) >> s2)
//   quant / dequant                                                   quant-generic.c:51-121, :618-669
//   inv1  u[y][c]  = clip16((sum_j dq[j][c] Tv[j][y] + 64) >> 7)       dct-generic.c:438 (clip)
//   inv2  r[y][x]  = clip16((sum_c u[y][c] Th[c][x] + a4) >> s4)
template <int N, bool ROWPAIRS>
__device__ __forceinline__ uint32_t tu_coef_pair(const int16_t *T, int o, int t)
{
  if constexpr (ROWPAIRS) return reinterpret_cast<const uint32_t *>(T)[(o * N + 2 * t) >> 1];        // (T[o][2t], T[o][2t+1])
  else return (uint32_t)(uint16_t)T[(2 * t) * N + o] | ((uint32_t)(uint16_t)T[(2 * t + 1) * N + o] << 16);   // (T[2t][o], T[2t+1][o])
}
// out[l][o] = (sum_t in[l][t] . pair(o, t) + add) >> shift, optionally clipped to int16
template <int N, bool ROWPAIRS, bool CLIP>
__device__ __forceinline__ void tu_lane_pass(const uint32_t (&in)[N][N / 2], const int16_t *T, int shift, int (&out)[N][N])
{
  const int add = shift > 0 ? 1 << (shift - 1) : 0;
#pragma unroll
  for (int o = 0; o < N; ++o) {
    uint32_t cp[N / 2];
#pragma unroll
    for (int t = 0; t < N / 2; ++t) cp[t] = tu_coef_pair<N, ROWPAIRS>(T, o, t);
#pragma unroll
    for (int l = 0; l < N; ++l) {
      int acc = add;
#pragma unroll
      for (int t = 0; t < N / 2; ++t) acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, in[l][t]), __builtin_bit_cast(v2s, cp[t]), acc, false);
      acc >>= shift;
      out[l][o] = CLIP ? clampi(acc, -32768, 32767) : acc;      // forward passes: the int16 truncation happens when packing
    }
  }
}
// next[o][l/2] = (v[l][o], v[l+1][o]) low halves: lines become taps
template <int N>
__device__ __forceinline__ void tu_repack(const int (&v)[N][N], uint32_t (&next)[N][N / 2])
{
#pragma unroll
  for (int o = 0; o < N; ++o)
#pragma unroll
    for (int l = 0; l < N; l += 2) next[o][l >> 1] = __builtin_amdgcn_perm((uint32_t)v[l + 1][o], (uint32_t)v[l][o], 0x05040100u);
}

// MODE (as tu_roundtrip_kernel): TU_FWD stops after the forward transform (coefficients to coeff_out), TU_INV starts from
// dequantised coefficients in coeff_out.
template <typename PX, int N, int MODE = TU_FULL>
__global__ void __launch_bounds__(256)
tu_lane_kernel(tr_params P, quant_params Q, const PX *__restrict__ orig, int orig_stride,
               const PX *__restrict__ pred, int pred_stride, PX *__restrict__ rec, int rec_stride,
               const uvghip_tu_t *__restrict__ tus, int n, int16_t *__restrict__ coeff_out, uint8_t *__restrict__ has_coeffs)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uvghip_tu_t tu = tus[i];
  const int16_t *Th = tr_matrix_dev(P.type_hor, N), *Tv = tr_matrix_dev(P.type_ver, N);
  int pr[N][N];
  uint32_t a[N][N / 2];
  int v[N][N];
  int16_t *co = coeff_out + (size_t)i * (N * N);
  if constexpr (MODE != TU_INV) {
#pragma unroll
    for (int y = 0; y < N; ++y) {
      int o[N];
      load_row<PX, N>(orig, orig_stride, tu.x, tu.y + y, o);
      load_row<PX, N>(pred, pred_stride, tu.x, tu.y + y, pr[y]);
#pragma unroll
      for (int x = 0; x < N; x += 2)
        a[y][x >> 1] = __builtin_amdgcn_perm((uint32_t)(o[x + 1] - pr[y][x + 1]), (uint32_t)(o[x] - pr[y][x]), 0x05040100u);
    }
    tu_lane_pass<N, true, false>(a, Th, P.f1.shift, v);            // v[y][c]
    tu_repack<N>(v, a);                                             // a[c][y pairs]
    tu_lane_pass<N, true, false>(a, Tv, P.f2.shift, v);            // v[c][j]
  }
  if constexpr (MODE == TU_FWD) {                                   // coefficients (int16 truncation, dct-generic.c:724-725) row-major [j][c]
    uint32_t *dst = reinterpret_cast<uint32_t *>(co);
    if constexpr (N == 8) {
#pragma unroll
      for (int j = 0; j < N; ++j)
        *reinterpret_cast<uint4 *>(dst + j * 4) =
            make_uint4(__builtin_amdgcn_perm((uint32_t)v[1][j], (uint32_t)v[0][j], 0x05040100u), __builtin_amdgcn_perm((uint32_t)v[3][j], (uint32_t)v[2][j], 0x05040100u),
                       __builtin_amdgcn_perm((uint32_t)v[5][j], (uint32_t)v[4][j], 0x05040100u), __builtin_amdgcn_perm((uint32_t)v[7][j], (uint32_t)v[6][j], 0x05040100u));
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j)
        *reinterpret_cast<uint2 *>(dst + j * 2) =
            make_uint2(__builtin_amdgcn_perm((uint32_t)v[1][j], (uint32_t)v[0][j], 0x05040100u), __builtin_amdgcn_perm((uint32_t)v[3][j], (uint32_t)v[2][j], 0x05040100u));
    }
    return;
  }
  if constexpr (MODE == TU_INV) {
    // prediction rows for the final add; dequantised coefficients [j][c] -> a[c][j pairs] (taps of inv1 run over j)
#pragma unroll
    for (int y = 0; y < N; ++y) load_row<PX, N>(pred, pred_stride, tu.x, tu.y + y, pr[y]);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(co);
    uint32_t rows[N][N / 2];
    if constexpr (N == 8) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const uint4 w = *reinterpret_cast<const uint4 *>(src + j * 4);
        rows[j][0] = w.x; rows[j][1] = w.y; rows[j][2] = w.z; rows[j][3] = w.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const uint2 w = *reinterpret_cast<const uint2 *>(src + j * 2);
        rows[j][0] = w.x; rows[j][1] = w.y;
      }
    }
    if (Q.dq_on_load) {                                              // (uniform) levels in: uvg_dequant on the way, quant-generic.c:618-669
#pragma unroll
      for (int j = 0; j < N; ++j)
#pragma unroll
        for (int c = 0; c < N / 2; ++c) {
          const int lo = dequant_one((int)(int16_t)(rows[j][c] & 0xffffu), Q), hi = dequant_one((int)(int16_t)(rows[j][c] >> 16), Q);
          rows[j][c] = __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
        }
    }
#pragma unroll
    for (int c = 0; c < N; ++c)
#pragma unroll
      for (int j = 0; j < N; j += 2)
        a[c][j >> 1] = __builtin_amdgcn_perm(rows[j + 1][c >> 1], rows[j][c >> 1], (c & 1) ? 0x07060302u : 0x05040100u);
  }
  if constexpr (MODE == TU_FULL) {
  // quantise, store levels row-major [j][c], dequantise
  int any = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    int lv[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
      lv[c] = quant_one32((int)(int16_t)v[c][j], Q);
      any |= lv[c];
      v[c][j] = dequant_one(lv[c], Q);
    }
    if constexpr (N == 8) {
      uint4 w;
      w.x = (uint32_t)(uint16_t)lv[0] | ((uint32_t)lv[1] << 16); w.y = (uint32_t)(uint16_t)lv[2] | ((uint32_t)lv[3] << 16);
      w.z = (uint32_t)(uint16_t)lv[4] | ((uint32_t)lv[5] << 16); w.w = (uint32_t)(uint16_t)lv[6] | ((uint32_t)lv[7] << 16);
      *reinterpret_cast<uint4 *>(co + j * 8) = w;
    } else {
      uint2 w;
      w.x = (uint32_t)(uint16_t)lv[0] | ((uint32_t)lv[1] << 16); w.y = (uint32_t)(uint16_t)lv[2] | ((uint32_t)lv[3] << 16);
      *reinterpret_cast<uint2 *>(co + j * 4) = w;
    }
  }
  if (has_coeffs) has_coeffs[i] = any != 0;
  // v[c][j] = dequantised; taps of inv1 run over j: already the second index -> pack pairs along j
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int j = 0; j < N; j += 2) a[c][j >> 1] = __builtin_amdgcn_perm((uint32_t)v[c][j + 1], (uint32_t)v[c][j], 0x05040100u);
  }   // MODE == TU_FULL
  tu_lane_pass<N, false, true>(a, Tv, P.i1.shift, v);            // v[c][y] = sum_j dq[j][c] Tv[j][y]
  tu_repack<N>(v, a);                                             // a[y][c pairs]
  tu_lane_pass<N, false, true>(a, Th, P.i2.shift, v);            // v[y][x] = sum_c u[y][c] Th[c][x]
#pragma unroll
  for (int y = 0; y < N; ++y) {
    PX *q = rec + (size_t)(tu.y + y) * rec_stride + tu.x;
    int r[N];
#pragma unroll
    for (int x = 0; x < N; ++x) r[x] = clampi((int)(int16_t)(v[y][x] + pr[y][x]), 0, px_traits<PX>::maxv);   // quant-generic.c:594-595
    if constexpr (sizeof(PX) == 1) {
#pragma unroll
      for (int x = 0; x < N; x += 4)
        *reinterpret_cast<u32_unaligned *>(q + x) = (uint32_t)r[x] | ((uint32_t)r[x + 1] << 8) | ((uint32_t)r[x + 2] << 16) | ((uint32_t)r[x + 3] << 24);
    } else {
#pragma unroll
      for (int x = 0; x < N; x += 2) *reinterpret_cast<u32_unaligned *>(q + x) = (uint32_t)r[x] | ((uint32_t)r[x + 1] << 16);
    }
  }
}

// Stand-alone 4x4 / 8x8 transforms (uvghip_transform_batch without zero-out) on the same lane-per-block passes: the block
// never leaves the lane's registers.  Forward: in[y][x] -> out[j][c] truncated to int16 (dct-generic.c:724-725); inverse:
// in[j][c] -> out[y][x], both passes clipped to int16 (:735-736).
template <int N, bool INV>
__global__ void __launch_bounds__(256)
tr_lane_kernel(tr_params P, const int16_t *__restrict__ in, int16_t *__restrict__ out, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int16_t *Th = tr_matrix_dev(P.type_hor, N), *Tv = tr_matrix_dev(P.type_ver, N);
  const uint32_t *src = reinterpret_cast<const uint32_t *>(in + (size_t)i * (N * N));
  uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)i * (N * N));
  uint32_t a[N][N / 2];
  int v[N][N];
  if constexpr (!INV) {
#pragma unroll
    for (int y = 0; y < N; ++y)
#pragma unroll
      for (int x = 0; x < N / 2; ++x) a[y][x] = src[y * (N / 2) + x];     // pairs along x are adjacent in memory
    tu_lane_pass<N, true, false>(a, Th, P.f1.shift, v);            // v[y][c]
    tu_repack<N>(v, a);                                             // a[c][y pairs]
    tu_lane_pass<N, true, false>(a, Tv, P.f2.shift, v);            // v[c][j]
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int c = 0; c < N; c += 2) dst[j * (N / 2) + (c >> 1)] = __builtin_amdgcn_perm((uint32_t)v[c + 1][j], (uint32_t)v[c][j], 0x05040100u);
  } else {
    uint32_t rows[N][N / 2];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int c = 0; c < N / 2; ++c) rows[j][c] = src[j * (N / 2) + c];   // (coeff[j][2c], coeff[j][2c+1])
    // taps of the first inverse pass run over j: a[c][j/2] = (coeff[j][c], coeff[j+1][c])
#pragma unroll
    for (int c = 0; c < N; ++c)
#pragma unroll
      for (int j = 0; j < N; j += 2)
        a[c][j >> 1] = __builtin_amdgcn_perm(rows[j + 1][c >> 1], rows[j][c >> 1], (c & 1) ? 0x07060302u : 0x05040100u);
    tu_lane_pass<N, false, true>(a, Tv, P.i1.shift, v);            // v[c][y]
    tu_repack<N>(v, a);                                             // a[y][c pairs]
    tu_lane_pass<N, false, true>(a, Th, P.i2.shift, v);            // v[y][x]
#pragma unroll
    for (int y = 0; y < N; ++y)
#pragma unroll
      for (int x = 0; x < N; x += 2) dst[y * (N / 2) + (x >> 1)] = __builtin_amdgcn_perm((uint32_t)v[y][x + 1], (uint32_t)v[y][x], 0x05040100u);
  }
}

// called by uvghip_transform_batch (dct.hip) for square 4x4 / 8x8 blocks without zero-out
int uvghip_launch_tr_lane(const tr_params &P, bool inverse, const int16_t *in, int16_t *out, int n, hipStream_t st)
{
  const int g = (n + 255) / 256;
  if (P.w == 4) { if (inverse) tr_lane_kernel<4, true><<<g, 256, 0, st>>>(P, in, out, n); else tr_lane_kernel<4, false><<<g, 256, 0, st>>>(P, in, out, n); }
  else { if (inverse) tr_lane_kernel<8, true><<<g, 256, 0, st>>>(P, in, out, n); else tr_lane_kernel<8, false><<<g, 256, 0, st>>>(P, in, out, n); }
  UVGHIP_CHECK_LAUNCH();
}

// ---- fused TU round trip, 16x16 / 32x32 without zero-out: one wave = 1024 coefficients ----------
// A wave owns one 32x32 TU (or four 16x16 TUs) and walks the four passes on its own two LDS line
// buffers with no workgroup barrier after the matrices are staged.  Each lane computes a 4 lines x 4
// outputs register tile per pass: per 8 taps it loads 4 + 4 b128 rows (input lines, matrix rows) and
// issues 64 v_dot2 -- 8 LDS dwords per 64 MACs instead of 2 per MAC.  Pitches (32x32: 20 dwords per
// row, matrix rows 16..31 skewed by 4 dwords; 16x16: 12 dwords per row, 196 per block) keep every b128
// 16-byte aligned and put the lanes of each b128 service group on different 4-dword bank slots.  Results leave a pass already packed along the next pass's tap
// dimension (two dwords per output = one ds_write_b64).
template <int N> struct tuw {
  static constexpr int PITCH = N == 32 ? 40 : 24;                 // int16 per line
  // matrix row o starts at o * PITCH + skew(o): the 8 row groups a service group touches must differ in bank slot
  __host__ __device__ static constexpr int mrow(int o) { return o * PITCH + (N == 32 && o >= 16 ? 8 : 0); }
  static constexpr int BPW = 1024 / (N * N);                      // TUs per wave
  static constexpr int BLK = N * PITCH + (N == 16 ? 8 : 0);       // int16 per TU in a line buffer
  static constexpr int BUF = BPW * BLK;
  static constexpr int MAT = N * PITCH;
};

// acc[i][j] = sum_k A[l0+i][k] * B[o0+j][k]
template <int N>
__device__ __forceinline__ void tuw_mma(const int16_t *A, const int16_t *B, int (&acc)[4][4])
{
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
#pragma unroll
  for (int kk = 0; kk < N / 8; ++kk) {
    uint4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const uint4 *>(A + i * tuw<N>::PITCH + kk * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const uint4 *>(B + j * tuw<N>::PITCH + kk * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int s = acc[i][j];
        s = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a[i].x), __builtin_bit_cast(v2s, b[j].x), s, false);
        s = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a[i].y), __builtin_bit_cast(v2s, b[j].y), s, false);
        s = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a[i].z), __builtin_bit_cast(v2s, b[j].z), s, false);
        s = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a[i].w), __builtin_bit_cast(v2s, b[j].w), s, false);
        acc[i][j] = s;
      }
  }
}
__device__ __forceinline__ void tuw_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// v[i][j] (line l0+i, output o0+j) -> dst[o0+j][l0 .. l0+3] (lines become taps)
template <int N>
__device__ __forceinline__ void tuw_store_transposed(int16_t *dst, int l0, const int (&v)[4][4])
{
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<uint2 *>(dst + j * tuw<N>::PITCH + l0) =
        make_uint2(__builtin_amdgcn_perm((uint32_t)v[1][j], (uint32_t)v[0][j], 0x05040100u),
                   __builtin_amdgcn_perm((uint32_t)v[3][j], (uint32_t)v[2][j], 0x05040100u));
}

template <typename PX, int N, int MODE = TU_FULL>
__global__ void __launch_bounds__(256)
tu_wave_kernel(tr_params P, quant_params Q, const PX *__restrict__ orig, int orig_stride,
               const PX *__restrict__ pred, int pred_stride, PX *__restrict__ rec, int rec_stride,
               const uvghip_tu_t *__restrict__ tus, int n, int16_t *__restrict__ coeff_out, uint8_t *__restrict__ has_coeffs)
{
  using W = tuw<N>;
  constexpr int MF = 0, MI = MODE == TU_FULL ? 2 : 0;                  // first forward / inverse matrix (the halves stage only their two)
  __shared__ __attribute__((aligned(16))) int16_t sM[MODE == TU_FULL ? 4 : 2][W::MAT];   // Bf1, Bf2, Bi1, Bi2
  __shared__ __attribute__((aligned(16))) int16_t sBuf[4][2][W::BUF];  // per wave: two line buffers
  {
    const int16_t *Th = tr_matrix_dev(P.type_hor, N), *Tv = tr_matrix_dev(P.type_ver, N);
    for (int e = threadIdx.x; e < N * N; e += 256) {
      const int r = e / N, c = e - r * N;                                // compile-time N: shifts
      if constexpr (MODE != TU_INV) {
        sM[MF][W::mrow(r) + c] = Th[e];                                  // Bf1[c'][k]  = Th[c'][k]
        sM[MF + 1][W::mrow(r) + c] = Tv[e];                              // Bf2[j][y]   = Tv[j][y]
      }
      if constexpr (MODE != TU_FWD) {
        sM[MI][W::mrow(c) + r] = Tv[e];                                  // Bi1[y][j]   = Tv[j][y]
        sM[MI + 1][W::mrow(c) + r] = Th[e];                              // Bi2[x][c']  = Th[c'][x]
      }
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + wave;
  const int tu0 = unit * W::BPW;
  if (tu0 >= n) return;
  int16_t *bufA = sBuf[wave][0], *bufB = sBuf[wave][1];
  // lane -> (TU b, line group, output group)
  const int b = N == 32 ? 0 : lane >> 4;
  const int l0 = (N == 32 ? lane >> 3 : (lane >> 2) & 3) * 4;
  const int o0 = (N == 32 ? lane & 7 : lane & 3) * 4;
  const bool on = tu0 + b < n;
  const uvghip_tu_t tu = tus[on ? tu0 + b : tu0];

  int acc[4][4];
  if constexpr (MODE == TU_INV) {
    // dequantised coefficients, one row segment of 16 per lane: in[j = row][c = x0 + k] -> bufA[c][j] (lines c, taps j)
    const int row = N == 32 ? lane >> 1 : lane & 15, x0 = N == 32 ? (lane & 1) * 16 : 0;
    const int lb = N == 32 ? 0 : lane >> 4;
    const int16_t *src = coeff_out + (size_t)(tu0 + lb < n ? tu0 + lb : tu0) * (N * N) + row * N + x0;
    const uint4 v0 = *reinterpret_cast<const uint4 *>(src), v1 = *reinterpret_cast<const uint4 *>(src + 8);
    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    int16_t *dst = bufA + lb * W::BLK + x0 * W::PITCH + row;
    const bool dq = Q.dq_on_load != 0;                                  // (uniform) levels in: uvg_dequant on the way
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int cv = (int)(int16_t)(w[k >> 1] >> (16 * (k & 1)));
      dst[k * W::PITCH] = (int16_t)(dq ? dequant_one(cv, Q) : cv);
    }
  }
  if constexpr (MODE != TU_INV) {
  // residual rows -> bufA[b][y][x]
  {
    const int row = N == 32 ? lane >> 1 : lane & 15, x0 = N == 32 ? (lane & 1) * 16 : 0;
    const int lb = N == 32 ? 0 : lane >> 4;
    const uvghip_tu_t t2 = tus[tu0 + lb < n ? tu0 + lb : tu0];
    int o[16], p[16];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int o8[8], p8[8];
      load_row<PX, 8>(orig, orig_stride, t2.x + x0 + q * 8, t2.y + row, o8);
      load_row<PX, 8>(pred, pred_stride, t2.x + x0 + q * 8, t2.y + row, p8);
#pragma unroll
      for (int k = 0; k < 8; ++k) { o[q * 8 + k] = o8[k]; p[q * 8 + k] = p8[k]; }
    }
    int16_t *dst = bufA + lb * W::BLK + row * W::PITCH + x0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<uint2 *>(dst + q * 4) =
          make_uint2(__builtin_amdgcn_perm((uint32_t)(o[q * 4 + 1] - p[q * 4 + 1]), (uint32_t)(o[q * 4] - p[q * 4]), 0x05040100u),
                     __builtin_amdgcn_perm((uint32_t)(o[q * 4 + 3] - p[q * 4 + 3]), (uint32_t)(o[q * 4 + 2] - p[q * 4 + 2]), 0x05040100u));
  }
  tuw_sync();

  // fwd1: lines y, outputs c -> bufB[c][y]
  tuw_mma<N>(bufA + b * W::BLK + l0 * W::PITCH, sM[MF] + W::mrow(o0), acc);
  {
    const int add = P.f1.shift > 0 ? 1 << (P.f1.shift - 1) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (acc[i][j] + add) >> P.f1.shift;     // int16 truncation on packing
    tuw_store_transposed<N>(bufB + b * W::BLK + o0 * W::PITCH, l0, acc);
  }
  tuw_sync();
  // fwd2: lines c, outputs j -> quantise; levels to HBM [j][c]; dequantised -> bufA[c][j]
  tuw_mma<N>(bufB + b * W::BLK + l0 * W::PITCH, sM[MF + 1] + W::mrow(o0), acc);
  if constexpr (MODE == TU_FWD) {                                       // coefficients (int16 truncation) to HBM [j][c]
    if (!on) return;
    const int add = 1 << (P.f2.shift - 1);
    int16_t *co = coeff_out + (size_t)(tu0 + b) * (N * N);
#pragma unroll
    for (int j = 0; j < 4; ++j) {     // row j = o0 + j, columns c = l0 .. l0+3
      int cv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) cv[i] = (acc[i][j] + add) >> P.f2.shift;
      *reinterpret_cast<uint2 *>(co + (o0 + j) * N + l0) =
          make_uint2(__builtin_amdgcn_perm((uint32_t)cv[1], (uint32_t)cv[0], 0x05040100u), __builtin_amdgcn_perm((uint32_t)cv[3], (uint32_t)cv[2], 0x05040100u));
    }
    return;
  }
  int any = 0;
  {
    const int add = 1 << (P.f2.shift - 1);
    int lv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lv[i][j] = quant_one32((int)(int16_t)((acc[i][j] + add) >> P.f2.shift), Q);
        any |= lv[i][j];
        acc[i][j] = dequant_one(lv[i][j], Q);
      }
    if (on) {
      int16_t *co = coeff_out + (size_t)(tu0 + b) * (N * N);
#pragma unroll
      for (int j = 0; j < 4; ++j)     // row j = o0 + j, columns c = l0 .. l0+3
        *reinterpret_cast<uint2 *>(co + (o0 + j) * N + l0) =
            make_uint2((uint32_t)(uint16_t)lv[0][j] | ((uint32_t)lv[1][j] << 16), (uint32_t)(uint16_t)lv[2][j] | ((uint32_t)lv[3][j] << 16));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)       // natural orientation: line c = l0 + i, taps j = o0 .. o0+3
      *reinterpret_cast<uint2 *>(bufA + b * W::BLK + (l0 + i) * W::PITCH + o0) =
          make_uint2(__builtin_amdgcn_perm((uint32_t)acc[i][1], (uint32_t)acc[i][0], 0x05040100u),
                     __builtin_amdgcn_perm((uint32_t)acc[i][3], (uint32_t)acc[i][2], 0x05040100u));
  }
  if (has_coeffs) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(any != 0);
    if (N == 32) { if (lane == 0) has_coeffs[tu0] = m != 0; }
    else if ((lane & 15) == 0 && on) has_coeffs[tu0 + b] = ((m >> (lane & 48)) & 0xffffull) != 0;
  }
  }   // MODE != TU_INV
  tuw_sync();
  // inv1: lines c, taps j, outputs y -> bufB[y][c]
  tuw_mma<N>(bufA + b * W::BLK + l0 * W::PITCH, sM[MI] + W::mrow(o0), acc);
  {
    const int add = 1 << (P.i1.shift - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = clampi((acc[i][j] + add) >> P.i1.shift, -32768, 32767);
    tuw_store_transposed<N>(bufB + b * W::BLK + o0 * W::PITCH, l0, acc);
  }
  tuw_sync();
  // inv2: lines y, taps c, outputs x -> + pred -> rec
  tuw_mma<N>(bufB + b * W::BLK + l0 * W::PITCH, sM[MI + 1] + W::mrow(o0), acc);
  if (on) {
    const int add = 1 << (P.i2.shift - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int p4[4];
      load_row<PX, 4>(pred, pred_stride, tu.x + o0, tu.y + l0 + i, p4);
      int r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = clampi((acc[i][j] + add) >> P.i2.shift, -32768, 32767);
        r[j] = clampi((int)(int16_t)(v + p4[j]), 0, px_traits<PX>::maxv);          // quant-generic.c:594-595
      }
      PX *q = rec + (size_t)(tu.y + l0 + i) * rec_stride + tu.x + o0;
      if constexpr (sizeof(PX) == 1) *reinterpret_cast<u32_unaligned *>(q) = (uint32_t)r[0] | ((uint32_t)r[1] << 8) | ((uint32_t)r[2] << 16) | ((uint32_t)r[3] << 24);
      else { *reinterpret_cast<u32_unaligned *>(q) = (uint32_t)r[0] | ((uint32_t)r[1] << 16); *reinterpret_cast<u32_unaligned *>(q + 2) = (uint32_t)r[2] | ((uint32_t)r[3] << 16); }
    }
  }
}

// Stand-alone 16x16 / 32x32 transforms (uvghip_transform_batch without zero-out) on the wave-per-1024-coefficient passes.
// Forward: in[y][x] -> out[j][c] (int16 truncation, dct-generic.c:724-725); inverse: in[j][c] -> out[y][x], clipped after
// each pass (:735-736).  The inverse stages its input transposed (its first pass contracts over j).
template <int N, bool INV>
__global__ void __launch_bounds__(256)
tr_wave_kernel(tr_params P, const int16_t *__restrict__ in, int16_t *__restrict__ out, int n)
{
  using W = tuw<N>;
  __shared__ __attribute__((aligned(16))) int16_t sM[2][W::MAT];
  __shared__ __attribute__((aligned(16))) int16_t sBuf[4][2][W::BUF];
  {
    const int16_t *Th = tr_matrix_dev(P.type_hor, N), *Tv = tr_matrix_dev(P.type_ver, N);
    for (int e = threadIdx.x; e < N * N; e += 256) {
      const int r = e / N, c = e - r * N;
      if constexpr (!INV) { sM[0][W::mrow(r) + c] = Th[e]; sM[1][W::mrow(r) + c] = Tv[e]; }      // Bf1[c'][k], Bf2[j][y]
      else { sM[0][W::mrow(c) + r] = Tv[e]; sM[1][W::mrow(c) + r] = Th[e]; }                     // Bi1[y][j], Bi2[x][c']
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tu0 = (blockIdx.x * 4 + wave) * W::BPW;
  if (tu0 >= n) return;
  int16_t *bufA = sBuf[wave][0], *bufB = sBuf[wave][1];
  const int b = N == 32 ? 0 : lane >> 4;
  const int l0 = (N == 32 ? lane >> 3 : (lane >> 2) & 3) * 4;
  const int o0 = (N == 32 ? lane & 7 : lane & 3) * 4;
  const bool on = tu0 + b < n;
  {
    // one input row segment of 16 coefficients per lane
    const int row = N == 32 ? lane >> 1 : lane & 15, x0 = N == 32 ? (lane & 1) * 16 : 0;
    const int lb = N == 32 ? 0 : lane >> 4;
    const int16_t *src = in + (size_t)(tu0 + lb < n ? tu0 + lb : tu0) * (N * N) + row * N + x0;
    const uint4 v0 = *reinterpret_cast<const uint4 *>(src), v1 = *reinterpret_cast<const uint4 *>(src + 8);
    if constexpr (!INV) {
      int16_t *dst = bufA + lb * W::BLK + row * W::PITCH + x0;
      *reinterpret_cast<uint4 *>(dst) = v0; *reinterpret_cast<uint4 *>(dst + 8) = v1;
    } else {
      // in[j = row][c = x0 + k] -> bufA[c][j]
      const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      int16_t *dst = bufA + lb * W::BLK + x0 * W::PITCH + row;
#pragma unroll
      for (int k = 0; k < 16; ++k) dst[k * W::PITCH] = (int16_t)(w[k >> 1] >> (16 * (k & 1)));
    }
  }
  tuw_sync();
  int acc[4][4];
  const tr_pass &p1 = INV ? P.i1 : P.f1, &p2 = INV ? P.i2 : P.f2;
  tuw_mma<N>(bufA + b * W::BLK + l0 * W::PITCH, sM[0] + W::mrow(o0), acc);
  {
    const int add = p1.shift > 0 ? 1 << (p1.shift - 1) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = (acc[i][j] + add) >> p1.shift;
        acc[i][j] = INV ? clampi(v, -32768, 32767) : v;            // forward: int16 truncation on packing
      }
    tuw_store_transposed<N>(bufB + b * W::BLK + o0 * W::PITCH, l0, acc);
  }
  tuw_sync();
  tuw_mma<N>(bufB + b * W::BLK + l0 * W::PITCH, sM[1] + W::mrow(o0), acc);
  if (!on) return;
  const int add2 = 1 << (p2.shift - 1);
  int16_t *dstb = out + (size_t)(tu0 + b) * (N * N);
  if constexpr (!INV) {
    // lines c = l0 + i, outputs j = o0 + jj: out[j][c]
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      int v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (acc[i][jj] + add2) >> p2.shift;
      *reinterpret_cast<uint2 *>(dstb + (o0 + jj) * N + l0) =
          make_uint2(__builtin_amdgcn_perm((uint32_t)v[1], (uint32_t)v[0], 0x05040100u), __builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[2], 0x05040100u));
    }
  } else {
    // lines y = l0 + i, outputs x = o0 + jj: out[y][x]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int v[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) v[jj] = clampi((acc[i][jj] + add2) >> p2.shift, -32768, 32767);
      *reinterpret_cast<uint2 *>(dstb + (l0 + i) * N + o0) =
          make_uint2(__builtin_amdgcn_perm((uint32_t)v[1], (uint32_t)v[0], 0x05040100u), __builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[2], 0x05040100u));
    }
  }
}

int uvghip_launch_tr_wave(const tr_params &P, bool inverse, const int16_t *in, int16_t *out, int n, hipStream_t st)
{
  const int units = P.w == 32 ? n : (n + 3) / 4, g = (units + 3) / 4;
  if (P.w == 16) { if (inverse) tr_wave_kernel<16, true><<<g, 256, 0, st>>>(P, in, out, n); else tr_wave_kernel<16, false><<<g, 256, 0, st>>>(P, in, out, n); }
  else { if (inverse) tr_wave_kernel<32, true><<<g, 256, 0, st>>>(P, in, out, n); else tr_wave_kernel<32, false><<<g, 256, 0, st>>>(P, in, out, n); }
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_tu_roundtrip_batch(int bitdepth, int type_hor, int type_ver, int skip_width, int skip_height,
                                         int width, int height, int qp_scaled, int slice_is_intra,
                                         const void *orig, int orig_stride, const void *pred, int pred_stride,
                                         void *rec, int rec_stride, const uvghip_tu_t *tus, int n,
                                         int16_t *coeff_out, uint8_t *has_coeffs, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!tr_valid_dim(width) || !tr_valid_dim(height) || type_hor < 0 || type_hor > 2 || type_ver < 0 || type_ver > 2 ||
      skip_width < 0 || skip_width >= width || skip_height < 0 || skip_height >= height)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (int rc = check_qargs(bitdepth, width, height, qp_scaled)) return rc;
  // as uvghip_transform_batch: zero-out counts in units of 4, and the level buffer is written with 8 / 16 byte stores
  if ((skip_width & 3) || (skip_height & 3) || ((uintptr_t)coeff_out & 15)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  const tr_params P = tr_make_params(bitdepth, type_hor, type_ver, width, height, skip_width, skip_height);
  const quant_params Q = make_quant_params(bitdepth, width, height, qp_scaled, 0, slice_is_intra);
  hipStream_t st = uvghip_stream(stream);
  if (width == height && width <= 8 && skip_width == 0 && skip_height == 0) {
    const int g = (n + 255) / 256;
#define TU_LANE(PX, N) tu_lane_kernel<PX, N><<<g, 256, 0, st>>>(P, Q, (const PX *)orig, orig_stride, (const PX *)pred, pred_stride, (PX *)rec, rec_stride, tus, n, coeff_out, has_coeffs)
    if (bitdepth == 8) { if (width == 4) TU_LANE(uint8_t, 4); else TU_LANE(uint8_t, 8); }
    else { if (width == 4) TU_LANE(uint16_t, 4); else TU_LANE(uint16_t, 8); }
#undef TU_LANE
    UVGHIP_CHECK_LAUNCH();
  }
  if (width == height && width >= 16 && skip_width == 0 && skip_height == 0) {
    const int units = width == 32 ? n : (n + 3) / 4, g = (units + 3) / 4;
#define TU_WAVE(PX, N) tu_wave_kernel<PX, N><<<g, 256, 0, st>>>(P, Q, (const PX *)orig, orig_stride, (const PX *)pred, pred_stride, (PX *)rec, rec_stride, tus, n, coeff_out, has_coeffs)
    if (bitdepth == 8) { if (width == 16) TU_WAVE(uint8_t, 16); else TU_WAVE(uint8_t, 32); }
    else { if (width == 16) TU_WAVE(uint16_t, 16); else TU_WAVE(uint16_t, 32); }
#undef TU_WAVE
    UVGHIP_CHECK_LAUNCH();
  }
  const int bpg = 1024 / (width * height);
  const int grid = (n + bpg - 1) / bpg;
  if (bitdepth == 8)
    tu_roundtrip_kernel<uint8_t><<<grid, 256, 0, st>>>(P, Q, (const uint8_t *)orig, orig_stride, (const uint8_t *)pred, pred_stride,
                                                       (uint8_t *)rec, rec_stride, tus, n, bpg, coeff_out, has_coeffs);
  else
    tu_roundtrip_kernel<uint16_t><<<grid, 256, 0, st>>>(P, Q, (const uint16_t *)orig, orig_stride, (const uint16_t *)pred, pred_stride,
                                                        (uint16_t *)rec, rec_stride, tus, n, bpg, coeff_out, has_coeffs);
  UVGHIP_CHECK_LAUNCH();
}

// ---- the two halves of the round trip as entry points of their own (a host that batches the quantiser over several
//      pictures, as the pipeline does for RDOQ, runs them per picture around one quantiser launch) ----
static int tu_half_check(int bitdepth, int type_hor, int type_ver, int skip_width, int skip_height, int width, int height, const char *who)
{
  if ((bitdepth != 8 && bitdepth != 10) || !tr_valid_dim(width) || !tr_valid_dim(height) || type_hor < 0 || type_hor > 2 || type_ver < 0 ||
      type_ver > 2 || skip_width < 0 || skip_width >= width || skip_height < 0 || skip_height >= height)
    return uvghip_set_error(hipErrorInvalidValue, who);
  return 0;
}

// square blocks without zero-out or transform skip take the register / wave kernels of the fused round trip (their 8 / 16 byte
// coefficient accesses want the buffer 16-byte aligned); everything else the generic LDS kernel
static bool tu_half_fast(int width, int height, int skip_width, int skip_height, int use_trskip, const void *coef)
{
  return width == height && skip_width == 0 && skip_height == 0 && !use_trskip && ((uintptr_t)coef & 15) == 0;
}
#define TU_HALF_LAUNCH(KERNEL, PX, N, MODE, G, ORIG, OSTRIDE, COEF)                                                           \
  KERNEL<PX, N, MODE><<<G, 256, 0, st>>>(P, Q, (const PX *)(ORIG), OSTRIDE, (const PX *)pred, pred_stride, (PX *)rec, rec_stride, \
                                         tus, n, COEF, nullptr)
#define TU_HALF_DISPATCH(MODE, ORIG, OSTRIDE, COEF)                                                                           \
  do {                                                                                                                        \
    if (width <= 8) {                                                                                                         \
      const int g = (n + 255) / 256;                                                                                          \
      if (bitdepth == 8) { if (width == 4) TU_HALF_LAUNCH(tu_lane_kernel, uint8_t, 4, MODE, g, ORIG, OSTRIDE, COEF);          \
                           else TU_HALF_LAUNCH(tu_lane_kernel, uint8_t, 8, MODE, g, ORIG, OSTRIDE, COEF); }                   \
      else { if (width == 4) TU_HALF_LAUNCH(tu_lane_kernel, uint16_t, 4, MODE, g, ORIG, OSTRIDE, COEF);                       \
             else TU_HALF_LAUNCH(tu_lane_kernel, uint16_t, 8, MODE, g, ORIG, OSTRIDE, COEF); }                                \
    } else {                                                                                                                  \
      const int units = width == 32 ? n : (n + 3) / 4, g = (units + 3) / 4;                                                   \
      if (bitdepth == 8) { if (width == 16) TU_HALF_LAUNCH(tu_wave_kernel, uint8_t, 16, MODE, g, ORIG, OSTRIDE, COEF);        \
                           else TU_HALF_LAUNCH(tu_wave_kernel, uint8_t, 32, MODE, g, ORIG, OSTRIDE, COEF); }                  \
      else { if (width == 16) TU_HALF_LAUNCH(tu_wave_kernel, uint16_t, 16, MODE, g, ORIG, OSTRIDE, COEF);                     \
             else TU_HALF_LAUNCH(tu_wave_kernel, uint16_t, 32, MODE, g, ORIG, OSTRIDE, COEF); }                               \
    }                                                                                                                         \
    UVGHIP_CHECK_LAUNCH();                                                                                                    \
  } while (0)

extern "C" int uvghip_tu_forward_batch(int bitdepth, int type_hor, int type_ver, int skip_width, int skip_height, int width, int height,
                                       int use_trskip, const void *orig, int orig_stride, const void *pred, int pred_stride,
                                       const uvghip_tu_t *tus, int n, int16_t *coef_out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (int rc = tu_half_check(bitdepth, type_hor, type_ver, skip_width, skip_height, width, height, __func__)) return rc;
  if (n <= 0) return 0;
  const tr_params P = tr_make_params(bitdepth, type_hor, type_ver, width, height, skip_width, skip_height);
  const quant_params Q = make_quant_params(bitdepth, width, height, 22, 0, 1);       // unused by this half
  const int bpg = 1024 / (width * height), grid = (n + bpg - 1) / bpg;
  hipStream_t st = uvghip_stream(stream);
  if (tu_half_fast(width, height, skip_width, skip_height, use_trskip, coef_out)) {
    void *rec = nullptr; const int rec_stride = 0;
    TU_HALF_DISPATCH(TU_FWD, orig, orig_stride, coef_out);
  }
  if (bitdepth == 8) tu_roundtrip_kernel<uint8_t, TU_FWD><<<grid, 256, 0, st>>>(P, Q, (const uint8_t *)orig, orig_stride, (const uint8_t *)pred, pred_stride, nullptr, 0, tus, n, bpg, coef_out, nullptr, use_trskip);
  else tu_roundtrip_kernel<uint16_t, TU_FWD><<<grid, 256, 0, st>>>(P, Q, (const uint16_t *)orig, orig_stride, (const uint16_t *)pred, pred_stride, nullptr, 0, tus, n, bpg, coef_out, nullptr, use_trskip);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_tu_inverse_batch(int bitdepth, int type_hor, int type_ver, int skip_width, int skip_height, int width, int height,
                                       int use_trskip, const int16_t *coef_in, const void *pred, int pred_stride, void *rec, int rec_stride,
                                       const uvghip_tu_t *tus, int n, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (int rc = tu_half_check(bitdepth, type_hor, type_ver, skip_width, skip_height, width, height, __func__)) return rc;
  if (n <= 0) return 0;
  const tr_params P = tr_make_params(bitdepth, type_hor, type_ver, width, height, skip_width, skip_height);
  const quant_params Q = make_quant_params(bitdepth, width, height, 22, 0, 1);
  const int bpg = 1024 / (width * height), grid = (n + bpg - 1) / bpg;
  hipStream_t st = uvghip_stream(stream);
  int16_t *c = const_cast<int16_t *>(coef_in);
  if (tu_half_fast(width, height, skip_width, skip_height, use_trskip, coef_in)) {
    const void *orig = nullptr; const int orig_stride = 0;
    TU_HALF_DISPATCH(TU_INV, orig, orig_stride, c);
  }
  if (bitdepth == 8) tu_roundtrip_kernel<uint8_t, TU_INV><<<grid, 256, 0, st>>>(P, Q, nullptr, 0, (const uint8_t *)pred, pred_stride, (uint8_t *)rec, rec_stride, tus, n, bpg, c, nullptr, use_trskip);
  else tu_roundtrip_kernel<uint16_t, TU_INV><<<grid, 256, 0, st>>>(P, Q, nullptr, 0, (const uint16_t *)pred, pred_stride, (uint16_t *)rec, rec_stride, tus, n, bpg, c, nullptr, use_trskip);
  UVGHIP_CHECK_LAUNCH();
}

// uvghip_dequant_batch + uvghip_tu_inverse_batch in one launch for the square shapes of the register / wave kernels: the
// kernels dequantise the levels as they load them (one launch and one trip of the coefficients through HBM less).
extern "C" int uvghip_tu_dequant_inverse_batch(int bitdepth, int type_hor, int type_ver, int width, int height, int qp_scaled,
                                               const int16_t *levels, const void *pred, int pred_stride, void *rec, int rec_stride,
                                               const uvghip_tu_t *tus, int n, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (int rc = tu_half_check(bitdepth, type_hor, type_ver, 0, 0, width, height, __func__)) return rc;
  if (int rc = check_qargs(bitdepth, width, height, qp_scaled)) return rc;
  if (!tu_half_fast(width, height, 0, 0, 0, levels))
    return uvghip_set_error(hipErrorNotSupported, "uvghip_tu_dequant_inverse_batch: square blocks, 16-byte aligned levels (else dequant_batch + tu_inverse_batch)");
  if (n <= 0) return 0;
  const tr_params P = tr_make_params(bitdepth, type_hor, type_ver, width, height, 0, 0);
  quant_params Q = make_quant_params(bitdepth, width, height, qp_scaled, 0, 1);
  Q.dq_on_load = 1;
  hipStream_t st = uvghip_stream(stream);
  int16_t *c = const_cast<int16_t *>(levels);
  const void *orig = nullptr; const int orig_stride = 0;
  TU_HALF_DISPATCH(TU_INV, orig, orig_stride, c);
}

// ---- the staged round trip: every branch of uvg_quantize_residual (quant-generic.c:460-612) ----
__global__ void __launch_bounds__(256) has_coeffs_kernel(const int16_t *__restrict__ q, int len, int n, uint8_t *__restrict__ has)
{
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= n) return;
  int any = 0;
  for (int i = lane; i < len; i += 64) any |= q[(size_t)b * len + i];
  any = __any(any != 0);
  if (lane == 0) has[b] = any ? 1 : 0;
}

int uvghip_rdoq_launch_checked(int bitdepth, const int16_t *coef, int16_t *q_coef, int width, int height, int n, int color, int block_type,
                               int cbf_u, int lfnst_idx, int mts_idx, int qp_scaled, double lambda, const uvghip_rdoq_ctx_t *ctx_host,
                               void *workspace, size_t workspace_bytes, uint8_t *has_coeffs, void *stream, int signhide = 0)
{
  return (signhide ? uvghip_rdoq_signhide_batch : uvghip_rdoq_batch)(bitdepth, coef, q_coef, width, height, n, color, block_type, cbf_u, lfnst_idx, mts_idx, qp_scaled, lambda, ctx_host,
                           workspace, workspace_bytes, nullptr, has_coeffs, stream);
}

static size_t qr_coef_bytes(int width, int height, int n) { return (((size_t)width * height * n * sizeof(int16_t)) + 255) & ~(size_t)255; }

extern "C" size_t uvghip_quantize_residual_workspace_bytes(const uvghip_qr_params_t *p, int n)
{
  if (!p || n <= 0) return 0;
  const bool rdoq = p->rdoq_enable && (p->width > 4 || !p->rdoq_skip) && !p->use_trskip;
  return 2 * qr_coef_bytes(p->width, p->height, n) +
         (rdoq ? (p->signhide_enable ? uvghip_rdoq_signhide_workspace_bytes : uvghip_rdoq_workspace_bytes)(p->width, p->height, n) : 0);
}

extern "C" int uvghip_quantize_residual_batch(int bitdepth, const uvghip_qr_params_t *p, const void *orig, int orig_stride,
                                              const void *pred, int pred_stride, void *rec, int rec_stride, const uvghip_tu_t *tus,
                                              int n, const uvghip_lfnst_tu_t *lfnst_tus, int16_t *coeff_out, uint8_t *has_coeffs,
                                              void *workspace, size_t workspace_bytes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!p) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int width = p->width, height = p->height;
  if (!tr_valid_dim(width) || !tr_valid_dim(height) || p->type_hor < 0 || p->type_hor > 2 || p->type_ver < 0 || p->type_ver > 2 ||
      p->skip_width < 0 || p->skip_width >= width || p->skip_height < 0 || p->skip_height >= height || p->color < 0 || p->color > 2 ||
      !coeff_out || !has_coeffs)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (int rc = check_qargs(bitdepth, width, height, p->qp_scaled)) return rc;
  if (p->dep_quant) return uvghip_set_error(hipErrorNotSupported, "uvghip_quantize_residual_batch: dependent quantisation is not built");
  if (p->rdoq_enable && p->use_trskip)
    return uvghip_set_error(hipErrorNotSupported, "uvghip_quantize_residual_batch: transform-skip RDOQ (uvg_ts_rdoq) is not built");
  if (n <= 0) return 0;
  if (!workspace || workspace_bytes < uvghip_quantize_residual_workspace_bytes(p, n))
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_quantize_residual_batch: workspace");
  const bool rdoq = p->rdoq_enable && (width > 4 || !p->rdoq_skip) && !p->use_trskip;
  // The quantiser's and dequantiser's transform-skip flag is NOT use_trskip: the reference passes
  // cur_cu->tr_idx == MTS_SKIP && color == COLOR_Y (quant-generic.c:537-539, 558-559), so a chroma block coded with transform
  // skip (cur_pu->tr_skip & (1 << color)) gets the identity transform but the ordinary quantiser shifts.
  const int quant_ts = p->use_trskip && p->color == 0;
  hipStream_t st = uvghip_stream(stream);
  int16_t *coef = static_cast<int16_t *>(workspace);
  int16_t *deq = reinterpret_cast<int16_t *>(static_cast<char *>(workspace) + qr_coef_bytes(width, height, n));
  void *rdoq_ws = static_cast<char *>(workspace) + 2 * qr_coef_bytes(width, height, n);
  // (1) residual -> transform (or transform skip) -> coefficients
  if (int rc = uvghip_tu_forward_batch(bitdepth, p->type_hor, p->type_ver, p->skip_width, p->skip_height, width, height, p->use_trskip,
                                       orig, orig_stride, pred, pred_stride, tus, n, coef, stream))
    return rc;
  // (2) forward LFNST (intra CUs, cfg.lfnst; :507-510): where it applies is the caller's call (uvg_fwd_lfnst, transform.c:988:
  //     luma, or chroma of a separate tree) -- lfnst_tus == NULL means "nowhere", although quant / RDOQ still see lfnst_idx
  if (lfnst_tus)
    if (int rc = uvghip_lfnst_batch(0, coef, width, height, lfnst_tus, n, stream)) return rc;
  // (3) quantisation: RDOQ (:527-531) or uvg_quant (:535-539)
  if (rdoq) {
    if (int rc = uvghip_rdoq_launch_checked(bitdepth, coef, coeff_out, width, height, n, p->color, p->cu_type, p->cbf_u, p->lfnst_idx,
                                            p->color == 0 ? p->mts_idx : 0, p->qp_scaled, p->lambda, &p->ctx, rdoq_ws,
                                            workspace_bytes - 2 * qr_coef_bytes(width, height, n), has_coeffs, stream, p->signhide_enable))
      return rc;
  } else if (p->signhide_enable) {
    if (int rc = uvghip_quant_signhide_batch(bitdepth, coef, coeff_out, width, height, n, p->qp_scaled, quant_ts, p->slice_is_intra,
                                             p->lfnst_idx, stream))
      return rc;
  } else {
    if (int rc = (p->lfnst_idx ? uvghip_quant_lfnst_batch : uvghip_quant_batch)(bitdepth, coef, coeff_out, width, height, n, p->qp_scaled,
                                                                                  quant_ts, p->slice_is_intra, stream))
      return rc;
  }
  if (!rdoq || p->signhide_enable) has_coeffs_kernel<<<(n + 3) / 4, 256, 0, st>>>(coeff_out, width * height, n, has_coeffs);
  { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return uvghip_set_error(e__, __func__); }
  // (4) dequantisation, inverse LFNST, inverse transform, reconstruction (:556-597; without coefficients the inverse of
  //     zeros is zero and rec = pred, the copy of :599-609)
  if (int rc = uvghip_dequant_batch(bitdepth, coeff_out, deq, width, height, n, p->qp_scaled, quant_ts, stream)) return rc;
  if (lfnst_tus)
    if (int rc = uvghip_lfnst_batch(1, deq, width, height, lfnst_tus, n, stream)) return rc;
  return uvghip_tu_inverse_batch(bitdepth, p->type_hor, p->type_ver, p->skip_width, p->skip_height, width, height, p->use_trskip,
                                 deq, pred, pred_stride, rec, rec_stride, tus, n, stream);
}

// ---- joint Cb-Cr residual coding: uvg_quant_cbcr_residual (quant-generic.c:241-442) ----
namespace {

// combined residual of the TUs (:268-302): mask = joint_cb_cr * (jccr_sign ? -1 : 1); C division truncates toward zero
template <typename PX>
__global__ void __launch_bounds__(256)
jccr_residual_kernel(const PX *__restrict__ u_orig, const PX *__restrict__ v_orig, int ostride, const PX *__restrict__ u_pred,
                     const PX *__restrict__ v_pred, int pstride, const uvghip_tu_t *__restrict__ tus, int n, int l2w, int l2h, int mask,
                     int16_t *__restrict__ res)
{
  const int wh = 1 << (l2w + l2h);
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n << (l2w + l2h)) return;
  const int b = (int)(e >> (l2w + l2h)), rem = (int)(e & (size_t)(wh - 1)), y = rem >> l2w, x = rem & ((1 << l2w) - 1);
  const size_t po = (size_t)(tus[b].y + y) * ostride + tus[b].x + x, pp = (size_t)(tus[b].y + y) * pstride + tus[b].x + x;
  const int cbx = (int16_t)((int)u_orig[po] - (int)u_pred[pp]), crx = (int16_t)((int)v_orig[po] - (int)v_pred[pp]);
  int c;
  switch (mask) {
    case 2: c = (4 * cbx + 2 * crx) / 5; break;
    case -2: c = (4 * cbx - 2 * crx) / 5; break;
    case 3: c = (cbx + crx) / 2; break;
    case -3: c = (cbx - crx) / 2; break;
    case 1: c = (4 * crx + 2 * cbx) / 5; break;
    default: c = (4 * crx - 2 * cbx) / 5; break;         // -1
  }
  res[e] = (int16_t)c;
}

// both reconstructions from the decoded joint residual (:381-424); early_skip / no coefficients: the predictions (:427-437)
template <typename PX>
__global__ void __launch_bounds__(256)
jccr_recon_kernel(const int16_t *__restrict__ res, const uint8_t *__restrict__ has, const PX *__restrict__ u_pred,
                  const PX *__restrict__ v_pred, int pstride, PX *__restrict__ u_rec, PX *__restrict__ v_rec, int rstride,
                  const uvghip_tu_t *__restrict__ tus, int n, int l2w, int l2h, int mask, int early_skip)
{
  const int wh = 1 << (l2w + l2h);
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n << (l2w + l2h)) return;
  const int b = (int)(e >> (l2w + l2h)), rem = (int)(e & (size_t)(wh - 1)), y = rem >> l2w, x = rem & ((1 << l2w) - 1);
  const size_t pp = (size_t)(tus[b].y + y) * pstride + tus[b].x + x, pr = (size_t)(tus[b].y + y) * rstride + tus[b].x + x;
  const int pu = u_pred[pp], pv = v_pred[pp];
  int ru = pu, rv = pv;
  if (has[b] && !early_skip) {
    const int c = res[e];
    int16_t ur, vr;
    switch (mask) {
      case 2: ur = (int16_t)c; vr = (int16_t)(c >> 1); break;
      case -2: ur = (int16_t)c; vr = (int16_t)(-c >> 1); break;
      case 3: ur = (int16_t)c; vr = (int16_t)c; break;
      case -3: ur = (int16_t)c; vr = (int16_t)(-c); break;
      case 1: ur = (int16_t)(c >> 1); vr = (int16_t)c; break;
      default: ur = (int16_t)(-c >> 1); vr = (int16_t)c; break;   // -1
    }
    ru = clampi((int)(int16_t)(ur + pu), 0, px_traits<PX>::maxv);
    rv = clampi((int)(int16_t)(vr + pv), 0, px_traits<PX>::maxv);
  }
  u_rec[pr] = (PX)ru; v_rec[pr] = (PX)rv;
}

__global__ void __launch_bounds__(256)
jccr_ret_kernel(const uint8_t *__restrict__ has, int n, int joint, uint8_t *__restrict__ ret)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ret[i] = has[i] ? (uint8_t)joint : 0;
}

}  // namespace

extern "C" size_t uvghip_quant_cbcr_residual_workspace_bytes(const uvghip_qr_params_t *p, int n)
{
  if (!p || n <= 0) return 0;
  const bool rdoq = p->rdoq_enable && (p->width > 4 || !p->rdoq_skip);
  return 3 * qr_coef_bytes(p->width, p->height, n) + (((size_t)n + 255) & ~(size_t)255) +
         (rdoq ? (p->signhide_enable ? uvghip_rdoq_signhide_workspace_bytes : uvghip_rdoq_workspace_bytes)(p->width, p->height, n) : 0);
}

extern "C" int uvghip_quant_cbcr_residual_batch(int bitdepth, const uvghip_qr_params_t *p, int joint_cb_cr, int jccr_sign,
                                                const void *u_orig, const void *v_orig, int orig_stride, const void *u_pred,
                                                const void *v_pred, int pred_stride, void *u_rec, void *v_rec, int rec_stride,
                                                const uvghip_tu_t *tus, int n, const uvghip_lfnst_tu_t *lfnst_tus, int16_t *coeff_out,
                                                uint8_t *ret_out, int early_skip, void *workspace, size_t workspace_bytes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!p || joint_cb_cr < 1 || joint_cb_cr > 3 || !u_orig || !v_orig || !u_pred || !v_pred || !u_rec || !v_rec || !coeff_out || !ret_out)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int width = p->width, height = p->height;
  if (!tr_valid_dim(width) || !tr_valid_dim(height) || width < 4 || height < 4 || p->type_hor < 0 || p->type_hor > 2 || p->type_ver < 0 ||
      p->type_ver > 2 || p->skip_width < 0 || p->skip_width >= width || p->skip_height < 0 || p->skip_height >= height)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (int rc = check_qargs(bitdepth, width, height, p->qp_scaled)) return rc;
  if (p->dep_quant) return uvghip_set_error(hipErrorNotSupported, "uvghip_quant_cbcr_residual_batch: dependent quantisation is not built");
  if (n <= 0) return 0;
  if (!workspace || workspace_bytes < uvghip_quant_cbcr_residual_workspace_bytes(p, n))
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_quant_cbcr_residual_batch: workspace");
  const bool rdoq = p->rdoq_enable && (width > 4 || !p->rdoq_skip);
  const int color = joint_cb_cr == 1 ? 2 : 1;                      // the plane whose contexts / QP the joint block is coded with
  const int mask = joint_cb_cr * (jccr_sign ? -1 : 1);
  const int l2w = 31 - __builtin_clz(width), l2h = 31 - __builtin_clz(height);
  hipStream_t st = uvghip_stream(stream);
  char *wsb = static_cast<char *>(workspace);
  const size_t cb = qr_coef_bytes(width, height, n);
  int16_t *res = reinterpret_cast<int16_t *>(wsb), *coef = reinterpret_cast<int16_t *>(wsb + cb), *deq = reinterpret_cast<int16_t *>(wsb + 2 * cb);
  uint8_t *has = reinterpret_cast<uint8_t *>(wsb + 3 * cb);
  void *rdoq_ws = wsb + 3 * cb + (((size_t)n + 255) & ~(size_t)255);
  const size_t total = (size_t)n * width * height;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (bitdepth == 8) jccr_residual_kernel<uint8_t><<<grid, 256, 0, st>>>((const uint8_t *)u_orig, (const uint8_t *)v_orig, orig_stride, (const uint8_t *)u_pred, (const uint8_t *)v_pred, pred_stride, tus, n, l2w, l2h, mask, res);
  else jccr_residual_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t *)u_orig, (const uint16_t *)v_orig, orig_stride, (const uint16_t *)u_pred, (const uint16_t *)v_pred, pred_stride, tus, n, l2w, l2h, mask, res);
  { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return uvghip_set_error(e__, __func__); }
  { hipError_t e = hipGetLastError(); if (e != hipSuccess) return uvghip_set_error(e, __func__); }
  // uvg_transform2d (:305) -> [uvg_fwd_lfnst] -> uvg_rdoq | uvg_quant (:310-340) -> has_coeffs
  if (int rc = uvghip_transform_batch(bitdepth, 0, p->type_hor, p->type_ver, width, height, p->skip_width, p->skip_height, res, coef, n, stream)) return rc;
  if (lfnst_tus)
    if (int rc = uvghip_lfnst_batch(0, coef, width, height, lfnst_tus, n, stream)) return rc;
  if (rdoq) {
    if (int rc = uvghip_rdoq_launch_checked(bitdepth, coef, coeff_out, width, height, n, color, p->cu_type, p->cbf_u, p->lfnst_idx, 0,
                                            p->qp_scaled, p->lambda, &p->ctx, rdoq_ws,
                                            (p->signhide_enable ? uvghip_rdoq_signhide_workspace_bytes : uvghip_rdoq_workspace_bytes)(width, height, n),
                                            has, stream, p->signhide_enable))
      return rc;
  } else if (p->signhide_enable) {
    if (int rc = uvghip_quant_signhide_batch(bitdepth, coef, coeff_out, width, height, n, p->qp_scaled, 0, p->slice_is_intra, p->lfnst_idx, stream))
      return rc;
  } else {
    if (int rc = (p->lfnst_idx ? uvghip_quant_lfnst_batch : uvghip_quant_batch)(bitdepth, coef, coeff_out, width, height, n, p->qp_scaled, 0,
                                                                                  p->slice_is_intra, stream))
      return rc;
  }
  if (!rdoq || p->signhide_enable) has_coeffs_kernel<<<(n + 3) / 4, 256, 0, st>>>(coeff_out, width * height, n, has);
  { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return uvghip_set_error(e__, __func__); }
  // uvg_dequant -> [uvg_inv_lfnst] -> uvg_itransform2d -> both reconstructions (:355-437)
  if (int rc = uvghip_dequant_batch(bitdepth, coeff_out, deq, width, height, n, p->qp_scaled, 0, stream)) return rc;
  if (lfnst_tus)
    if (int rc = uvghip_lfnst_batch(1, deq, width, height, lfnst_tus, n, stream)) return rc;
  if (int rc = uvghip_transform_batch(bitdepth, 1, p->type_hor, p->type_ver, width, height, p->skip_width, p->skip_height, deq, res, n, stream)) return rc;
  if (bitdepth == 8) jccr_recon_kernel<uint8_t><<<grid, 256, 0, st>>>(res, has, (const uint8_t *)u_pred, (const uint8_t *)v_pred, pred_stride, (uint8_t *)u_rec, (uint8_t *)v_rec, rec_stride, tus, n, l2w, l2h, mask, early_skip);
  else jccr_recon_kernel<uint16_t><<<grid, 256, 0, st>>>(res, has, (const uint16_t *)u_pred, (const uint16_t *)v_pred, pred_stride, (uint16_t *)u_rec, (uint16_t *)v_rec, rec_stride, tus, n, l2w, l2h, mask, early_skip);
  { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return uvghip_set_error(e__, __func__); }
  jccr_ret_kernel<<<(n + 255) / 256, 256, 0, st>>>(has, n, joint_cb_cr, ret_out);
  UVGHIP_CHECK_LAUNCH();
}

// =================================================== drop-in strategy layer ====
namespace {

// coeff_abs_sum_func (strategies-quant.h:87): (const coeff_t *coeffs, size_t length)
uint32_t coeff_abs_sum_hip(const int16_t *coeffs, size_t length)
{
  percall_ctx *c = percall_get(length * 2 + 1024);
  const size_t oi = c->take(length * 2), oo = c->take(4);
  memcpy(c->hp<int16_t>(oi), coeffs, length * 2);
  c->upload(oi, length * 2);
  c->must(launch_coeff_cost(c->dp<int16_t>(oi), (int)length, 1, c->dp<uint32_t>(oo), 0, 0, c->stream), "coeff_abs_sum");
  c->download(oo, 4);
  c->sync();
  return *c->hp<uint32_t>(oo);
}
// fast_coeff_cost_func (strategies-quant.h:88): (const coeff_t *coeff, int32_t width, int32_t height, uint64_t weights)
uint32_t fast_coeff_cost_hip(const int16_t *coeff, int32_t width, int32_t height, uint64_t weights)
{
  const size_t length = (size_t)width * height;
  percall_ctx *c = percall_get(length * 2 + 1024);
  const size_t oi = c->take(length * 2), oo = c->take(4);
  memcpy(c->hp<int16_t>(oi), coeff, length * 2);
  c->upload(oi, length * 2);
  c->must(launch_coeff_cost(c->dp<int16_t>(oi), (int)length, 1, c->dp<uint32_t>(oo), 1, weights, c->stream), "fast_coeff_cost");
  c->download(oo, 4);
  c->sync();
  return *c->hp<uint32_t>(oo);
}

}  // namespace

// quant / dequant / quantize_residual / quant_cbcr_residual take encoder_state_t*
// (strategies-quant.h:48-86): they cannot be bound without the encoder's own
// headers, so they are registered by the host-side shim of INTEGRATION.md
// (csrc/shim/strategies-hip-state.c), which extracts plain-value views and calls
// the uvghip_*_percall entry points of csrc/percall_state.hip; those run
// uvghip_quant_batch / uvghip_dequant_batch / uvghip_quantize_residual_batch /
// uvghip_quant_cbcr_residual_batch on one block.  The two state-free functions register here.
extern "C" int uvg_strategy_register_quant_hip(void *opaque, uint8_t bitdepth)
{
  if (!uvghip_ready() && uvghip_init(0) != 0) return 0;
  int ok = 1;
  ok &= uvghip_do_register(opaque, "coeff_abs_sum", (void *)&coeff_abs_sum_hip);
  ok &= uvghip_do_register(opaque, "fast_coeff_cost", (void *)&fast_coeff_cost_hip);
  return ok;
}
