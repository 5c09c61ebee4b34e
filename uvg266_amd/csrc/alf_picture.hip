// The reconstruction half of the per-picture ALF process from the encoder's decisions (src/alf.c:5032-5137 alf_reconstruct,
// :4332-4368 + :2925-2986 the APSs' coded filters -> per-class tables, :5244-5279 the fixed filter sets and clipping values,
// :1726-1775 + :1626-1725 CC-ALF): classification, the luma 7x7 filter by filter set, the chroma 5x5 filter by alternative, the
// cross-component filter of both chroma planes -- one call per picture over the library's own block kernels (alf.hip) plus the
// CC-ALF kernel below.  The DERIVATION of the decisions (alf_encoder, alf_encoder_ctb, derive_cc_alf_filter: the double-precision
// solver and the RD loops) is host work upstream and is not part of this file.
#include "uvghip_common.h"
#include "vvc_alf_tables.h"
#include <cstring>
#include <vector>

namespace {

constexpr int CLASSES = 25, LC = 13, CCF = 7, LN = CLASSES * LC, APS_WORDS = 2 * LN + CLASSES + 2, N_APS = 8, N_FIXED = 16, N_ALT = 8, N_CC = 4, CCW = 8;

// filter_blk_cc_alf (alf.c:1626-1725), 4:2:0: a chroma sample gets a 7-tap high-pass of the luma plane around its position added.
// One thread per chroma sample of a rectangle; luma reads are clamped to the picture as the reference's padded copy would give
// them; the rows used bend at the luma virtual boundary (row 60 of every 64).
template <typename PX>
__global__ void __launch_bounds__(256)
cc_alf_kernel(const PX *__restrict__ luma, int lstride, PX *__restrict__ chroma, int cstride, int pic_w, int pic_h,
              const uvghip_rect_t *__restrict__ rects, const int32_t *__restrict__ filter_idx, const int16_t *__restrict__ coef)
{
  const int fi = filter_idx[blockIdx.x];
  if (fi < 0) return;                        // control idc 0: the CTU keeps its samples (alf.c:1749-1751)
  const uvghip_rect_t R = rects[blockIdx.x];
  __shared__ int sF[CCF];
  if (threadIdx.x < CCF) sF[threadIdx.x] = coef[fi * CCW + threadIdx.x];
  __syncthreads();
  constexpr int maxv = px_traits<PX>::maxv, half = 1 << (px_traits<PX>::depth - 1);
  for (int i = threadIdx.x; i < R.w * R.h; i += blockDim.x) {
    const int yy = i / R.w, xx = i - yy * R.w;
    const int x = R.x + xx, y = R.y + yy, lx = x << 1, ly = y << 1, pos = ly & 63;
    int o1 = 1, o2 = -1, o3 = 2;
    if (pos == 58 || pos == 61) o3 = o1;
    else if (pos == 59 || pos == 60) { o1 = 0; o2 = 0; o3 = 0; }
    auto L = [&](int dx, int dy) { return (int)luma[(size_t)clampi(ly + dy, 0, pic_h - 1) * lstride + clampi(lx + dx, 0, pic_w - 1)]; };
    const int cur = L(0, 0);
    int sum = sF[0] * (L(0, o2) - cur) + sF[1] * (L(-1, 0) - cur) + sF[2] * (L(1, 0) - cur) + sF[3] * (L(-1, o1) - cur) +
              sF[4] * (L(0, o1) - cur) + sF[5] * (L(1, o1) - cur) + sF[6] * (L(0, o3) - cur);
    sum = (sum + 64) >> 7;
    sum = clampi(sum + half, 0, maxv) - half;
    PX *d = chroma + (size_t)y * cstride + x;
    *d = (PX)clampi(sum + (int)*d, 0, maxv);
  }
}

// get_blk_stats_cc_alf / calc_covariance_cc_alf (alf.c:2583-2779), 4:2:0: the CC-ALF covariance of one chroma rectangle (a CTU) -- per
// sample the seven luma tap differences e[k] around (2x, 2y) of the picture before ALF and d = org - rec of the chroma plane after its
// ALF; ee = sum e e^T (symmetric 7 x 7), y = sum e d, pix = sum d d.  HBM-bound and tiny (36 sums over <= 1024 samples): one workgroup
// per rectangle, <= 4 samples per thread in 32-bit partial sums (|e|, |d| < 2^10), 64-bit wave and workgroup reduction.
template <typename PX>
__global__ void __launch_bounds__(256)
cc_alf_stats_kernel(const PX *__restrict__ org, int ostride, const PX *__restrict__ rec_c, int cstride, const PX *__restrict__ luma, int lstride,
                    int pic_w, int pic_h, const uvghip_rect_t *__restrict__ rects, long long *__restrict__ ee, int32_t *__restrict__ yv,
                    long long *__restrict__ pix)
{
  const uvghip_rect_t R = rects[blockIdx.x];
  constexpr int NS = 36;                        // 28 products k <= l, 7 cross terms, the energy
  int acc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) acc[i] = 0;
  const bool last_row = (R.y << 1) + 64 >= pic_h;        // the reference moves the virtual boundary out of reach there (alf.c:2652-2655)
  for (int i = threadIdx.x; i < R.w * R.h; i += blockDim.x) {
    const int yy = i / R.w, xx = i - yy * R.w;
    const int vbd = last_row ? -1000 : (((yy << 1) & 63) - 60);
    int o_m1 = -1, o_p1 = 1, o_p2 = 2;
    if (vbd == -2 || vbd == 1) o_p2 = o_p1;
    else if (vbd == -1 || vbd == 0) { o_m1 = 0; o_p1 = 0; o_p2 = 0; }
    const int lx = (R.x + xx) << 1, ly = (R.y + yy) << 1;
    auto L = [&](int dx, int dy) { return (int)luma[(size_t)clampi(ly + dy, 0, pic_h - 1) * lstride + clampi(lx + dx, 0, pic_w - 1)]; };
    const int c = L(0, 0);
    const int e[7] = {L(0, o_m1) - c, L(-1, 0) - c, L(1, 0) - c, L(-1, o_p1) - c, L(0, o_p1) - c, L(1, o_p1) - c, L(0, o_p2) - c};
    const int d = (int)org[(size_t)(R.y + yy) * ostride + R.x + xx] - (int)rec_c[(size_t)(R.y + yy) * cstride + R.x + xx];
    int at = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
      for (int l = k; l < 7; ++l) acc[at++] += e[k] * e[l];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[28 + k] += e[k] * d;
    acc[35] += d * d;
  }
  __shared__ long long sSum[4][NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    long long v = acc[i];
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sSum[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    const long long v = sSum[0][threadIdx.x] + sSum[1][threadIdx.x] + sSum[2][threadIdx.x] + sSum[3][threadIdx.x];
    const int i = threadIdx.x;
    if (i < 28) {
      int k = 0, base = 0;
      while (i >= base + 7 - k) { base += 7 - k; ++k; }
      const int l = k + (i - base);
      ee[(size_t)blockIdx.x * 49 + k * 7 + l] = v; ee[(size_t)blockIdx.x * 49 + l * 7 + k] = v;
    } else if (i < 35) yv[(size_t)blockIdx.x * 7 + (i - 28)] = (int32_t)v;
    else pix[blockIdx.x] = v;
  }
}

void clip_values(int bitdepth, int16_t v[4])       // alf.c:5248-5260
{
  v[0] = (int16_t)(1 << bitdepth);
  for (int i = 1; i < 4; ++i) v[i] = (int16_t)(1 << (7 - 2 * i + bitdepth - 8));
}

}  // namespace

extern "C" int uvghip_cc_alf_filter_batch(int bitdepth, const void *luma, int luma_stride, void *chroma, int chroma_stride, int pic_w, int pic_h,
                                          const uvghip_rect_t *rects, const int32_t *filter_idx, int n, const int16_t *coef, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (n <= 0) return 0;
  if (!luma || !chroma || !rects || !filter_idx || !coef || pic_w <= 0 || pic_h <= 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) cc_alf_kernel<uint8_t><<<n, 256, 0, st>>>((const uint8_t *)luma, luma_stride, (uint8_t *)chroma, chroma_stride, pic_w, pic_h, rects, filter_idx, coef);
  else cc_alf_kernel<uint16_t><<<n, 256, 0, st>>>((const uint16_t *)luma, luma_stride, (uint16_t *)chroma, chroma_stride, pic_w, pic_h, rects, filter_idx, coef);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_cc_alf_stats_batch(int bitdepth, const void *org, int org_stride, const void *rec, int rec_stride, const void *luma, int luma_stride,
                                         int pic_w, int pic_h, const uvghip_rect_t *rects, int n, int64_t *ee, int32_t *y, int64_t *pix_acc, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (n <= 0) return 0;
  if (!org || !rec || !luma || !rects || !ee || !y || !pix_acc || pic_w <= 0 || pic_h <= 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8)
    cc_alf_stats_kernel<uint8_t><<<n, 256, 0, st>>>((const uint8_t *)org, org_stride, (const uint8_t *)rec, rec_stride, (const uint8_t *)luma, luma_stride, pic_w, pic_h, rects,
                                                    (long long *)ee, y, (long long *)pix_acc);
  else
    cc_alf_stats_kernel<uint16_t><<<n, 256, 0, st>>>((const uint16_t *)org, org_stride, (const uint16_t *)rec, rec_stride, (const uint16_t *)luma, luma_stride, pic_w, pic_h, rects,
                                                     (long long *)ee, y, (long long *)pix_acc);
  UVGHIP_CHECK_LAUNCH();
}

// Host only (no device needed): the filter tables the block filter takes, from what the bitstream carries.
extern "C" int uvghip_alf_expand_tables(int bitdepth, int n_luma_aps, const int16_t *luma_aps, const int16_t *chroma_aps, int16_t *luma_coef, int16_t *luma_clip,
                                        int16_t *chroma_coef, int16_t *chroma_clip)
{
  if ((bitdepth != 8 && bitdepth != 10) || n_luma_aps < 0 || n_luma_aps > N_APS || (n_luma_aps && !luma_aps) || !luma_coef || !luma_clip)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  int16_t cv[4];
  clip_values(bitdepth, cv);
  const int16_t centre = (int16_t)(1 << (bitdepth - 1));
  for (int s = 0; s < N_FIXED; ++s)                       // fixed_filter_set_coeff_dec / clip_default (alf.c:5262-5277)
    for (int cl = 0; cl < CLASSES; ++cl) {
      const int f = VVC_ALF_FIXED_MAP[s][cl];
      int16_t *c = luma_coef + ((size_t)s * CLASSES + cl) * LC, *k = luma_clip + ((size_t)s * CLASSES + cl) * LC;
      for (int i = 0; i < LC - 1; ++i) { c[i] = VVC_ALF_FIXED_COEF[f][i]; k[i] = cv[0]; }
      c[LC - 1] = centre; k[LC - 1] = cv[0];
    }
  for (int a = 0; a < N_APS; ++a) {                       // alf_reconstruct_coeff, luma (alf.c:2964-2984)
    int16_t *c0 = luma_coef + (size_t)(N_FIXED + a) * LN, *k0 = luma_clip + (size_t)(N_FIXED + a) * LN;
    if (a >= n_luma_aps) { memset(c0, 0, sizeof(int16_t) * LN); memset(k0, 0, sizeof(int16_t) * LN); continue; }
    const int16_t *aps = luma_aps + (size_t)a * APS_WORDS, *co = aps, *ki = aps + LN, *map = aps + 2 * LN;
    const int n_filters = aps[2 * LN + CLASSES], non_linear = aps[2 * LN + CLASSES + 1];
    if (n_filters < 1 || n_filters > CLASSES) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_expand_tables: number of signalled luma filters");
    for (int cl = 0; cl < CLASSES; ++cl) {
      const int f = map[cl];
      if (f < 0 || f >= n_filters) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_expand_tables: class -> filter index");
      for (int i = 0; i < LC - 1; ++i) {
        const int ci = non_linear ? ki[f * LC + i] : 0;
        if (ci < 0 || ci > 3) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_expand_tables: clip index");
        c0[cl * LC + i] = co[f * LC + i]; k0[cl * LC + i] = cv[ci];
      }
      c0[cl * LC + LC - 1] = centre; k0[cl * LC + LC - 1] = cv[0];
    }
  }
  if (chroma_aps && chroma_coef && chroma_clip) {         // ... chroma: one 7-entry set per alternative (alf.c:2950-2962)
    const int non_linear = chroma_aps[2 * N_ALT * CCF + 1];
    for (int t = 0; t < N_ALT; ++t) {
      for (int i = 0; i < CCF - 1; ++i) {
        const int ci = non_linear ? chroma_aps[(N_ALT + t) * CCF + i] : 0;
        if (ci < 0 || ci > 3) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_expand_tables: chroma clip index");
        chroma_coef[t * CCF + i] = chroma_aps[t * CCF + i]; chroma_clip[t * CCF + i] = cv[ci];
      }
      chroma_coef[t * CCF + CCF - 1] = centre; chroma_clip[t * CCF + CCF - 1] = cv[0];
    }
  }
  return 0;
}

namespace {
struct ws_layout { size_t cls, rects_y, rects_c, set_y, set_c[2], cc_idx[2], luma_coef, luma_clip, chroma_coef, chroma_clip, cc_coef, total; int cls_stride; };
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
ws_layout layout_of(int w, int h)
{
  const size_t n = (size_t)((w + 63) / 64) * ((h + 63) / 64);
  ws_layout L;
  size_t at = 0;
  auto take = [&](size_t bytes) { const size_t o = at; at = align_up(at + bytes, 256); return o; };
  L.cls_stride = (w + 3) / 4;
  L.cls = take((size_t)L.cls_stride * ((h + 3) / 4));
  L.rects_y = take(n * sizeof(uvghip_rect_t)); L.rects_c = take(n * sizeof(uvghip_rect_t));
  L.set_y = take(n * 4); L.set_c[0] = take(n * 4); L.set_c[1] = take(n * 4); L.cc_idx[0] = take(n * 4); L.cc_idx[1] = take(n * 4);
  L.luma_coef = take(sizeof(int16_t) * (N_FIXED + N_APS) * LN); L.luma_clip = take(sizeof(int16_t) * (N_FIXED + N_APS) * LN);
  L.chroma_coef = take(sizeof(int16_t) * N_ALT * CCF); L.chroma_clip = take(sizeof(int16_t) * N_ALT * CCF);
  L.cc_coef = take(sizeof(int16_t) * 2 * N_CC * CCW);
  L.total = at;
  return L;
}
}  // namespace

extern "C" size_t uvghip_alf_reconstruct_workspace_bytes(int pic_w, int pic_h)
{
  if (pic_w <= 0 || pic_h <= 0) return 0;
  return layout_of(pic_w, pic_h).total;
}

extern "C" int uvghip_alf_reconstruct_picture(int bitdepth, const uvghip_alf_picture_t *p, void *workspace, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!p || !workspace || p->width <= 0 || p->height <= 0 || (p->width & 7) || (p->height & 7) || !p->in_y || !p->in_u || !p->in_v || !p->out_y || !p->out_u ||
      !p->out_v || p->in_stride < p->width || p->out_stride < p->width || p->in_stride_c < p->width / 2 || p->out_stride_c < p->width / 2 || !p->ctu_flags ||
      !p->filter_set_idx || p->n_luma_aps < 0 || p->n_luma_aps > N_APS || (p->n_luma_aps && !p->luma_aps) || p->classification_shift < 8 || p->classification_shift > 20)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int w = p->width, h = p->height, cw = w / 2, ch = h / 2, wc = (w + 63) / 64, hc = (h + 63) / 64, n = wc * hc;
  const size_t b = bitdepth == 8 ? 1 : 2;
  hipStream_t st = uvghip_stream(stream);
  // the picture ALF leaves is the picture it got wherever a CTU is not filtered
  UVGHIP_TRY(hipMemcpy2DAsync(p->out_y, (size_t)p->out_stride * b, p->in_y, (size_t)p->in_stride * b, (size_t)w * b, h, hipMemcpyDeviceToDevice, st));
  UVGHIP_TRY(hipMemcpy2DAsync(p->out_u, (size_t)p->out_stride_c * b, p->in_u, (size_t)p->in_stride_c * b, (size_t)cw * b, ch, hipMemcpyDeviceToDevice, st));
  UVGHIP_TRY(hipMemcpy2DAsync(p->out_v, (size_t)p->out_stride_c * b, p->in_v, (size_t)p->in_stride_c * b, (size_t)cw * b, ch, hipMemcpyDeviceToDevice, st));
  const bool cc_on = p->alf_full && (p->cc_alf_enabled[0] || p->cc_alf_enabled[1]);
  if (!p->slice_enabled[0]) {
    // alf_reconstruct returns at once (alf.c:5035): no chroma ALF either.  CC-ALF without it reads a buffer only alf_reconstruct fills (:5066).
    if (cc_on) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_reconstruct_picture: CC-ALF without luma ALF is undefined upstream");
    return 0;
  }
  if ((p->slice_enabled[1] || p->slice_enabled[2]) && !p->chroma_aps) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_reconstruct_picture: chroma APS");
  if (cc_on && !p->cc_coeff) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_reconstruct_picture: CC-ALF coefficients");
  const ws_layout L = layout_of(w, h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  // host side: rectangles, per-CTU indices, tables -- staged in one buffer, one upload each
  std::vector<uvghip_rect_t> ry(n), rc(n);
  std::vector<int32_t> sy(n), sc[2] = {std::vector<int32_t>(n), std::vector<int32_t>(n)}, ci[2] = {std::vector<int32_t>(n), std::vector<int32_t>(n)};
  for (int k = 0; k < n; ++k) {
    const int x = (k % wc) * 64, y = (k / wc) * 64, bw = x + 64 > w ? w - x : 64, bh = y + 64 > h ? h - y : 64;
    ry[k] = uvghip_rect_t{x, y, bw, bh}; rc[k] = uvghip_rect_t{x / 2, y / 2, bw / 2, bh / 2};
    const int set = p->filter_set_idx[k];
    if (p->ctu_flags[k] && (set < 0 || set >= N_FIXED + p->n_luma_aps)) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_reconstruct_picture: filter set index");
    sy[k] = p->ctu_flags[k] ? set : -1;
    for (int c = 0; c < 2; ++c) {
      const int alt = p->ctu_flags[(3 + c) * n + k], ctl = p->ctu_flags[(5 + c) * n + k];
      if (alt >= N_ALT || ctl > N_CC) return uvghip_set_error(hipErrorInvalidValue, "uvghip_alf_reconstruct_picture: chroma alternative / CC-ALF control");
      sc[c][k] = p->slice_enabled[1 + c] && p->ctu_flags[(1 + c) * n + k] ? alt : -1;
      ci[c][k] = cc_on && p->cc_alf_enabled[c] && ctl ? ctl - 1 : -1;
    }
  }
  std::vector<int16_t> lco((size_t)(N_FIXED + N_APS) * LN), lcl((size_t)(N_FIXED + N_APS) * LN), cco(N_ALT * CCF), ccl(N_ALT * CCF);
  if (int rc2 = uvghip_alf_expand_tables(bitdepth, p->n_luma_aps, p->luma_aps, p->chroma_aps, lco.data(), lcl.data(), cco.data(), ccl.data())) return rc2;
  auto up = [&](size_t off, const void *src, size_t bytes) { return hipMemcpyAsync(ws + off, src, bytes, hipMemcpyHostToDevice, st); };
  UVGHIP_TRY(up(L.rects_y, ry.data(), n * sizeof(uvghip_rect_t))); UVGHIP_TRY(up(L.rects_c, rc.data(), n * sizeof(uvghip_rect_t)));
  UVGHIP_TRY(up(L.set_y, sy.data(), n * 4));
  for (int c = 0; c < 2; ++c) { UVGHIP_TRY(up(L.set_c[c], sc[c].data(), n * 4)); UVGHIP_TRY(up(L.cc_idx[c], ci[c].data(), n * 4)); }
  UVGHIP_TRY(up(L.luma_coef, lco.data(), lco.size() * 2)); UVGHIP_TRY(up(L.luma_clip, lcl.data(), lcl.size() * 2));
  UVGHIP_TRY(up(L.chroma_coef, cco.data(), cco.size() * 2)); UVGHIP_TRY(up(L.chroma_clip, ccl.data(), ccl.size() * 2));
  if (cc_on) UVGHIP_TRY(up(L.cc_coef, p->cc_coeff, sizeof(int16_t) * 2 * N_CC * CCW));
  UVGHIP_TRY(hipStreamSynchronize(st));        // the staging vectors go out of scope (pageable memory: the copies are done by now anyway)
  auto rects = [&](size_t off) { return reinterpret_cast<const uvghip_rect_t *>(ws + off); };
  auto i32 = [&](size_t off) { return reinterpret_cast<const int32_t *>(ws + off); };
  auto i16 = [&](size_t off) { return reinterpret_cast<const int16_t *>(ws + off); };
  if (int rc2 = uvghip_alf_classify_frame(bitdepth, p->in_y, p->in_stride, w, h, p->classification_shift, ws + L.cls, L.cls_stride, stream)) return rc2;
  if (int rc2 = uvghip_alf_filter_batch(bitdepth, p->in_y, p->in_stride, p->out_y, p->out_stride, w, h, 0, rects(L.rects_y), i32(L.set_y), n, i16(L.luma_coef), i16(L.luma_clip),
                                        ws + L.cls, L.cls_stride, stream))
    return rc2;
  for (int c = 0; c < 2; ++c) {
    if (!p->slice_enabled[1 + c]) continue;
    if (int rc2 = uvghip_alf_filter_batch(bitdepth, c ? p->in_v : p->in_u, p->in_stride_c, c ? p->out_v : p->out_u, p->out_stride_c, cw, ch, 1, rects(L.rects_c), i32(L.set_c[c]), n,
                                          i16(L.chroma_coef), i16(L.chroma_clip), nullptr, 0, stream))
      return rc2;
  }
  for (int c = 0; c < 2 && cc_on; ++c) {
    if (!p->cc_alf_enabled[c]) continue;
    if (int rc2 = uvghip_cc_alf_filter_batch(bitdepth, p->in_y, p->in_stride, c ? p->out_v : p->out_u, p->out_stride_c, w, h, rects(L.rects_c), i32(L.cc_idx[c]), n,
                                             i16(L.cc_coef) + c * N_CC * CCW, stream))
      return rc2;
  }
  return 0;
}
