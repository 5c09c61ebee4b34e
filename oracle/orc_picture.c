/*
 * oracle/orc_picture.c -- restatement of the "picture" strategy group
 * (SAD / SATD / SSD / residual / bi-pred average) of uvg266's generic-C
 * backend.  TEST INFRASTRUCTURE ONLY (see orc_common.h).
 *
 * Reference followed (all under /root/reference/src):
 *   strategies/generic/picture-generic.c:99-112    reg_sad
 *   strategies/generic/picture-generic.c:118-200   4x4 Hadamard, (satd+1)>>1
 *   strategies/generic/picture-generic.c:256-348   8x8 Hadamard, (sad+2)>>2
 *   strategies/strategies-picture.h:54-109         NxN / any_size tiling
 *   strategies/generic/picture-generic.c:412-477   any_size_quad (incl. quirk)
 *   strategies/generic/picture-generic.c:1052-1113 sad_NxN, sad_NxN_dual
 *   strategies/generic/picture-generic.c:1115-1130 pixels_calc_ssd
 *   strategies/generic/picture-generic.c:1132-1193 bipred averages
 *   strategies/generic/picture-generic.c:1266-1331 ver_sad / hor_sad
 *   strategies/generic/picture-generic.c:1360-1369 generate_residual
 *   image.c:280-296,310-428,438-473                border SAD dispatch
 */
#include "orc_common.h"

/* ---------------------------------------------------------------- SAD -- */

ORC_EXPORT unsigned ORC_FN(reg_sad)(const orc_px *a, const orc_px *b, int w, int h,
                                    unsigned sa, unsigned sb)
{
  unsigned acc = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      acc += (unsigned)orc_iabs((int)a[y * sa + x] - (int)b[y * sb + x]);
  return acc;
}

/* one replicated reference row (picture-generic.c:1266) */
ORC_EXPORT unsigned ORC_FN(ver_sad)(const orc_px *pic, const orc_px *ref_row, int w, int h,
                                    unsigned pic_stride)
{
  unsigned acc = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      acc += (unsigned)orc_iabs((int)pic[y * pic_stride + x] - (int)ref_row[x]);
  return acc;
}

/* one replicated reference column (picture-generic.c:1291, static helper) */
static unsigned col_sad(const orc_px *pic, const orc_px *ref_col, int w, int h,
                        unsigned pic_stride, unsigned ref_stride)
{
  unsigned acc = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      acc += (unsigned)orc_iabs((int)pic[y * pic_stride + x] - (int)ref_col[y * ref_stride]);
  return acc;
}

/* picture-generic.c:1308: left overhang XOR right overhang XOR neither */
ORC_EXPORT unsigned ORC_FN(hor_sad)(const orc_px *pic, const orc_px *ref, int w, int h,
                                    unsigned pic_stride, unsigned ref_stride,
                                    unsigned left, unsigned right)
{
  if (left) {
    return col_sad(pic, ref + left, (int)left, h, pic_stride, ref_stride)
         + ORC_FN(reg_sad)(pic + left, ref + left, w - (int)left, h, pic_stride, ref_stride);
  }
  if (right) {
    return ORC_FN(reg_sad)(pic, ref, w - (int)right, h, pic_stride, ref_stride)
         + col_sad(pic + w - right, ref + w - right - 1, (int)right, h, pic_stride, ref_stride);
  }
  return ORC_FN(reg_sad)(pic, ref, w, h, pic_stride, ref_stride);
}

/* single replicated corner pixel (image.c:280) */
static unsigned corner_sad(const orc_px *pic, const orc_px *ref_px, int w, int h, unsigned pic_stride)
{
  const int r = *ref_px;
  unsigned acc = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      acc += (unsigned)orc_iabs((int)pic[y * pic_stride + x] - r);
  return acc;
}

/*
 * uvg_image_calc_sad (image.c:438) restated on raw planes.  `pic`/`ref` point
 * at pixel (0,0) of each luma plane; ref_w/ref_h are the reference frame's
 * visible dimensions.  The case analysis mirrors image.c:322-427 (the same
 * branch order, so the same overhang combinations are honoured or ignored).
 * Result is shifted by depth-8 like image.c:472.
 */
ORC_EXPORT unsigned ORC_FN(image_calc_sad)(const orc_px *pic, int pic_stride,
                                           const orc_px *ref, int ref_stride,
                                           int ref_w, int ref_h,
                                           int pic_x, int pic_y, int ref_x, int ref_y,
                                           int bw, int bh)
{
  unsigned res;
  if (ref_x >= 0 && ref_x <= ref_w - bw && ref_y >= 0 && ref_y <= ref_h - bh) {
    res = ORC_FN(reg_sad)(pic + pic_y * pic_stride + pic_x, ref + ref_y * ref_stride + ref_x,
                          bw, bh, (unsigned)pic_stride, (unsigned)ref_stride);
    return res >> ORC_DSHIFT;
  }
  if (ref_x > ref_w) ref_x = ref_w;
  if (ref_y > ref_h) ref_y = ref_h;
  if (ref_x + bw < 0) ref_x = -bw;
  if (ref_y + bh < 0) ref_y = -bh;

  const int left   = ref_x < 0 ? -ref_x : 0;
  const int top    = ref_y < 0 ? -ref_y : 0;
  const int right  = ref_x + bw > ref_w ? ref_x + bw - ref_w : 0;
  const int bottom = ref_y + bh > ref_h ? ref_y + bh - ref_h : 0;

  const orc_px *p = pic + pic_y * pic_stride + pic_x;
  /* may point outside the plane; only ever dereferenced after projection */
  const orc_px *r = ref + (ptrdiff_t)ref_y * ref_stride + ref_x;
  const unsigned ps = (unsigned)pic_stride, rs = (unsigned)ref_stride;
  res = 0;

  if (top && left) {
    res += corner_sad(p, r + top * ref_stride + left, left, top, ps);
    res += ORC_FN(ver_sad)(p + left, r + top * ref_stride + left, bw - left, top, ps);
    res += ORC_FN(hor_sad)(p + top * pic_stride, r + top * ref_stride, bw, bh - top, ps, rs,
                           (unsigned)left, (unsigned)right);
  } else if (top && right) {
    res += ORC_FN(ver_sad)(p, r + top * ref_stride, bw - right, top, ps);
    res += corner_sad(p + bw - right, r + top * ref_stride + (bw - right - 1), right, top, ps);
    res += ORC_FN(hor_sad)(p + top * pic_stride, r + top * ref_stride, bw, bh - top, ps, rs,
                           (unsigned)left, (unsigned)right);
  } else if (bottom && left) {
    res += ORC_FN(hor_sad)(p, r, bw, bh - bottom, ps, rs, (unsigned)left, (unsigned)right);
    res += corner_sad(p + (bh - bottom) * pic_stride,
                      r + (bh - bottom - 1) * ref_stride + left, left, bottom, ps);
    res += ORC_FN(ver_sad)(p + (bh - bottom) * pic_stride + left,
                           r + (bh - bottom - 1) * ref_stride + left, bw - left, bottom, ps);
  } else if (bottom && right) {
    res += ORC_FN(hor_sad)(p, r, bw, bh - bottom, ps, rs, (unsigned)left, (unsigned)right);
    res += ORC_FN(ver_sad)(p + (bh - bottom) * pic_stride,
                           r + (bh - bottom - 1) * ref_stride, bw - right, bottom, ps);
    res += corner_sad(p + (bh - bottom) * pic_stride + bw - right,
                      r + (bh - bottom - 1) * ref_stride + bw - right - 1, right, bottom, ps);
  } else if (top) {
    res += ORC_FN(ver_sad)(p, r + top * ref_stride, bw, top, ps);
    res += ORC_FN(reg_sad)(p + top * pic_stride, r + top * ref_stride, bw, bh - top, ps, rs);
  } else if (bottom) {
    res += ORC_FN(reg_sad)(p, r, bw, bh - bottom, ps, rs);
    res += ORC_FN(ver_sad)(p + (bh - bottom) * pic_stride,
                           r + (bh - bottom - 1) * ref_stride, bw, bottom, ps);
  } else if (left | right) {
    res += ORC_FN(hor_sad)(p, r, bw, bh, ps, rs, (unsigned)left, (unsigned)right);
  } else {
    res += ORC_FN(reg_sad)(p, r, bw, bh, ps, rs);
  }
  return res >> ORC_DSHIFT;
}

/* contiguous NxN SAD (picture-generic.c:1052); shifted by depth-8 */
ORC_EXPORT unsigned ORC_FN(sad_nxn)(const orc_px *a, const orc_px *b, int n)
{
  unsigned acc = 0;
  for (int i = 0; i < n * n; ++i) acc += (unsigned)orc_iabs((int)a[i] - (int)b[i]);
  return acc >> ORC_DSHIFT;
}

/* two predictions (stride 32*32 apart, pred_buffer) against one original */
ORC_EXPORT void ORC_FN(sad_nxn_dual)(const orc_px *preds, const orc_px *orig, int n, unsigned *out)
{
  out[0] = ORC_FN(sad_nxn)(preds, orig, n);
  out[1] = ORC_FN(sad_nxn)(preds + 32 * 32, orig, n);
}

/* --------------------------------------------------------------- SATD -- */

/* In-place unnormalised Walsh-Hadamard on n (4 or 8) values with stride st.
 * Output order differs from the reference's butterfly order, but the set of
 * magnitudes is the same and the DC term is index 0 in both. */
static void wht1d(int32_t *v, int n, int st)
{
  for (int half = n >> 1; half >= 1; half >>= 1) {
    for (int base = 0; base < n; base += 2 * half) {
      for (int i = 0; i < half; ++i) {
        const int32_t p = v[(base + i) * st], q = v[(base + i + half) * st];
        v[(base + i) * st] = p + q;
        v[(base + i + half) * st] = p - q;
      }
    }
  }
}

static unsigned satd4_tile(const orc_px *a, int sa, const orc_px *b, int sb)
{
  int32_t d[16];
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) d[y * 4 + x] = (int)a[y * sa + x] - (int)b[y * sb + x];
  for (int y = 0; y < 4; ++y) wht1d(d + 4 * y, 4, 1);
  for (int x = 0; x < 4; ++x) wht1d(d + x, 4, 4);
  int32_t s = 0;
  for (int i = 1; i < 16; ++i) s += orc_iabs(d[i]);
  s += orc_iabs(d[0]) >> 2;            /* picture-generic.c:194-196 */
  return (unsigned)((s + 1) >> 1);
}

static unsigned satd8_tile(const orc_px *a, int sa, const orc_px *b, int sb)
{
  int32_t d[64];
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) d[y * 8 + x] = (int)a[y * sa + x] - (int)b[y * sb + x];
  for (int y = 0; y < 8; ++y) wht1d(d + 8 * y, 8, 1);
  for (int x = 0; x < 8; ++x) wht1d(d + x, 8, 8);
  int32_t s = 0;
  for (int i = 1; i < 64; ++i) s += orc_iabs(d[i]);
  s += orc_iabs(d[0]) >> 2;            /* picture-generic.c:341-343 */
  return (unsigned)((s + 2) >> 2);
}

/* satd_4x4 ... satd_64x64 on contiguous blocks (strategies-picture.h:54-70) */
ORC_EXPORT unsigned ORC_FN(satd_nxn)(const orc_px *a, const orc_px *b, int n)
{
  if (n == 4) return satd4_tile(a, 4, b, 4);   /* no depth shift: picture-generic.c:170 */
  unsigned acc = 0;
  for (int y = 0; y < n; y += 8)
    for (int x = 0; x < n; x += 8) acc += satd8_tile(a + y * n + x, n, b + y * n + x, n);
  return acc >> ORC_DSHIFT;
}

/* orig is the FIRST Hadamard operand in the dual forms (picture-generic.c:389) */
ORC_EXPORT void ORC_FN(satd_nxn_dual)(const orc_px *preds, const orc_px *orig, int n, unsigned *out)
{
  out[0] = ORC_FN(satd_nxn)(orig, preds, n);
  out[1] = ORC_FN(satd_nxn)(orig, preds + 32 * 32, n);
}

/* strategies-picture.h:76-109 */
ORC_EXPORT unsigned ORC_FN(satd_any_size)(int w, int h, const orc_px *a, int sa,
                                          const orc_px *b, int sb)
{
  unsigned acc = 0;
  if (w % 8 != 0) {
    for (int y = 0; y < h; y += 4) acc += satd4_tile(a + y * sa, sa, b + y * sb, sb);
    a += 4; b += 4; w -= 4;
  }
  if (h % 8 != 0) {
    for (int x = 0; x < w; x += 4) acc += satd4_tile(a + x, sa, b + x, sb);
    a += 4 * sa; b += 4 * sb; h -= 4;
  }
  for (int y = 0; y < h; y += 8)
    for (int x = 0; x < w; x += 8)
      acc += satd8_tile(a + y * sa + x, sa, b + y * sb + x, sb);
  return acc >> ORC_DSHIFT;
}

/*
 * satd_any_size_quad (picture-generic.c:412-477): four predictors sharing one
 * original.  Reproduces the reference's indexing exactly, including the quirk
 * that the 8x8 loop starts at y = (h-4) % 8 computed AFTER h -= 4, with
 * absolute row offsets from the un-advanced base pointers (so for h % 8 == 4,
 * h > 4 it re-covers rows 0..h-5).  `valid` is accepted and ignored, as in
 * the reference.
 */
ORC_EXPORT void ORC_FN(satd_any_size_quad)(int w, int h, const orc_px *const preds[4], int ps,
                                           const orc_px *orig, int os, unsigned *costs)
{
  unsigned acc[4] = {0, 0, 0, 0};
  const int wmod = w % 8;
  if (wmod != 0) {
    for (int y = 0; y < h; y += 4)
      for (int k = 0; k < 4; ++k)
        acc[k] += satd4_tile(orig + y * os, os, preds[k] + y * ps, ps);
    w -= 4;
  }
  if (h % 8 != 0) {
    /* note: starts from column 0 of both planes (pred_ptrs_tmp / orig_ptr are
     * re-initialised from the base pointers in the reference) */
    for (int x = 0; x < w; x += 4)
      for (int k = 0; k < 4; ++k)
        acc[k] += satd4_tile(orig + x, os, preds[k] + x, ps);
    h -= 4;
  }
  for (int y = h % 8; y < h; y += 8)
    for (int x = wmod; x < w; x += 8)
      for (int k = 0; k < 4; ++k)
        acc[k] += satd8_tile(orig + y * os + x, os, preds[k] + y * ps + x, ps);
  for (int k = 0; k < 4; ++k) costs[k] = acc[k] >> ORC_DSHIFT;
}

/* ------------------------------------------------ SSD / residual / bipred -- */

ORC_EXPORT unsigned ORC_FN(pixels_calc_ssd)(const orc_px *ref, const orc_px *rec, int ref_stride,
                                            int rec_stride, int w, int h)
{
  int acc = 0;   /* int accumulator like picture-generic.c:1119 */
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int d = (int)ref[y * ref_stride + x] - (int)rec[y * rec_stride + x];
      acc += d * d;
    }
  return (unsigned)(acc >> (2 * ORC_DSHIFT));
}

ORC_EXPORT void ORC_FN(generate_residual)(const orc_px *ref, const orc_px *pred, int16_t *res,
                                          int w, int h, int ref_stride, int pred_stride)
{
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      res[y * w + x] = (int16_t)((int)ref[y * ref_stride + x] - (int)pred[y * pred_stride + x]);
}

/*
 * Bi-prediction average (picture-generic.c:1132-1193).  mode bit0: L0 operand
 * is 14-bit intermediate (int16) instead of pixels; bit1: same for L1.
 * Operands are contiguous w*h; dst has dst_stride.
 */
ORC_EXPORT void ORC_FN(bipred_average)(orc_px *dst, int dst_stride, const void *l0, const void *l1,
                                       int mode, int w, int h)
{
  const int shift = 15 - ORC_BIT_DEPTH, off = 1 << (shift - 1);
  for (int i = 0; i < w * h; ++i) {
    const int16_t s0 = (mode & 1) ? ((const int16_t *)l0)[i]
                                  : (int16_t)(((const orc_px *)l0)[i] << (14 - ORC_BIT_DEPTH));
    const int16_t s1 = (mode & 2) ? ((const int16_t *)l1)[i]
                                  : (int16_t)(((const orc_px *)l1)[i] << (14 - ORC_BIT_DEPTH));
    const int32_t v = ((int32_t)s0 + (int32_t)s1 + off) >> shift;
    dst[(i / w) * dst_stride + (i % w)] = orc_clip_px(v);
  }
}

/* ------------------------------------------------- IBC hash, variance -- */
/* picture-generic.c:1371-1443: byte-wise table-driven CRC-32C (reflected polynomial 0x82F63B78), rows in order,
 * 10-bit samples as two bytes (low first: the reference walks the uvg_pixel buffer as uint8_t on a little-endian
 * host). */
ORC_EXPORT uint32_t ORC_FN(crc32c_nxn)(const orc_px *buf, uint32_t stride, int n)
{
  static uint32_t table[256];
  static int have = 0;
  if (!have) {
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u))); table[i] = c; }
    have = 1;
  }
  uint32_t crc = 0xFFFFFFFFu;
  for (int y = 0; y < n; ++y) {
    const uint8_t *b = (const uint8_t *)(buf + (size_t)y * stride);
    for (int i = 0; i < n * (int)sizeof(orc_px); ++i) crc = (crc >> 8) ^ table[(crc ^ b[i]) & 0xFF];
  }
  return crc ^ 0xFFFFFFFFu;
}

/* picture-generic.c:1334-1357, same accumulation order */
ORC_EXPORT double ORC_FN(pixel_var)(const orc_px *arr, uint32_t len)
{
  double sum = 0, var = 0;
  for (uint32_t i = 0; i < len; ++i) sum += arr[i];
  const double mean = sum / (double)len;
  for (uint32_t i = 0; i < len; ++i) { const double t = (double)arr[i] - mean; var += t * t; }
  return var / len;
}
