/*
 * oracle/orc_ipol.c -- restatement of the "ipol" strategy group on whole
 * reference planes (frame coordinates, edge replication = get_extended_block):
 *   sample_quarterpel_luma(_hi)   strategies/generic/ipol-generic.c:134-211
 *   sample_octpel_chroma(_hi)     :681-758
 *   get_extended_block            :761-810 (coordinates clamped to the picture)
 *   filter_hpel/qpel_blocks_*     :213-679: each of the four candidate blocks a
 *       fractional-ME step produces is the motion-compensated prediction at the
 *       integer MV plus a half-/quarter-sample offset (search_inter.c:1133-1216);
 *       restated here as ipol_sample at that fractional position (tools/refcheck
 *       proves the identity against the reference's four-block functions).
 *   bi-prediction average lives in orc_picture.c.
 * TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"
#include "orc_tables.h"

unsigned ORC_FN(satd_any_size)(int, int, const orc_px *, int, const orc_px *, int);

static inline int px_at(const orc_px *p, int stride, int w, int h, int x, int y)
{
  return p[(size_t)orc_clip3(0, h - 1, y) * stride + orc_clip3(0, w - 1, x)];
}

/*
 * Block of w x h samples whose top-left integer position in the reference
 * plane is (x0,y0), at fractional phase (fx,fy): luma 1/16 units with the
 * 8-tap filter, chroma 1/32 units with the 4-tap filter.
 * hi = 0: pixels (clipped, rounded with 14-bit offset); hi = 1: int16 14-bit
 * intermediates (ipol-generic.c:180-211).  dst element type follows hi.
 */
ORC_EXPORT void ORC_FN(ipol_sample)(const orc_px *ref, int stride, int pic_w, int pic_h, int x0, int y0,
                                    int w, int h, int fx, int fy, int is_chroma, int hi, void *dst, int dst_stride)
{
  const int taps = is_chroma ? 4 : 8, off = is_chroma ? 1 : 3;
  const int8_t *fh = is_chroma ? ORC_T_CHROMA_FILTER + 4 * fx : ORC_T_LUMA_FILTER + 8 * fx;
  const int8_t *fv = is_chroma ? ORC_T_CHROMA_FILTER + 4 * fy : ORC_T_LUMA_FILTER + 8 * fy;
  const int shift1 = ORC_BIT_DEPTH - 8, wp_shift = 14 - ORC_BIT_DEPTH, wp_off = 1 << (wp_shift - 1);
  int16_t *tmp = malloc(sizeof(int16_t) * (size_t)(h + taps - 1) * w);
  for (int y = 0; y < h + taps - 1; ++y)
    for (int x = 0; x < w; ++x) {
      int32_t acc = 0;
      for (int k = 0; k < taps; ++k) acc += fh[k] * px_at(ref, stride, pic_w, pic_h, x0 + x - off + k, y0 + y - off);
      tmp[y * w + x] = (int16_t)(acc >> shift1);
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int32_t acc = 0;
      for (int k = 0; k < taps; ++k) acc += fv[k] * tmp[(y + k) * w + x];
      if (hi) ((int16_t *)dst)[y * dst_stride + x] = (int16_t)(acc >> 6);
      else ((orc_px *)dst)[y * dst_stride + x] = orc_clip_px(((acc >> 6) + wp_off) >> wp_shift);
    }
  free(tmp);
}

/*
 * Fractional-ME costs (search_inter.c:1108-1172): SATD of the current block
 * against the prediction at integer position (rx,ry) displaced by each
 * candidate (dx,dy) given in 1/16 sample units (multiples of 4).
 */
ORC_EXPORT void ORC_FN(frac_satd)(const orc_px *cur, int cur_stride, int cx, int cy,
                                  const orc_px *ref, int ref_stride, int pic_w, int pic_h, int rx, int ry,
                                  int w, int h, const int16_t *cand_xy, int n_cand, uint32_t *costs)
{
  orc_px pred[64 * 64];
  for (int c = 0; c < n_cand; ++c) {
    const int mvx = cand_xy[2 * c], mvy = cand_xy[2 * c + 1];
    ORC_FN(ipol_sample)(ref, ref_stride, pic_w, pic_h, rx + (mvx >> 4), ry + (mvy >> 4), w, h, mvx & 15, mvy & 15,
                        0, 0, pred, w);
    costs[c] = ORC_FN(satd_any_size)(w, h, cur + (size_t)cy * cur_stride + cx, cur_stride, pred, w);
  }
}

/*
 * uvg_get_extended_block_generic / _wraparound_generic (ipol-generic.c:761-883) with the uvg_epol_args fields as plain
 * arguments.  Returns 1 when the block + padding lies inside the picture: the reference then hands out pointers into the
 * source (*ext_off = offset of the padded top-left in src, *ext_s = src_s) and `buf` is not touched.  Otherwise 0:
 * buf = (pad_t + blk_h + pad_b + pad_b_simd) rows of pad_l + blk_w + pad_r samples, edge-replicated (or taken modulo the
 * picture width for the wrap-around variant; rows are clamped in both), the pad_b_simd rows zeroed.
 */
ORC_EXPORT int ORC_FN(get_extended_block)(int wrap, const orc_px *src, int src_w, int src_h, int src_s, int blk_x, int blk_y,
                                          int blk_w, int blk_h, int pad_l, int pad_r, int pad_t, int pad_b, int pad_b_simd,
                                          orc_px *buf, long *ext_off, int *ext_s)
{
  const int min_y = blk_y - pad_t, max_y = blk_y + blk_h + pad_b + pad_b_simd - 1;
  const int min_x = blk_x - pad_l;
  const int max_x = blk_x + blk_w + pad_r - (wrap ? 0 : 1);       /* :765 vs :823: the wrap-around variant's max_x is exclusive */
  const int oob = min_y < 0 || max_y >= src_h || min_x < 0 || max_x >= src_w;
  if (!oob) {
    *ext_off = (long)(blk_y - pad_t) * src_s + (blk_x - pad_l);
    *ext_s = src_s;
    return 1;
  }
  const int es = pad_l + blk_w + pad_r;
  *ext_off = 0;
  *ext_s = es;
  int y;
  for (y = -pad_t; y < blk_h + pad_b; ++y) {
    const orc_px *row = src + (size_t)orc_clip3(0, src_h - 1, blk_y + y) * src_s;
    orc_px *dst = buf + (size_t)(y + pad_t) * es;
    for (int i = 0; i < es; ++i) {
      int x = min_x + i;
      if (wrap) { if (x < 0) x += src_w; else if (x >= src_w) x -= src_w; }     /* :836-855: two memcpy pieces */
      else x = orc_clip3(0, src_w - 1, x);                                     /* :780-795: left / middle / right runs */
      dst[i] = row[x];
    }
  }
  for (int k = 0; k < pad_b_simd; ++k) memset(buf + (size_t)(y + pad_t + k) * es, 0, sizeof(orc_px) * es);
  return 0;
}
