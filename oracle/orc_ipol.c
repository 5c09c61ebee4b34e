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
