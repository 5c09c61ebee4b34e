// Per-call ("drop-in") entry points of the `alf` strategy group (strategies-alf.h:48-109): the four typedefs take encoder_state_t and
// the encoder's alf_classifier / alf_covariance structs, so -- like quant / dequant / quantize_residual -- the typedef-exact functions
// live in the shim compiled inside the encoder tree (csrc/shim/strategies-hip-state.c: field extraction, classifier <-> one byte per
// 4x4 block), and everything that touches samples is here: a call stages the whole plane(s) it refers to on the calling thread's
// stream (the generic functions read a margin around the block out of the encoder's padded planes: the whole-picture kernels clamp at
// the picture's edges instead, which is the same samples -- alf.c:5150-5170 adjust_pixels), runs the batched kernel on the one
// rectangle, and brings the result back.  The parity path: correct, not fast.
#include "uvghip_common.h"
#include "percall.h"
#include <cstring>

namespace {
template <typename T> size_t stage_plane(percall_ctx *c, const void *src, int stride, int w, int h)
{
  return c->stage_block(src, (size_t)stride, w, h, sizeof(T));
}
}  // namespace

// uvg_alf_derive_classification_blk (alf-generic.c:49-288) for the block [x, x + w) x [y, y + h) of the luma plane `rec` (pic_w x pic_h,
// stride in samples): cls_out[(h / 4)][(w / 4)] = class_idx | transpose_idx << 5 of every 4x4 block (uvghip_alf_classify_frame's byte).
extern "C" int uvghip_alf_classify_percall(int bitdepth, const void *rec, int rec_stride, int pic_w, int pic_h, int shift, int x, int y, int w, int h, uint8_t *cls_out)
{
  if ((bitdepth != 8 && bitdepth != 10) || !rec || !cls_out || (x & 3) || (y & 3) || (w & 3) || (h & 3) || w <= 0 || h <= 0 || x + w > pic_w || y + h > pic_h) return -1;
  const size_t b = bitdepth == 8 ? 1 : 2, plane = (size_t)pic_w * pic_h * b;
  const int cw = (pic_w + 3) / 4, chh = (pic_h + 3) / 4;
  percall_ctx *c = percall_get(plane + (size_t)cw * chh + 4096);
  const size_t op = bitdepth == 8 ? stage_plane<uint8_t>(c, rec, rec_stride, pic_w, pic_h) : stage_plane<uint16_t>(c, rec, rec_stride, pic_w, pic_h);
  const size_t oc = c->take((size_t)cw * chh);
  c->upload(op, plane);
  c->must(uvghip_alf_classify_band(bitdepth, c->dp<void>(op), pic_w, pic_w, pic_h, shift, c->dp<uint8_t>(oc), cw, y, y + h, c->stream), "alf classification");
  c->download(oc, (size_t)cw * chh);
  c->sync();
  for (int r = 0; r < h / 4; ++r) memcpy(cls_out + (size_t)r * (w / 4), c->hp<uint8_t>(oc) + (size_t)(y / 4 + r) * cw + x / 4, (size_t)(w / 4));
  return 0;
}

// uvg_alf_filter_7x7_blk / _5x5_blk (alf-generic.c:290-737) for the block [x, x + w) x [y, y + h) of `src` (the pre-ALF copy of the
// plane, pic_w x pic_h): coef / clip = one filter set ([25][13] luma with the per-4x4 class bytes `cls` of the whole plane, rows of
// cls_stride; [7] chroma, cls NULL).  dst_block[h][w] receives the filtered samples.
extern "C" int uvghip_alf_filter_percall(int bitdepth, int is_chroma, const void *src, int src_stride, int pic_w, int pic_h, int x, int y, int w, int h, const int16_t *coef,
                                         const int16_t *clip, const uint8_t *cls, int cls_stride, void *dst_block)
{
  if ((bitdepth != 8 && bitdepth != 10) || !src || !coef || !clip || !dst_block || (!is_chroma && !cls) || (x & 3) || (y & 3) || w <= 0 || h <= 0 || w > 64 || h > 64 ||
      x + w > pic_w || y + h > pic_h)
    return -1;
  const size_t b = bitdepth == 8 ? 1 : 2, plane = (size_t)pic_w * pic_h * b, nco = is_chroma ? 7 : 25 * 13;
  const int cw = (pic_w + 3) / 4, chh = (pic_h + 3) / 4;
  percall_ctx *c = percall_get(2 * plane + (size_t)cw * chh + 4 * nco + 8192);
  const size_t os = bitdepth == 8 ? stage_plane<uint8_t>(c, src, src_stride, pic_w, pic_h) : stage_plane<uint16_t>(c, src, src_stride, pic_w, pic_h);
  const size_t oco = c->take(nco * 2), ocl = c->take(nco * 2), orc = c->take(sizeof(uvghip_rect_t)), osi = c->take(4), ocs = c->take((size_t)cw * chh);
  const size_t od = c->take(plane);
  memcpy(c->hp<int16_t>(oco), coef, nco * 2);
  memcpy(c->hp<int16_t>(ocl), clip, nco * 2);
  *c->hp<uvghip_rect_t>(orc) = uvghip_rect_t{x, y, w, h};
  *c->hp<int32_t>(osi) = 0;
  if (!is_chroma) for (int r = 0; r < chh; ++r) memcpy(c->hp<uint8_t>(ocs) + (size_t)r * cw, cls + (size_t)r * cls_stride, (size_t)cw);
  c->upload(os, od - os);
  c->must(uvghip_alf_filter_batch(bitdepth, c->dp<void>(os), pic_w, c->dp<void>(od), pic_w, pic_w, pic_h, is_chroma, c->dp<uvghip_rect_t>(orc), c->dp<int32_t>(osi), 1,
                                  c->dp<int16_t>(oco), c->dp<int16_t>(ocl), is_chroma ? nullptr : c->dp<uint8_t>(ocs), cw, c->stream), "alf filter");
  c->download(od + ((size_t)y * pic_w) * b, (size_t)h * pic_w * b);
  c->sync();
  for (int r = 0; r < h; ++r) memcpy((char *)dst_block + (size_t)r * w * b, c->hp<char>(od) + ((size_t)(y + r) * pic_w + x) * b, (size_t)w * b);
  return 0;
}

// uvg_alf_get_blk_stats (alf-generic.c:742-999) of the block [x, x + w) x [y, y + h): org / rec the whole planes.  Outputs in
// uvghip_alf_stats_batch's layout for the ONE rectangle: ee[ncls][13][13][4][4] int64, yv[ncls][13][4] int32, pix_acc[ncls] int64
// (ncls = 25 luma with `cls`, 1 chroma) -- the block's own sums; the caller adds them to its alf_covariance.
extern "C" int uvghip_alf_stats_percall(int bitdepth, int is_chroma, const void *org, int org_stride, const void *rec, int rec_stride, int pic_w, int pic_h, int x, int y, int w,
                                        int h, const uint8_t *cls, int cls_stride, int64_t *ee, int32_t *yv, int64_t *pix_acc)
{
  if ((bitdepth != 8 && bitdepth != 10) || !org || !rec || !ee || !yv || !pix_acc || (!is_chroma && !cls) || (x & 3) || (w & 3) || w <= 0 || h <= 0 || w > 64 || h > 64 ||
      x + w > pic_w || y + h > pic_h)
    return -1;
  const size_t b = bitdepth == 8 ? 1 : 2, plane = (size_t)pic_w * pic_h * b, ncls = is_chroma ? 1 : 25;
  const int cw = (pic_w + 3) / 4, chh = (pic_h + 3) / 4;
  const size_t n_ee = ncls * 13 * 13 * 16, n_y = ncls * 13 * 4;
  percall_ctx *c = percall_get(2 * plane + (size_t)cw * chh + n_ee * 8 + n_y * 4 + ncls * 8 + 8192);
  const size_t oo = bitdepth == 8 ? stage_plane<uint8_t>(c, org, org_stride, pic_w, pic_h) : stage_plane<uint16_t>(c, org, org_stride, pic_w, pic_h);
  const size_t orr = bitdepth == 8 ? stage_plane<uint8_t>(c, rec, rec_stride, pic_w, pic_h) : stage_plane<uint16_t>(c, rec, rec_stride, pic_w, pic_h);
  const size_t orc = c->take(sizeof(uvghip_rect_t)), ocs = c->take((size_t)cw * chh);
  const size_t oe = c->take(n_ee * 8), oy = c->take(n_y * 4), opx = c->take(ncls * 8);
  *c->hp<uvghip_rect_t>(orc) = uvghip_rect_t{x, y, w, h};
  if (!is_chroma) for (int r = 0; r < chh; ++r) memcpy(c->hp<uint8_t>(ocs) + (size_t)r * cw, cls + (size_t)r * cls_stride, (size_t)cw);
  c->upload(oo, oe - oo);
  c->must(uvghip_alf_stats_batch(bitdepth, c->dp<void>(oo), pic_w, c->dp<void>(orr), pic_w, pic_w, pic_h, is_chroma, c->dp<uvghip_rect_t>(orc), 1,
                                 is_chroma ? nullptr : c->dp<uint8_t>(ocs), cw, c->dp<int64_t>(oe), c->dp<int32_t>(oy), c->dp<int64_t>(opx), c->stream), "alf statistics");
  c->download(oe, opx + ncls * 8 - oe);
  c->sync();
  memcpy(ee, c->hp<int64_t>(oe), n_ee * 8);
  memcpy(yv, c->hp<int32_t>(oy), n_y * 4);
  memcpy(pix_acc, c->hp<int64_t>(opx), ncls * 8);
  return 0;
}
