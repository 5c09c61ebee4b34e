/*
 * orc_mip.c -- matrix-based intra prediction (MIP), restated from
 *   src/strategies/generic/intra-generic.c:441-470  uvg_mip_boundary_downsampling_1D
 *   :472-524  uvg_mip_reduced_pred   (offset term 32 - 32 * sum(in), >> 6, + in_offset, clip to the sample range)
 *   :527-577  uvg_mip_pred_upsampling_1D
 *   :579-727  mip_predict_generic    (size ids, reduced boundary, transposition, up-sampling order)
 * The weights (H.266 8.4.5.2.4) come from orc_mip_tables.h, generated from the reference's measured responses
 * (tools/refcheck/dump_mip.c).  top / left: the unfiltered reference rows with the corner at index 0 (the layout of
 * uvg_intra_ref, src/intra.h:48-51).  TEST INFRASTRUCTURE ONLY.
 */
#include "orc_common.h"
#include "orc_mip_tables.h"

static void mip_down(int *dst, const int *src, int src_len, int dst_len)
{
  if (dst_len < src_len) {
    const int f = src_len / dst_len;
    int lg = 0; while ((1 << (lg + 1)) <= f) ++lg;
    const int rnd = 1 << (lg - 1);
    int si = 0;
    for (int d = 0; d < dst_len; ++d) { int s = 0; for (int k = 0; k < f; ++k) s += src[si++]; dst[d] = (s + rnd) >> lg; }
  } else for (int i = 0; i < dst_len; ++i) dst[i] = src[i];
}

/* intra-generic.c:527-577 */
static void mip_up(int *dst, const int *src, const int *boundary, int size_ups, int size_orth, int src_step, int src_stride,
                   int dst_step, int dst_stride, int boundary_step, int factor)
{
  int lg = 0; while ((1 << (lg + 1)) <= factor) ++lg;
  const int rnd = 1 << (lg - 1);
  const int *src_line = src, *bline = boundary + boundary_step - 1;
  int *dst_line = dst;
  for (int o = 0; o < size_orth; ++o) {
    const int *before = bline, *behind = src_line;
    int *cur = dst_line;
    for (int u = 0; u < size_ups; ++u) {
      int sb = (*before) << lg, sh = 0;
      for (int pos = 1; pos <= factor; ++pos) { sb -= *before; sh += *behind; *cur = (sb + sh + rnd) >> lg; cur += dst_step; }
      before = behind; behind += src_step;
    }
    src_line += src_stride; dst_line += dst_stride; bline += boundary_step;
  }
}

/* out: width*height samples, row-major */
ORC_EXPORT void ORC_FN(mip_predict)(const orc_px *top, const orc_px *left, int width, int height, int mip_mode, int transpose,
                                    orc_px *out)
{
  const int size_id = (width == 4 && height == 4) ? 0 : ((width == 4 || height == 4 || (width == 8 && height == 8)) ? 1 : 2);
  const int rb = size_id == 0 ? 2 : 4, rp = size_id < 2 ? 4 : 8, in_size = 2 * rb;
  const int ups_h = width / rp, ups_v = height / rp;
  int rt[64], rl[64];
  for (int i = 0; i < 64; ++i) { rt[i] = i < width ? top[1 + i] : 0; rl[i] = i < height ? left[1 + i] : 0; }
  int bd[8], bdt[8];
  mip_down(bd, rt, width, rb); mip_down(bd + rb, rl, height, rb);
  for (int i = 0; i < rb; ++i) { bdt[i] = bd[rb + i]; bdt[rb + i] = bd[i]; }      /* transposed: left first */
  const int off = bd[0], off_t = bdt[0];
  const int half = 1 << (ORC_BIT_DEPTH - 1);
  bd[0] = size_id < 2 ? half - off : 0; bdt[0] = size_id < 2 ? half - off_t : 0;
  for (int i = 1; i < in_size; ++i) { bd[i] -= off; bdt[i] -= off_t; }
  const int *in = transpose ? bdt : bd;
  const int in_off = transpose ? off_t : off;
  const uint8_t *M = size_id == 0 ? ORC_MIP0 + (size_t)mip_mode * 16 * 4 : (size_id == 1 ? ORC_MIP1 + (size_t)mip_mode * 16 * 8
                                                                                           : ORC_MIP2 + (size_t)mip_mode * 64 * 8);
  int sum = 0;
  for (int i = 0; i < in_size; ++i) sum += in[i];
  const int offset = 32 - 32 * sum;
  int red[64], tmp[64];
  for (int k = 0; k < rp * rp; ++k) {
    int acc = 0;
    for (int i = 0; i < in_size; ++i) acc += in[i] * M[i * rp * rp + k];     /* tables are input-major: [input][output] */
    int v = ((acc + offset) >> 6) + in_off;
    tmp[k] = v < 0 ? 0 : (v > ORC_PX_MAX ? ORC_PX_MAX : v);
  }
  if (transpose) { for (int y = 0; y < rp; ++y) for (int x = 0; x < rp; ++x) red[y * rp + x] = tmp[x * rp + y]; }
  else memcpy(red, tmp, sizeof(int) * (size_t)rp * rp);

  int result[64 * 64];
  memset(result, 0, sizeof result);
  if (ups_h > 1 || ups_v > 1) {
    const int *ver_src = red;
    int ver_src_step = width;
    if (ups_h > 1) {
      int *hor_dst = result + (ups_v - 1) * width;
      ver_src = hor_dst; ver_src_step *= ups_v;
      mip_up(hor_dst, red, rl, rp, rp, 1, rp, 1, ver_src_step, ups_v, ups_h);
    }
    if (ups_v > 1) mip_up(result, ver_src, rt, rp, width, ver_src_step, 1, width, 1, 1, ups_v);
  } else memcpy(result, red, sizeof(int) * (size_t)rp * rp);
  for (int i = 0; i < width * height; ++i) out[i] = (orc_px)result[i];
}
