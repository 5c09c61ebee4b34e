/*
 * Dev-time tool: recover the MIP weight matrices (H.266 8.4.5.2.4) as *responses of the reference*.
 * mip_predict (src/strategies/generic/intra-generic.c:579) is, before clipping and up-sampling,
 *   red[k] = ((sum_j w[k][j] * in[j] + 32 - 32 * sum_j in[j]) >> 6) + in_offset
 * with in[0] = 2^(depth-1) - bdry[0] (size ids 0 and 1; 0 for id 2), in[j] = bdry[j] - bdry[0].  Setting one
 * in[j] to 64 and the rest to 0 gives red[k] = w[k][j] - 32 + bdry[0]; the reduced prediction sits unchanged at
 * the odd positions of an up-sampled block.  Runs against the 10-bit build so that nothing clips.
 * Output: tests/golden/ref_mipmat.bin -> tools/gen_mip_tables.py.  Nothing is transcribed from source text.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "global.h"
#include "strategyselector.h"
#include "intra.h"
#include "strategies/strategies-intra.h"

static uvg_intra_references R;
static uvg_pixel dst[32 * 32];

/* boundary entry j (0..2*rb-1: top then left) covers `f` consecutive samples */
static void set_bdry(int rb, int w, int h, int j, int base, int val_j)
{
  for (int i = 0; i < INTRA_REF_LENGTH; ++i) { R.ref.top[i] = (uvg_pixel)base; R.ref.left[i] = (uvg_pixel)base; }
  if (j < 0) return;
  const int on_top = j < rb, idx = on_top ? j : j - rb, len = on_top ? w : h, f = len / rb;
  uvg_pixel *arr = on_top ? R.ref.top : R.ref.left;
  for (int k = 0; k < f; ++k) arr[1 + idx * f + k] = (uvg_pixel)val_j;     /* samples start at index 1 */
}

int main(void)
{
  if (UVG_BIT_DEPTH != 10) { fprintf(stderr, "build against the 10-bit reference\n"); return 2; }
  if (!uvg_strategyselector_init(0, UVG_BIT_DEPTH)) return 2;
  const int half = 1 << (UVG_BIT_DEPTH - 1);
  static int16_t m0[16][16][4], m1[8][16][8], m2[6][64][8];
  memset(m2, 0, sizeof m2);
  for (int id = 0; id < 3; ++id) {
    const int n = id == 0 ? 4 : (id == 1 ? 8 : 16), rb = id == 0 ? 2 : 4, rp = id < 2 ? 4 : 8, ups = n / rp;
    const int modes = id == 0 ? 16 : (id == 1 ? 8 : 6), in_size = 2 * rb;
    for (int mode = 0; mode < modes; ++mode)
      for (int j = 0; j < in_size; ++j) {
        int b0;
        if (j == 0) {
          if (id == 2) continue;                 /* first matrix column is not used for the large blocks */
          b0 = half - 64; set_bdry(rb, n, n, -1, b0, 0);            /* in[0] = 64, everything else 0 */
        } else { b0 = half; set_bdry(rb, n, n, j, half, half + 64); }
        memset(dst, 0, sizeof dst);
        uvg_mip_predict(&R, (uint16_t)n, (uint16_t)n, dst, mode, false);
        for (int y = 0; y < rp; ++y)
          for (int x = 0; x < rp; ++x) {
            const int v = dst[(y * ups + ups - 1) * n + (x * ups + ups - 1)] - b0 + 32;
            const int k = y * rp + x;
            if (id == 0) m0[mode][k][j] = (int16_t)v; else if (id == 1) m1[mode][k][j] = (int16_t)v; else m2[mode][k][j] = (int16_t)v;
          }
      }
  }
  FILE *f = fopen("tests/golden/ref_mipmat.bin", "wb");
  const uint32_t magic = 0x4d495031;   /* "MIP1" */
  fwrite(&magic, 4, 1, f); fwrite(m0, sizeof m0, 1, f); fwrite(m1, sizeof m1, 1, f); fwrite(m2, sizeof m2, 1, f);
  fclose(f);
  int lo = 1000, hi = -1000;
  for (size_t i = 0; i < sizeof m0 / 2; ++i) { int v = ((int16_t *)m0)[i]; if (v < lo) lo = v; if (v > hi) hi = v; }
  for (size_t i = 0; i < sizeof m1 / 2; ++i) { int v = ((int16_t *)m1)[i]; if (v < lo) lo = v; if (v > hi) hi = v; }
  for (size_t i = 0; i < sizeof m2 / 2; ++i) { int v = ((int16_t *)m2)[i]; if (v < lo) lo = v; if (v > hi) hi = v; }
  printf("wrote tests/golden/ref_mipmat.bin, entries in [%d, %d]\n", lo, hi);
  return 0;
}
