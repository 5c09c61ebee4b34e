/*
 * Dev-time tool: recover the LFNST kernels as *responses of the reference*.
 * uvg_fwd_lfnst_NxN (src/transform.c:880) computes out[j] = (sum_i in[i] * M[j][i] + 64) >> 7, so the
 * response to 128 * e_i is exactly M[j][i].  The 4 sets x 2 kernels x (16x48 | 16x16) entries are the
 * normative H.266 tables 8.7.4.3; they are written as a fixture (tests/golden/ref_lfnstmat.bin) from which
 * tools/gen_lfnst_tables.py generates the headers the oracle and the product compile.  Nothing is
 * transcribed from the reference's source text.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include "global.h"
void uvg_fwd_lfnst_NxN(coeff_t *src, coeff_t *dst, const int8_t mode, const int8_t index, const int8_t size, int zero_out_size);
int main(void)
{
  FILE *f = fopen("tests/golden/ref_lfnstmat.bin", "wb");
  if (!f) { perror("ref_lfnstmat.bin"); return 2; }
  int16_t m8[4][2][16][48], m4[4][2][16][16];
  for (int set = 0; set < 4; ++set)
    for (int k = 0; k < 2; ++k) {
      for (int i = 0; i < 48; ++i) {
        coeff_t in[48] = {0}, out[48];
        in[i] = 128;
        uvg_fwd_lfnst_NxN(in, out, (int8_t)set, (int8_t)k, 8, 16);
        for (int j = 0; j < 16; ++j) m8[set][k][j][i] = out[j];
      }
      for (int i = 0; i < 16; ++i) {
        coeff_t in[16] = {0}, out[16];
        in[i] = 128;
        uvg_fwd_lfnst_NxN(in, out, (int8_t)set, (int8_t)k, 4, 16);
        for (int j = 0; j < 16; ++j) m4[set][k][j][i] = out[j];
      }
    }
  const uint32_t magic = 0x4c464e31;   /* "LFN1" */
  fwrite(&magic, 4, 1, f);
  fwrite(m8, sizeof m8, 1, f);
  fwrite(m4, sizeof m4, 1, f);
  fclose(f);
  printf("wrote tests/golden/ref_lfnstmat.bin (%zu bytes)\n", 4 + sizeof m8 + sizeof m4);
  return 0;
}
