/*
 * Dev-time cross-check: oracle/ (orcN_*) vs the reference's generic strategy.
 * See tools/refcheck/README.md.  Includes reference headers at build time
 * (-I$UVG_REF_ROOT/src); nothing from the reference is copied into this repo.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "strategyselector.h"
#include "image.h"

#include "refcheck.h"

int g_dump = 0;
int g_fail = 0;
FILE *g_out = NULL;

uint64_t g_rng = 0x9E3779B97F4A7C15ull;
uint32_t rnd(void) { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (uint32_t)(g_rng >> 16); }

void rec_begin(const char *name, int narr)
{
  if (!g_out) return;
  uint32_t magic = 0x52454631, nl = (uint32_t)strlen(name), na = (uint32_t)narr;
  fwrite(&magic, 4, 1, g_out); fwrite(&nl, 4, 1, g_out); fwrite(name, 1, nl, g_out); fwrite(&na, 4, 1, g_out);
}
void rec_arr(int code, const void *p, size_t n)
{
  if (!g_out) return;
  static const int sz[] = {1, 2, 2, 4, 4, 8, 8};  /* u8 u16 i16 i32 u32 i64 f64 */
  uint32_t c = (uint32_t)code, nn = (uint32_t)n;
  fwrite(&c, 4, 1, g_out); fwrite(&nn, 4, 1, g_out); fwrite(p, (size_t)sz[code], n, g_out);
}
void open_dump(const char *group)
{
  if (g_out) { fclose(g_out); g_out = NULL; }
  if (!g_dump) return;
  char path[256];
  snprintf(path, sizeof path, "tests/golden/ref_%s_%d.bin", group, UVG_BIT_DEPTH);
  g_out = fopen(path, "wb");
  if (!g_out) { perror(path); exit(2); }
}
void check(const char *what, int ok)
{
  if (!ok) { ++g_fail; fprintf(stderr, "MISMATCH[%d-bit] %s\n", UVG_BIT_DEPTH, what); }
}

void check_picture(void);
void check_dct(void);
void check_quant(void);
void check_rdoq(void);
void check_shim(void);
void check_coeffcost(void);
void check_jccr(void);
void check_signhide(void);
void check_intra(void);
void check_ipol(void);
void check_sao(void);
void check_alf(void);
void check_deblock(void);
void check_lfnst(void);
void check_hashvar(void);
void check_mip(void);
void check_dcfilt(void);

int main(int argc, char **argv)
{
  g_dump = argc > 1 && !strcmp(argv[1], "dump");
  if (!uvg_strategyselector_init(0, UVG_BIT_DEPTH)) { fprintf(stderr, "selector init failed\n"); return 2; }
  check_picture();
#ifdef HAVE_DCT
  check_dct();
#endif
#ifdef HAVE_QUANT
  check_quant();
#endif
#ifdef HAVE_INTRA
  check_intra();
#endif
#ifdef HAVE_IPOL
  check_ipol();
#endif
#ifdef HAVE_SAO
  check_sao();
#endif
#ifdef HAVE_ALF
  check_alf();
#endif
#ifdef HAVE_DEBLOCK
  check_deblock();
#endif
#if defined(HAVE_LFNST) && UVG_BIT_DEPTH == 8   /* depth-independent: the oracle exports it once */
  check_lfnst();
#endif
  check_hashvar();
#ifdef HAVE_INTRA
  check_mip();
  check_dcfilt();
#endif
#ifdef HAVE_QUANT
  check_rdoq();      /* last: the earlier groups keep their random streams, hence their committed goldens */
  check_shim();      /* ... and this one after it */
  check_coeffcost();
  check_jccr();
  check_signhide();
#endif
  if (g_out) fclose(g_out);
  printf("refcheck %d-bit: %s (%d mismatches)\n", UVG_BIT_DEPTH, g_fail ? "FAIL" : "OK", g_fail);
  return g_fail ? 1 : 0;
}

#include "rc_picture.inc"
#ifdef HAVE_DCT
#include "rc_dct.inc"
#endif
#ifdef HAVE_QUANT
#include "rc_quant.inc"
#include "rc_rdoq.inc"
#include "rc_shim.inc"
#include "rc_coeffcost.inc"
#include "rc_jccr.inc"
#include "rc_signhide.inc"
#endif
#ifdef HAVE_INTRA
#include "rc_intra.inc"
#endif
#ifdef HAVE_IPOL
#include "rc_ipol.inc"
#endif
#ifdef HAVE_SAO
#include "rc_sao.inc"
#endif
#ifdef HAVE_ALF
#include "rc_alf.inc"
#endif
#ifdef HAVE_DEBLOCK
#include "rc_deblock.inc"
#endif
#if defined(HAVE_LFNST) && UVG_BIT_DEPTH == 8
#include "rc_lfnst.inc"
#endif
#include "rc_hashvar.inc"
#ifdef HAVE_INTRA
#include "rc_mip.inc"
#include "rc_dcfilt.inc"
#endif
