#ifndef REFCHECK_H_
#define REFCHECK_H_
#include <stdint.h>
#include <stdio.h>
#if UVG_BIT_DEPTH == 8
#define ORC(n) orc8_##n
#else
#define ORC(n) orc10_##n
#endif
enum { A_U8 = 0, A_U16 = 1, A_I16 = 2, A_I32 = 3, A_U32 = 4, A_I64 = 5, A_F64 = 6 };
#define A_PX (UVG_BIT_DEPTH == 8 ? A_U8 : A_U16)
extern int g_dump, g_fail;
uint32_t rnd(void);
void rec_begin(const char *name, int narr);
void rec_arr(int code, const void *p, size_t n);
void open_dump(const char *group);
void check(const char *what, int ok);
static inline void fill_px(uvg_pixel *p, int n, int mode)
{
  /* mode 0: uniform random; 1: smooth + small noise; 2: extremes */
  for (int i = 0; i < n; ++i) {
    if (mode == 0) p[i] = rnd() & PIXEL_MAX;
    else if (mode == 1) p[i] = (uvg_pixel)((PIXEL_MAX / 2) + (int)(rnd() % 17) - 8 + ((i * 3) % 23));
    else p[i] = (rnd() & 1) ? PIXEL_MAX : 0;
  }
}
#endif
