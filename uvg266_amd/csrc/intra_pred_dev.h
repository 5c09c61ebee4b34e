// Intra prediction building blocks shared by intra.hip (batched kernels) and ctu_core.h (the closed-loop CTU search):
// mode geometry (wide-angle mapping, filtered/unfiltered reference choice, PDPC), DC value, one row segment of a prediction.
// Compiles for the device (hipcc) and, for the CPU emulation of the CTU search used by the CPU tests, for the host.
// Reference: src/intra.c:637-753, src/strategies/generic/intra-generic.c:55-437.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#if !defined(__HIPCC__)
#ifndef __device__
#define __device__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
#endif
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
#endif

// ---- constant tables (H.266 8.4.5.2.13: intraPredAngle, invAngle; table 25 fC) -------------
__device__ static const int16_t kSampleDisp[32] = {0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51,
                                                   57, 64, 73, 86, 102, 128, 171, 256, 341, 512, 1024};
__device__ static const int16_t kInvDisp[32] = {0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712,
                                                630, 565, 512, 468, 420, 364, 321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16};
__device__ static const int8_t kPreScale[32] = {8, 7, 6, 5, 5, 4, 4, 4, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 0, 0, 0,
                                                -1, -1, -2, -3};
__device__ static const int8_t kDistThres[8] = {24, 24, 24, 14, 2, 0, 0, 0};
__device__ static const int8_t kCubic[32][4] = {
  {0, 64, 0, 0},    {-1, 63, 2, 0},   {-2, 62, 4, 0},   {-2, 60, 7, -1},  {-2, 58, 10, -2}, {-3, 57, 12, -2},
  {-4, 56, 14, -2}, {-4, 55, 15, -2}, {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3},
  {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4}, {-4, 36, 36, -4}, {-4, 33, 39, -4},
  {-4, 30, 42, -4}, {-4, 29, 44, -5}, {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
  {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3}, {-2, 10, 58, -2}, {-1, 7, 60, -2},
  {0, 4, 62, -2},   {0, 2, 63, -1}};

__device__ __forceinline__ int ilog2_dev(int v) { return 31 - __clz(v); }

// Everything a lane needs to know about one candidate mode of a w x h block.
struct mode_info {
  int16_t mode;         // signalled mode (0 planar, 1 DC, 2..66 angular)
  int16_t pred_mode;    // after wide-angle mapping (-14..80)
  int8_t filtered;      // use the smoothed reference rows
  int8_t vertical;      // pred_mode >= 34: work domain == block domain, else transposed
  int16_t sample_disp;  // intraPredAngle
  int16_t inv_disp;     // invAngle
  int8_t scale;         // PDPC scale of the angular branch
  int8_t use_cubic;
  int8_t pdpc;
  int8_t frac;          // (|sample_disp| & 31) != 0
};

// intra.c:637-658 + :690-726 + intra-generic.c:118-136,206-246
__device__ inline mode_info make_mode_info(int mode, int w, int h, int is_chroma)
{
  const int lw = ilog2_dev(w), lh = ilog2_dev(h);
  mode_info M;
  M.mode = (int16_t)mode;
  int pm = mode;
  if (lw != lh && mode > 1 && mode <= 66) {
    const int shift_tab[6] = {0, 6, 10, 12, 14, 15};
    const int d = abs(lw - lh);
    if (lw > lh && mode < 2 + shift_tab[d]) pm += 65;
    else if (lh > lw && mode > 66 - shift_tab[d]) pm -= 65;
  }
  M.pred_mode = (int16_t)pm;
  M.filtered = 0;
  M.vertical = 1; M.sample_disp = 0; M.inv_disp = 0; M.scale = 0; M.use_cubic = 1; M.pdpc = 0; M.frac = 0;
  if (mode >= 2) {
    const int d50 = abs(pm - 50), d18 = abs(pm - 18);
    const int dist = d50 < d18 ? d50 : d18;
    const int thres = kDistThres[(lw + lh) >> 1];
    const int vertical = pm >= 34;
    const int mode_disp = vertical ? pm - 50 : 18 - pm;
    const int amd = abs(mode_disp);
    const int sd = (mode_disp < 0 ? -1 : 1) * kSampleDisp[amd];
    if (!is_chroma && !(w == 4 && h == 4) && dist > thres) {
      // int_fast8_t truncation of sample_disp in intra_predict_regular (intra.c:711)
      const int sd8 = (int)(int8_t)sd;
      if ((abs(sd8) & 31) == 0) M.filtered = 1;
    }
    M.vertical = (int8_t)vertical;
    M.sample_disp = (int16_t)sd;
    M.inv_disp = kInvDisp[amd];
    const int side_log2 = vertical ? lh : lw;
    int scale = side_log2 - kPreScale[amd];
    if (scale > 2) scale = 2;
    M.scale = (int8_t)scale;
    M.frac = (abs(sd) & 31) != 0;
    M.use_cubic = !(dist > thres && M.frac);
    int pdpc = (w >= 4 && h >= 4);
    if (sd != 0) {
      if (pm > 1 && pm < 67) {
        if (mode_disp < 0) pdpc = 0;
        else if (mode_disp > 0) pdpc = pdpc && scale >= 0;
      }
    }
    M.pdpc = (int8_t)pdpc;
  } else if (mode == 0) {
    M.filtered = !is_chroma && !(w == 4 && h == 4) && (w * h > 32);
    M.pdpc = (w >= 4 && h >= 4);
  } else {
    M.pdpc = (w >= 4 && h >= 4);
  }
  return M;
}

// LDS image of one block's reference rows: [top | left | ftop | fleft], `refn` entries each.
template <typename RP> struct ref_rows_t {       // RP: pointer to const uint16_t (any address space)
  RP top, left, ftop, fleft;
};
typedef ref_rows_t<const uint16_t *> ref_rows;


// intra.c:236-273
template <typename RP> __device__ inline int dc_value(RP top, RP left, int w, int h)
{
  int sum = 0;
  if (w >= h) for (int i = 0; i < w; ++i) sum += top[1 + i];
  if (w <= h) for (int j = 0; j < h; ++j) sum += left[1 + j];
  const int denom = w == h ? w << 1 : max(w, h);
  return (sum + (denom >> 1)) >> ilog2_dev(denom);
}

// NP consecutive predicted samples of one row of the WORK domain (for horizontal
// modes the work domain is the transposed block): row yd, columns xd0..xd0+NP-1.
// wd/hd: work-domain width/height.
template <int NP, typename RR>
__device__ __forceinline__ void predict_row(const mode_info &M, const RR &R, int dc, int is_chroma, int wd, int hd,
                                            int yd, int xd0, int maxv, int (&out)[NP])
{
  const auto top = M.filtered ? R.ftop : R.top;
  const auto left = M.filtered ? R.fleft : R.left;
  if (M.mode < 2) {
    const int lw = ilog2_dev(wd), lh = ilog2_dev(hd);
    const int scale = (lw + lh - 2) >> 2;
    const int sy = (yd << 1) >> scale, wt = 32 >> min(31, sy);
    const int l = left[yd + 1];
    if (M.mode == 0) {
      const int tr = top[wd + 1], bl = left[hd + 1];
      const int offset = 1 << (lw + lh), shift = 1 + lw + lh;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int x = xd0 + i, t = top[x + 1];
        const int hor = (l << lw) + (x + 1) * (tr - l);
        const int ver = (t << lh) + (yd + 1) * (bl - t);
        out[i] = ((hor << lh) + (ver << lw) + offset) >> shift;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) out[i] = dc;
    }
    if (M.pdpc) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int x = xd0 + i, sx = (x << 1) >> scale, wl = 32 >> min(31, sx);
        const int c = out[i];
        out[i] = c + ((wl * (l - c) + wt * ((int)top[x + 1] - c) + 32) >> 6);
      }
    }
    return;
  }
  // angular: main/side in the work domain
  const auto mainr = M.vertical ? top : left;
  const auto side = M.vertical ? left : top;
  const int sd = M.sample_disp;
  if (sd != 0) {
    const int inv = M.inv_disp;
    const int delta = sd * (yd + 1), di = delta >> 5, df = delta & 31;
    auto ref = [&](int idx) -> int {
      if (idx >= 0) return mainr[idx];
      int s = (-idx * inv + 256) >> 9;      // projected side reference (intra-generic.c:156-159)
      return side[min(s, hd)];
    };
    if (M.frac) {
      if (!is_chroma) {
        int f0, f1, f2, f3;
        if (M.use_cubic) { f0 = kCubic[df][0]; f1 = kCubic[df][1]; f2 = kCubic[df][2]; f3 = kCubic[df][3]; }
        else { f0 = 16 - (df >> 1); f1 = 32 - (df >> 1); f2 = 16 + (df >> 1); f3 = df >> 1; }
        int p[NP + 3];
#pragma unroll
        for (int k = 0; k < NP + 3; ++k) p[k] = ref(di + xd0 + k);
#pragma unroll
        for (int i = 0; i < NP; ++i)
          out[i] = clampi((f0 * p[i] + f1 * p[i + 1] + f2 * p[i + 2] + f3 * p[i + 3] + 32) >> 6, 0, maxv);
      } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int r1 = ref(xd0 + i + di + 1), r2 = ref(xd0 + i + di + 2);
          out[i] = r1 + ((df * (r2 - r1) + 16) >> 5);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) out[i] = ref(xd0 + i + di + 1);
    }
    if (M.pdpc) {
      const int scale = M.scale;
      const int lim = min(3 << scale, wd);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int x = xd0 + i;
        if (x < lim) {
          const int inv_sum = 256 + (x + 1) * inv;
          const int wl = 32 >> ((2 * x) >> scale);
          const int l = side[yd + (inv_sum >> 9) + 1];
          out[i] = out[i] + ((wl * (l - out[i]) + 32) >> 6);
        }
      }
    }
  } else {
    const int lw = ilog2_dev(wd), lh = ilog2_dev(hd);
    const int sc = (lw + lh - 2) >> 2;
    const int tl = mainr[0], l = side[1 + yd];
    const int lim = min(3 << sc, wd);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int x = xd0 + i;
      int v = mainr[1 + x];
      if (M.pdpc && x < lim) v = clampi(v + (((32 >> ((2 * x) >> sc)) * (l - tl) + 32) >> 6), 0, maxv);
      out[i] = v;
    }
  }
}

