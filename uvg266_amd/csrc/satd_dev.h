// Wave-level SATD building blocks shared by picture.hip, intra.hip and ipol.hip.
// Reference arithmetic: src/strategies/generic/picture-generic.c:118-200 (4x4),
// :256-348 (8x8), tiling src/strategies/strategies-picture.h:54-109.
#pragma once
#include "uvghip_common.h"

// Lane exchanges inside a 4- or 8-lane group as single DPP moves (no LDS crossbar traffic):
// xor 1 / xor 2 are quad permutes, "mirror8" maps lane i of every aligned 8-lane group to lane 7-i.
__device__ __forceinline__ int dpp_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }    // quad_perm [1,0,3,2]
__device__ __forceinline__ int dpp_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ int dpp_mirror8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true); } // row_half_mirror

__device__ __forceinline__ int dpp_mirror16(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true); } // row_mirror

// Sum over an aligned group of 4 or 8 lanes; every lane of the group gets the total.
template <int G> __device__ __forceinline__ int dpp_group_sum(int v)
{
  v += dpp_xor1(v);
  v += dpp_xor2(v);
  if constexpr (G == 8) v += dpp_mirror8(v);
  return v;
}

// Row-per-lane Walsh-Hadamard.  `v` holds one row of N differences; the N
// lanes r = 0..N-1 of an aligned lane group hold the N rows.  Horizontal pass
// in registers, vertical pass by DPP butterflies.  For N = 8 the three vertical
// stages pair lanes by xor 1, xor 2 and the 8-lane mirror (r <-> 7-r = r xor 7):
// {001, 010, 111} is a basis of Z2^3, and with the roles (who keeps the sum, who
// the difference) taken from the coordinates of r in that basis -- bit0^bit2,
// bit1^bit2, bit2 -- the result is the full set of Walsh functions.  The
// coefficient order differs from the reference's butterfly network but the
// multiset of magnitudes is identical and the DC term ends up in lane 0, v[0].
template <int N>
__device__ __forceinline__ void wht_rows(int (&v)[N], int r)
{
#pragma unroll
  for (int half = N / 2; half >= 1; half >>= 1) {
#pragma unroll
    for (int base = 0; base < N; base += 2 * half) {
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const int p = v[base + i], q = v[base + i + half];
        v[base + i] = p + q;
        v[base + i + half] = p - q;
      }
    }
  }
  const int b2 = N == 8 ? (r >> 2) & 1 : 0;
  const bool hi1 = ((r & 1) ^ b2) != 0, hi2 = (((r >> 1) & 1) ^ b2) != 0, hi3 = b2 != 0;
#pragma unroll
  for (int j = 0; j < N; ++j) { const int o = dpp_xor1(v[j]); v[j] = hi1 ? (o - v[j]) : (v[j] + o); }
#pragma unroll
  for (int j = 0; j < N; ++j) { const int o = dpp_xor2(v[j]); v[j] = hi2 ? (o - v[j]) : (v[j] + o); }
  if constexpr (N == 8) {
#pragma unroll
    for (int j = 0; j < N; ++j) { const int o = dpp_mirror8(v[j]); v[j] = hi3 ? (o - v[j]) : (v[j] + o); }
  }
}

// Cost of one 8x8 tile given this lane's row of differences (r = row index).
// Returned value is valid on all 8 lanes of the group.
__device__ __forceinline__ int satd8_cost(int (&d)[8], int r)
{
  wht_rows<8>(d, r);
  int s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += abs(d[j]);
  if (r == 0) s += (abs(d[0]) >> 2) - abs(d[0]);          // DC term counted as |dc|>>2
  s = dpp_group_sum<8>(s);
  return (s + 2) >> 2;                                     // picture-generic.c:345
}
__device__ __forceinline__ int satd4_cost(int (&d)[4], int r)
{
  wht_rows<4>(d, r);
  int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += abs(d[j]);
  if (r == 0) s += (abs(d[0]) >> 2) - abs(d[0]);
  s = dpp_group_sum<4>(s);
  return (s + 1) >> 1;                                     // picture-generic.c:197
}

// Tiling of a bw x bh block per satd_any_size (strategies-picture.h:76-109).
struct satd_tiling {
  int n4c;       // 4x4 tiles in the first column (bw % 8 != 0)
  int n4r;       // 4x4 tiles in the first row of the remainder (bh % 8 != 0)
  int x0, y0;    // origin of the 8x8 area
  int t8x, n8;   // 8x8 tiles per row, total
};
__host__ __device__ inline satd_tiling make_tiling(int bw, int bh)
{
  satd_tiling t;
  t.x0 = t.y0 = 0;
  t.n4c = 0; t.n4r = 0;
  int w = bw, h = bh;
  if (w % 8) { t.n4c = h / 4; t.x0 = 4; w -= 4; }
  if (h % 8) { t.n4r = w / 4; t.y0 = 4; h -= 4; }
  t.t8x = w / 8;
  t.n8 = t.t8x * (h / 8);
  return t;
}

// One block: lanes [0,lpb) of the group cooperate.  A/B loaders return the
// difference row  a - b  for `N` pixels at block-relative (x,y).
template <typename PX, typename LoadDiff>
__device__ __forceinline__ int satd_block(const satd_tiling &t, int l, int lpb, bool active, LoadDiff load_diff)
{
  const int r8 = l & 7, g = l >> 3, G = lpb >> 3;
  int acc = 0;
  // 8x8 tiles: one per 8-lane group per iteration
  const int it8 = (t.n8 + G - 1) / G;
  for (int i = 0; i < it8; ++i) {
    const int tile = i * G + g;
    const bool on = active && tile < t.n8;
    int d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (on) {
      const int ty = tile / t.t8x, tx = tile - ty * t.t8x;
      load_diff(t.x0 + tx * 8, t.y0 + ty * 8 + r8, d);
    }
    const int c = satd8_cost(d, r8);
    if (on && r8 == 0) acc += c;
  }
  // 4x4 tiles: two per 8-lane group per iteration (lanes 0-3 / 4-7)
  const int n4 = t.n4c + t.n4r;
  if (n4) {
    const int r4 = l & 3, sub = (l >> 2) & 1;
    const int it4 = (n4 + 2 * G - 1) / (2 * G);
    for (int i = 0; i < it4; ++i) {
      const int tile = (i * G + g) * 2 + sub;
      const bool on = active && tile < n4;
      int d[4] = {0, 0, 0, 0};
      if (on) {
        int x, y;
        if (tile < t.n4c) { x = 0; y = tile * 4; }            // first column, full height
        else { x = t.x0 + (tile - t.n4c) * 4; y = 0; }         // first row of the remainder
        int tmp[4];
        load_diff(x, y + r4, tmp);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = tmp[k];
      }
      const int c = satd4_cost(d, r4);
      if (on && r4 == 0) acc += c;
    }
  }
  return group_sum(acc, lpb);
}

