// Tile-per-lane SATD: one lane holds a whole 8x8 (or 4x4) tile of differences as packed
// 16-bit pairs and runs the 2-D Walsh-Hadamard entirely in its own registers with the
// packed VOP3P integer ops (v_pk_add/sub/mad/max/min_i16, v_dot2_u32_u16) -- two
// coefficients per instruction, no cross-lane traffic.
//
// Reference arithmetic: src/strategies/generic/picture-generic.c:118-200 (4x4), :256-348 (8x8).
//
// Range: differences are at most 10 bits + sign; five butterfly stages give |v| <= 32 * 1023 = 32736, which fits int16
// at both bit depths.  The sixth stage is never materialised.  Its absolute sum |a+b| + |a-b| comes from v_sad_u16,
// which needs unsigned operands: flipping bit 15 of ONE input (x ^ 0x8000 = x + 0x8000 mod 2^16) offsets every
// butterfly output that contains that input by +-0x8000 = 0x8000 (mod 2^16), because each output is a +-1 combination
// in which the input appears exactly once.  After five stages a value combines the rows of one parity and all columns,
// so biasing d[0][0].lo and d[1][0].lo biases everything; then, per half,
//   |a - b| = sad_u16(a', b'),   |a + b| = sad_u16(a', (0 - b')),      (0 - b') = -b + 0x8000 (mod 2^16)
// three instructions per register pair including the 32-bit accumulation (was: max, min, neg, max, dot2).
#pragma once
#include "uvghip_common.h"

typedef short pk_s16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_s16, a) + __builtin_bit_cast(pk_s16, b));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_s16, a) - __builtin_bit_cast(pk_s16, b));
}
// (lo, hi) -> (lo + hi, lo - hi): a single v_pk_mad_u16 x, (1,-1), x.swapped
__device__ __forceinline__ uint32_t pk_bfly_pair(uint32_t x)
{
  const pk_s16 v = __builtin_bit_cast(pk_s16, x);
  return __builtin_bit_cast(uint32_t, v * (pk_s16){1, -1} + v.yx);
}
__device__ __forceinline__ void pk_bfly(uint32_t &a, uint32_t &b)
{
  const uint32_t s = pk_add(a, b), t = pk_sub(a, b);
  a = s; b = t;
}
// acc + |a+b| + |a-b| over both halves of offset operands (see the header comment)
__device__ __forceinline__ uint32_t pk_last_stage_acc(uint32_t a, uint32_t b, uint32_t acc)
{
  acc = __builtin_amdgcn_sad_u16(a, b, acc);
  return __builtin_amdgcn_sad_u16(a, pk_sub(0u, b), acc);
}
__device__ __forceinline__ int pk_lo_s16(uint32_t x) { return (int)(int16_t)(x & 0xffffu); }

// d[r][c]: row r of the tile, columns 2c (low half) and 2c+1 (high half).  Destroys d.
// Returns the tile's SATD exactly as the reference's satd_8x8_subblock: sum of |coef| with the DC
// term counted as |DC| >> 2, then (sum + 2) >> 2.
__device__ __forceinline__ uint32_t satd8_tile_lane(uint32_t (&d)[8][4])
{
  d[0][0] ^= 0x8000u; d[1][0] ^= 0x8000u;                            // unsigned offset, see the header comment
#pragma unroll
  for (int r = 0; r < 8; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) d[r][c] = pk_bfly_pair(d[r][c]);     // columns 2c / 2c+1
    pk_bfly(d[r][0], d[r][2]); pk_bfly(d[r][1], d[r][3]);            // column pairs c / c+2
    pk_bfly(d[r][0], d[r][1]); pk_bfly(d[r][2], d[r][3]);            // column pairs c / c+1
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pk_bfly(d[r][c], d[r + 4][c]);       // rows r / r+4
    pk_bfly(d[0][c], d[2][c]); pk_bfly(d[1][c], d[3][c]);            // rows r / r+2
    pk_bfly(d[4][c], d[6][c]); pk_bfly(d[5][c], d[7][c]);
  }
  // DC = a + b of the (row 0, row 1) pair in column 0, both carrying the 0x8000 offset
  const int dc = (int)(d[0][0] & 0xffffu) + (int)(d[1][0] & 0xffffu) - 0x10000;
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 8; r += 2) acc = pk_last_stage_acc(d[r][c], d[r + 1][c], acc);   // rows r / r+1
  const uint32_t adc = (uint32_t)abs(dc);
  const uint32_t sum = acc - adc + (adc >> 2);
  return (sum + 2) >> 2;
}

// 4x4: d[r][c], c = 0..1.  (sum + 1) >> 1 (picture-generic.c:197).
__device__ __forceinline__ uint32_t satd4_tile_lane(uint32_t (&d)[4][2])
{
  d[0][0] ^= 0x8000u; d[1][0] ^= 0x8000u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    d[r][0] = pk_bfly_pair(d[r][0]); d[r][1] = pk_bfly_pair(d[r][1]);
    pk_bfly(d[r][0], d[r][1]);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) { pk_bfly(d[0][c], d[2][c]); pk_bfly(d[1][c], d[3][c]); }
  const int dc = (int)(d[0][0] & 0xffffu) + (int)(d[1][0] & 0xffffu) - 0x10000;
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    acc = pk_last_stage_acc(d[0][c], d[1][c], acc);
    acc = pk_last_stage_acc(d[2][c], d[3][c], acc);
  }
  const uint32_t adc = (uint32_t)abs(dc);
  const uint32_t sum = acc - adc + (adc >> 2);
  return (sum + 1) >> 1;
}
