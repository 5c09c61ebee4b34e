// Developer microbenchmark: issue rate of the integer VALU ops the block kernels lean on (gfx950).
// One wave per SIMD-slot, 8 independent accumulator chains per op, wall clock via hipEvents.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int OP> __device__ __forceinline__ uint32_t op(uint32_t a, uint32_t b, uint32_t c)
{
  if constexpr (OP == 0) return a + b;
  if constexpr (OP == 1) return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, a) + __builtin_bit_cast(s2, b));
  if constexpr (OP == 2) return (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b), (int)c, false);
  if constexpr (OP == 3) return (uint32_t)(__mul24((int)a, (int)b) + (int)c);
  if constexpr (OP == 4) return __builtin_amdgcn_alignbit(a, b, 16);
  if constexpr (OP == 5) return __builtin_amdgcn_perm(a, b, 0x05040100u);
  if constexpr (OP == 6) return __builtin_amdgcn_sad_u16(a, b, c);
  if constexpr (OP == 7) { s2 v = __builtin_bit_cast(s2, a); return __builtin_bit_cast(uint32_t, v * (s2){1, -1} + __builtin_bit_cast(s2, b)); }
  if constexpr (OP == 8) return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)));
  if constexpr (OP == 9) return a * b;
  if constexpr (OP == 10) return (uint32_t)__builtin_amdgcn_udot2(__builtin_bit_cast(u2, a), __builtin_bit_cast(u2, b), c, false);
  if constexpr (OP == 11) return __builtin_amdgcn_sad_u8(a, b, c);
  if constexpr (OP == 12) return (uint32_t)max((int)a, (int)b);
  if constexpr (OP == 13) return (uint32_t)__builtin_amdgcn_sdot4((int)a, (int)b, (int)c, false);
  if constexpr (OP == 14) return (uint32_t)((int)a >> 6);
  if constexpr (OP == 15) return a + b + c;
  return 0;
}
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed)
{
  uint32_t r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = seed * (i + 1) + threadIdx.x;
  uint32_t b = seed ^ threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = op<OP>(r[i], b, r[(i + 1) & 7]);
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, uint32_t *d)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  k<OP><<<blocks, 256>>>(d, 12345); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, 12345);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x ITER x 8 ops
  const double ops_per_simd = 4.0 * ITER * 8;
  printf("%-14s %8.3f ms  -> %6.2f ns per wave-op per SIMD (x2.4 GHz = %5.2f clk)\n", name, ms, ms * 1e6 / ops_per_simd, ms * 1e6 / ops_per_simd * 2.4);
}
int main()
{
  uint32_t *d; hipMalloc(&d, 256 * 4 * 256 * 4);
  run<0>("v_add_u32", d); run<15>("v_add3_u32", d); run<1>("v_pk_add_u16", d); run<2>("v_dot2_i32_i16", d); run<10>("v_dot2_u32_u16", d);
  run<13>("v_dot4_i32_i8", d); run<3>("v_mad_i32_i24", d); run<4>("v_alignbit", d); run<5>("v_perm", d); run<6>("v_sad_u16", d);
  run<11>("v_sad_u8", d); run<7>("v_pk_mad_u16", d); run<8>("v_pk_max_i16", d); run<9>("v_mul_lo_u32", d); run<12>("v_max_i32", d); run<14>("v_ashrrev", d);
  return 0;
}
