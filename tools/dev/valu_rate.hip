// Developer microbenchmark: issue rate of the integer VALU ops the block kernels lean on (gfx950).
// 4 workgroups of 4 waves per CU = 4 waves per SIMD, 8 independent accumulator chains per wave.  Every result is
// passed through an empty asm with a "+v" constraint, so the compiler can neither fold a chain into a closed form
// (round 1's v_add / v_max / v_mul_lo / v_ashr / v_alignbit / v_pk_add rows were folded: < 1 clk per op is impossible)
// nor drop it.  Two clocks per row: wall time via hipEvents (x nominal 2.4 GHz), and s_memtime ticks counted inside
// the kernel (shader cycles at whatever clock the chip actually sustains under this load) -- their ratio is the
// effective clock.  build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
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
  if constexpr (OP == 16) return __builtin_bit_cast(uint32_t, fmaf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), __builtin_bit_cast(float, c)));
  if constexpr (OP == 17) return (a & b) | c;                       // v_and_or_b32
  if constexpr (OP == 18) return (a << 3) + b;                      // v_lshl_add_u32
  return 0;
}
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t *out, unsigned long long *ticks, uint32_t seed)
{
  uint32_t r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = seed * (i + 1) + threadIdx.x;
  uint32_t b = seed ^ threadIdx.x;
  asm volatile("" : "+v"(b));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      r[i] = op<OP>(r[i], b, r[(i + 1) & 7]);
      asm volatile("" : "+v"(r[i]));                                // opaque: one real instruction per step
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char *name, uint32_t *d, unsigned long long *dt)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  k<OP><<<blocks, 256>>>(d, dt, 12345); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, dt, 12345);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[1024];
  hipMemcpy(h, dt, sizeof h, hipMemcpyDeviceToHost);
  double tk = 0; for (int i = 0; i < blocks; ++i) tk += (double)h[i];
  tk /= blocks;
  // per SIMD: 4 waves x ITER x 8 ops
  const double ops_per_simd = 4.0 * ITER * 8;
  printf("%-16s %8.3f ms  wall x 2.4 GHz = %5.2f clk per wave-op per SIMD | s_memtime: %5.2f ticks per wave-op per SIMD, %.0f ticks in %.3f ms = %.2f GHz\n",
         name, ms, ms * 1e6 / ops_per_simd * 2.4, tk / ops_per_simd, tk, ms, tk / (ms * 1e6));
}
int main()
{
  uint32_t *d; hipMalloc(&d, 256 * 4 * 256 * 4);
  unsigned long long *dt; hipMalloc(&dt, 1024 * 8);
  run<16>("v_fma_f32", d, dt);
  run<0>("v_add_u32", d, dt); run<15>("v_add3_u32", d, dt); run<17>("v_and_or_b32", d, dt); run<18>("v_lshl_add_u32", d, dt);
  run<1>("v_pk_add_u16", d, dt); run<2>("v_dot2_i32_i16", d, dt); run<10>("v_dot2_u32_u16", d, dt);
  run<13>("v_dot4_i32_i8", d, dt); run<3>("v_mad_i32_i24", d, dt); run<4>("v_alignbit", d, dt); run<5>("v_perm", d, dt); run<6>("v_sad_u16", d, dt);
  run<11>("v_sad_u8", d, dt); run<7>("v_pk_mad_u16", d, dt); run<8>("v_pk_max_i16", d, dt); run<9>("v_mul_lo_u32", d, dt); run<12>("v_max_i32", d, dt); run<14>("v_ashrrev", d, dt);
  return 0;
}
