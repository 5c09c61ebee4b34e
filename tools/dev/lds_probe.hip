// Developer microbenchmark: cost of LDS read patterns on gfx950 (cycles per wave-instruction per CU, four workgroups
// of 4 waves per CU, eight independent loads in flight per wave).  Patterns mirror what the intra search kernel
// does: same-address b64 reads from a few distinct table entries, dword-pair window reads at per-lane offsets, b128
// row reads with different lane strides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
enum { B32, B64, B128, R2 };
template <int KIND> __global__ void __launch_bounds__(256) k(const int *offs, uint32_t *out)
{
  __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 2654435761u;
  __syncthreads();
  const int o = offs[threadIdx.x & 63];      // dword offset of this lane
  uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < ITER; ++it) {
    asm volatile("" ::: "memory");           // the loads must be re-issued every iteration
#pragma unroll
    for (int u = 0; u < 8; ++u) {            // eight independent loads in flight; the pattern is the same mod 64 banks
      const int base = u * 64 + (it & 3) * 512;
      if constexpr (KIND == B32) { acc[u] += lds[base + o]; }
      if constexpr (KIND == B64) { const uint2 v = reinterpret_cast<const uint2 *>(lds)[(base >> 1) + (o >> 1)]; acc[u] += v.x ^ v.y; }   // o even
      if constexpr (KIND == B128) { const uint4 v = reinterpret_cast<const uint4 *>(lds)[(base >> 2) + (o >> 2)]; acc[u] += v.x ^ v.y ^ v.z ^ v.w; }   // o multiple of 4
      if constexpr (KIND == R2) { acc[u] += lds[base + o] ^ lds[base + o + 1]; }
    }
  }
  uint32_t t = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) t ^= acc[u];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int KIND> float run_quiet(const int *h, int *d_offs, uint32_t *d_out)
{
  hipMemcpy(d_offs, h, 64 * sizeof(int), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<1024, 256>>>(d_offs, d_out); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<1024, 256>>>(d_offs, d_out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e6f * 2.4f / (16.0f * ITER * 8);
}
template <int KIND> void run(const char *name, const int *h, int *d_offs, uint32_t *d_out)
{
  hipMemcpy(d_offs, h, 64 * sizeof(int), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<1024, 256>>>(d_offs, d_out); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<1024, 256>>>(d_offs, d_out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.3f ms  %6.1f clk per wave-instruction per CU (2.4 GHz)\n", name, ms, ms * 1e6 * 2.4 / (16.0 * ITER * 8));
}
int main()
{
  int *d_offs; uint32_t *d_out; hipMalloc(&d_offs, 256); hipMalloc(&d_out, 1024 * 256 * 4);
  int h[64];
  for (int l = 0; l < 64; ++l) h[l] = l;             run<B32>("b32 consecutive dwords", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 0;             run<B32>("b32 all lanes same dword", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 64 * (l & 1);  run<B32>("b32 two addresses, same bank", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 2 * l;         run<B64>("b64 consecutive", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 0;             run<B64>("b64 all lanes same entry", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 2 * ((l >> 1) & 1);  run<B64>("b64 two entries (lane pairs alternate)", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 2 * ((l >> 2) & 1) * 5;  run<B64>("b64 two entries 0 / 5", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 2 * (l & 3) * 3;  run<B64>("b64 four entries 0 / 3 / 6 / 9", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 2 * (l & 31);  run<B64>("b64 32 entries, each read by two lanes", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 2 * ((l * 7) & 31);  run<B64>("b64 32 entries permuted", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = l;             run<R2>("read2 (o, o+1), o = lane", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 2 * l;         run<R2>("read2 (o, o+1), o = 2 * lane", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 0;             run<R2>("read2 all lanes same pair", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 1);       run<R2>("read2 o = lane & 1 (windows one dword apart)", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 8 * (l & 1) + (l >> 1 & 1) * 3;  run<R2>("read2 o in {0, 8, 3, 11}", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l >> 2) * 145 + 8 * (l & 1) + ((l >> 1) & 1) * 3;  run<R2>("read2 16 blocks x {0, 8, 3, 11}, stride 145", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l >> 2) * 145 + 8 * (l & 1) + ((l >> 1) & 1) * 8;  run<R2>("read2 16 blocks x {0, 8, 8, 16}, stride 145", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l >> 2) * 145;  run<R2>("read2 16 blocks, 4 lanes each same window, stride 145", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = l * 81;        run<R2>("read2 64 blocks stride 81", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 4 * l;         run<B128>("b128 consecutive", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 68 * l % 4096; run<B128>("b128 lane stride 68 dwords (8x8 blocks)", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l >> 2) * 304 + ((l >> 1) & 1) * 72 + (l & 1) * 4;  run<B128>("b128 16x16 layout (OS 304, band 72)", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l >> 2) * 260 + ((l >> 1) & 1) * 64 + (l & 1) * 4;  run<B128>("b128 old 16x16 layout (OS 260, band 64)", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 0;             run<B128>("b128 all lanes same quad", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l >> 4) * 1156 + ((l >> 2) & 3) * 144 + (l & 3) * 4;  run<B128>("b128 32x32 layout (OS 1156, band 144)", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l >> 4) * 1028 + ((l >> 2) & 3) * 128 + (l & 3) * 4;  run<B128>("b128 old 32x32 layout (OS 1028, band 128)", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 22 * l;        run<B64>("b64 4x4 layout, lane stride 22 dwords", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = 20 * l;        run<B64>("b64 old 4x4 layout, lane stride 20 dwords", h, d_offs, d_out);
  // which lanes share a b128 pass, and how many banks are there?
  for (int l = 0; l < 64; ++l) h[l] = (l & 15) * 4 + (l >> 4) * 256;  run<B128>("b128 P2: 16 lanes distinct, groups 256 dwords apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 15) * 4 + (l >> 4) * 64;   run<B128>("b128 P2b: 16 lanes distinct, groups 64 dwords apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 7) * 4 + (l >> 3) * 256;   run<B128>("b128 P3: 8 lanes distinct, groups 256 dwords apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 7) * 4 + (l >> 3) * 32;    run<B128>("b128 P3b: 8 lanes distinct, groups 32 dwords apart (= consecutive)", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 7) * 4 + ((l >> 3) & 1) * 32 + (l >> 4) * 256;  run<B128>("b128 P4: 16 lanes cover 64 dwords, groups 256 apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 31) * 4 + (l >> 5) * 256;  run<B128>("b128 P5: 32 lanes cover 128 dwords, halves 256 apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 31) * 4 + (l >> 5) * 64;   run<B128>("b128 P5b: 32 lanes cover 128 dwords, halves 64 apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 15) * 4 + ((l >> 4) & 1) * 64 + (l >> 5) * 256;  run<B128>("b128 P6", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 31) * 2 + (l >> 5) * 256;  run<B64>("b64 Q1: 32 lanes distinct, halves 256 apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 31) * 2 + (l >> 5) * 64;   run<B64>("b64 Q1b: halves 64 apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 15) * 2 + (l >> 4) * 256;  run<B64>("b64 Q2: 16 lanes distinct, groups 256 apart", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = l + 64 * (l & 1);               run<B32>("b32 R1: odd lanes 64 dwords up", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 31) + 64 * (l >> 5);       run<B32>("b32 R2: halves on the same 32 banks mod 64", h, d_offs, d_out);
  for (int l = 0; l < 64; ++l) h[l] = (l & 31) + 128 * (l >> 5);      run<B32>("b32 R3: halves 128 dwords apart", h, d_offs, d_out);
  // sweep of the 32x32 original-tile layout: lane = block * 16 + tile_row * 4 + tile_column, b128 per lane
  printf("32x32 layout sweep (band dwords, block dwords -> clk):\n");
  for (int band = 128; band <= 176; band += 4)
    for (int os = 8 * band; os < 8 * band + 132 && 3 * os + 3 * band + 16 + 2048 < 8192; os += 4) {
      for (int l = 0; l < 64; ++l) h[l] = (l >> 4) * os + ((l >> 2) & 3) * band + (l & 3) * 4;
      const float c = run_quiet<B128>(h, d_offs, d_out);
      if (c < 5.5f) printf("  band %d os %d: %.1f\n", band, os, c);
    }
  printf("16x16 layout sweep (band dwords, block dwords -> clk):\n");
  for (int band = 64; band <= 96; band += 4)
    for (int os = 4 * band; os < 4 * band + 68; os += 4) {
      for (int l = 0; l < 64; ++l) h[l] = (l >> 2) * os + ((l >> 1) & 1) * band + (l & 1) * 4;
      if (15 * os + band + 8 + 2048 >= 8192) continue;
      const float c = run_quiet<B128>(h, d_offs, d_out);
      if (c < 5.5f) printf("  band %d os %d: %.1f\n", band, os, c);
    }
  return 0;
}
