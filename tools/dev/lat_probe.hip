// dev: what one lone wave pays per DEPENDENT operation on gfx950 (s_memtime ticks and the clock ratio are printed; the closed-loop CTU
// search is a dependency chain on one wave, so these latencies -- not throughputs -- are what its time is made of)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define N_ITER 16
#define T0() unsigned long long t0 = __builtin_amdgcn_s_memtime()
#define T1(slot) do { unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[slot] = t1 - t0; } while (0)
__global__ void k(unsigned long long *out, int *sink, const int *gmem)
{
  __shared__ int lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 4) & 4092;
  __syncthreads();
  int v = threadIdx.x, w = 1;
  double d = threadIdx.x, e = 1.0;
  { T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w));) } T1(0); }
  { T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));) } T1(1); }
  { T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v) : "v"(w));) } T1(2); }
  { int a = (threadIdx.x * 4) & 4092; T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a));) } T1(3); v += a; }
  { int a = (threadIdx.x * 4) & 252; T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("ds_bpermute_b32 %0, %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a));) } T1(4); v += a; }
  { int s; T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("v_readlane_b32 %1, %0, 3\n s_nop 0\n v_add_u32 %0, %0, %1" : "+v"(v), "=s"(s));) } T1(5); }
  { T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));) } T1(6); }
  { T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(w) : "vcc");) } T1(7); }
  { unsigned long long p = (unsigned long long)gmem; int a = 0; T0(); for (int i = 0; i < N_ITER; ++i) { REP16(asm volatile("global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)" : "+v"(a) : "s"(p));) } T1(9); v += a; }
  { unsigned long long p = (unsigned long long)gmem; int a = 0; T0(); for (int i = 0; i < N_ITER; ++i) { REP16(asm volatile("s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+s"(a) : "s"(p));) } T1(10); v += a; }
  { unsigned long long m; int s; T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("v_cmp_eq_u32 %1, %0, %0\n s_ff1_i32_b64 %2, %1\n v_readlane_b32 %2, %0, %2\n s_nop 0\n v_add_u32 %0, %0, %2" : "+v"(v), "=s"(m), "=s"(s));) } T1(11); }
  { T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(*(unsigned long long *)&d));) } T1(12); }
  { int a = (threadIdx.x * 4) & 4092; int b; T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("ds_write_b32 %0, %0\n ds_read_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 4092, %1" : "+v"(a), "=v"(b));) } T1(13); v += a; }
  { T0(); for (int i = 0; i < N_ITER; ++i) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %1" : "+v"(v), "+v"(w));) } T1(14); }
  { unsigned long long fp = (unsigned long long)(void *)lds; unsigned hi = (unsigned)(fp >> 32), a = (unsigned)fp + ((threadIdx.x * 4) & 4092);
    T0(); for (int i = 0; i < N_ITER; ++i) { REP64({ unsigned long long ad = ((unsigned long long)hi << 32) | a; asm volatile("flat_load_dword %0, %1\n s_waitcnt vmcnt(0) lgkmcnt(0)" : "=v"(a) : "v"(ad)); }) } T1(8); v += a; }
  { unsigned long long fp = (unsigned long long)(void *)lds; unsigned hi = (unsigned)(fp >> 32), a = (unsigned)fp + ((threadIdx.x * 4) & 4092), b;
    T0(); for (int i = 0; i < N_ITER; ++i) { REP64({ unsigned long long ad = ((unsigned long long)hi << 32) | a; asm volatile("flat_store_dword %1, %2\n flat_load_dword %0, %1\n s_waitcnt vmcnt(0) lgkmcnt(0)" : "=v"(b) : "v"(ad), "v"(a)); a = b; }) } T1(15); v += a; }
  sink[threadIdx.x] = v + (int)d + w;
}
int main()
{
  unsigned long long *d; int *s, *g; hipMalloc(&d, 16 * 8); hipMalloc(&s, 256); hipMalloc(&g, 4096); hipMemset(g, 0, 4096);
  for (int rep = 0; rep < 2; ++rep) { k<<<1, 64>>>(d, s, g); hipDeviceSynchronize(); }
  unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char *nm[16] = {"v_add_u32 (dependent)", "v_add_f64 (dependent)", "v_mul_lo_u32 (dependent)", "ds_read_b32 -> address", "ds_bpermute_b32 -> address", "v_readlane -> s_nop -> v_add", "v_mov_dpp row_shl:1 (dependent)",
                        "v_cmp + v_cndmask", "flat_load_dword (LDS aperture) -> address", "global_load (L2/L1 hit) -> address", "s_load_dword -> offset", "v_cmp -> s_ff1 -> v_readlane -> v_add", "s_memtime + wait", "ds_write + ds_read + wait + v_and", "2 independent v_add", "flat_store + flat_load (LDS aperture) + wait"};
  const int per[16] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 16, 64, 64, 64, 64, 64};
  printf("one wave of 64 lanes alone on a CU; s_memtime ticks (= shader cycles: a CTU of 43 M ticks takes 20.6 ms) per operation group\n");
  for (int i = 0; i < 16; ++i) printf("%-44s %8.2f ticks\n", nm[i], (double)h[i] / (per[i] * N_ITER));
  return 0;
}
