// Developer microbenchmark: what a near-empty 510 x 256 kernel costs on MI355X, and how that changes with
// LDS allocation and an early-exit load chain.  20 back-to-back launches per variant, hipEvent timing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct P8 { int v[8]; };
__global__ void __launch_bounds__(256) k_empty(const P8 *p, int *out) { if (p == nullptr) out[0] = 1; }
__global__ void __launch_bounds__(256) k_load_exit(const P8 *p, int *out) { const P8 q = p[blockIdx.x]; if (q.v[0] == 0) return; out[blockIdx.x] = q.v[1]; }
template <int LDS> __global__ void __launch_bounds__(256) k_lds_exit(const P8 *p, int *out)
{
  __shared__ int s[LDS / 4];
  const P8 q = p[blockIdx.x];
  s[threadIdx.x] = q.v[threadIdx.x & 7];
  if (q.v[0] == 0) return;
  __syncthreads();
  out[blockIdx.x] = s[(threadIdx.x + 1) & 255];
}
__global__ void __launch_bounds__(256) k_dynidx_exit(const P8 *p, int *out)
{
  __shared__ int s[8];
  const P8 q = p[blockIdx.x];
  if (q.v[0] == 0) return;
  if (threadIdx.x < 5) s[threadIdx.x] = q.v[3 + threadIdx.x];     // dynamic index into a register struct
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x & 7];
}
__global__ void __launch_bounds__(256) k_dyn_exit(const P8 *p, int *out)
{
  extern __shared__ int sd[];
  const P8 q = p[blockIdx.x];
  sd[threadIdx.x] = q.v[threadIdx.x & 7];
  if (q.v[0] == 0) return;
  __syncthreads();
  out[blockIdx.x] = sd[(threadIdx.x + 1) & 255];
}
template <typename F> void run(const char *name, F launch)
{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %7.2f us per launch\n", name, ms * 1000 / 20);
}
int main()
{
  P8 *p; int *out; (void)hipMalloc(&p, 1024 * sizeof(P8)); (void)hipMemset(p, 0, 1024 * sizeof(P8)); (void)hipMalloc(&out, 1 << 22);
  run("empty 510x256", [&] { k_empty<<<510, 256>>>(p, out); });
  run("empty 2040x256", [&] { k_empty<<<2040, 256>>>(p, out); });
  run("load+exit", [&] { k_load_exit<<<510, 256>>>(p, out); });
  run("lds 1K load+exit", [&] { k_lds_exit<1024><<<510, 256>>>(p, out); });
  run("lds 8K load+exit", [&] { k_lds_exit<8192><<<510, 256>>>(p, out); });
  run("lds 48K load+exit", [&] { k_lds_exit<49152><<<510, 256>>>(p, out); });
  run("lds 16K load+exit", [&] { k_lds_exit<16384><<<510, 256>>>(p, out); });
  run("lds 32K load+exit", [&] { k_lds_exit<32768><<<510, 256>>>(p, out); });
  run("lds 40K load+exit", [&] { k_lds_exit<40960><<<510, 256>>>(p, out); });
  (void)hipMemset(p, 1, 1024 * sizeof(P8));
  run("lds 1K no exit", [&] { k_lds_exit<1024><<<510, 256>>>(p, out); });
  run("lds 8K no exit", [&] { k_lds_exit<8192><<<510, 256>>>(p, out); });
  run("lds 48K no exit", [&] { k_lds_exit<49152><<<510, 256>>>(p, out); });
  run("load no exit", [&] { k_load_exit<<<510, 256>>>(p, out); });
  (void)hipMemset(p, 0, 1024 * sizeof(P8));
  run("lds 1K exit, 2040 WGs", [&] { k_lds_exit<1024><<<2040, 256>>>(p, out); });
  run("lds 1K exit, 128 WGs", [&] { k_lds_exit<1024><<<128, 256>>>(p, out); });
  for (int kb : {1, 4, 8, 10, 12, 14, 16, 24, 64}) { char nm[64]; snprintf(nm, 64, "dynamic lds %dK exit", kb); run(nm, [&] { k_dyn_exit<<<510, 256, kb * 1024>>>(p, out); }); }
  run("static 1K + dynamic 16K", [&] { k_lds_exit<1024><<<510, 256, 16384>>>(p, out); });
  run("static 8K + dynamic 16K", [&] { k_lds_exit<8192><<<510, 256, 16384>>>(p, out); });
  run("lds 8K exit, 64 threads", [&] { k_lds_exit<8192><<<510, 64>>>(p, out); });
  run("dyn-index struct exit", [&] { k_dynidx_exit<<<510, 256>>>(p, out); });
  return 0;
}
