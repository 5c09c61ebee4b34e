// dev: coeff_bits (wave) vs coeff_bits_serial on random blocks, on the device.  hipcc -O2 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../uvg266_amd/csrc/uvghip_common.h"
#include "../../uvg266_amd/csrc/ctu_core.h"
using namespace ctu;
struct result { double b_wave, b_serial; int bad_model; uint32_t mw, ms; };
__global__ void __launch_bounds__(256) k(const int16_t *blocks, const uint32_t *models, const int *meta, result *out, int nblk)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds<uint8_t> *S = reinterpret_cast<lds<uint8_t> *>(smem);
  build_scans(S);
  for (int i = threadIdx.x; i < 512; i += 256) tab_ebits()[i] = kEntropyBits[i];
  for (int i = threadIdx.x; i < NMODELS; i += 256) tab_rate()[i] = k_ctx_init[3][i];
  __syncthreads();
  for (int b = 0; b < nblk; ++b) {
    const int n = meta[2 * b], color = meta[2 * b + 1];
    for (int i = threadIdx.x; i < n * n; i += 256) S->lv[0][i] = blocks[b * 1024 + i];
    for (int i = threadIdx.x; i < NMODELS; i += 256) { S->cur[i] = models[i]; S->coder[i] = models[i]; }
    __syncthreads();
    double bw = 0, bs = 0;
    if (threadIdx.x < 64) bw = coeff_bits(S, S->cur, 1, S->lv[0], n, color);
    __syncthreads();
    if (threadIdx.x == 0) {
      bs = coeff_bits_serial(S->coder, S->scan + scan_base(ilog2_dev(n)), S->lv[0], n, color);
      int bad = -1;
      for (int i = 0; i < NMODELS; ++i) if (S->cur[i] != S->coder[i]) { bad = i; break; }
      out[b].b_wave = bw; out[b].b_serial = bs; out[b].bad_model = bad; out[b].mw = bad >= 0 ? S->cur[bad] : 0; out[b].ms = bad >= 0 ? S->coder[bad] : 0;
    }
    __syncthreads();
  }
}
int main()
{
  const int nblk = 400;
  std::vector<int16_t> blocks(nblk * 1024, 0);
  std::vector<int> meta(2 * nblk);
  std::vector<uint32_t> models(NMODELS);
  srand(7);
  for (int i = 0; i < NMODELS; ++i) { const uint32_t p = 2000 + rand() % 28000; models[i] = (p & 0x7fe0) | ((p & 0x7ffe) << 16); }
  for (int b = 0; b < nblk; ++b) {
    const int n = 4 << (rand() % 4), color = (b % 3 == 2) ? 1 + (rand() & 1) : 0, style = b % 6;
    meta[2 * b] = n; meta[2 * b + 1] = color;
    for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) {
      const double fall = 1.0 / (1.0 + 0.35 * (x + y));
      const int r = rand();
      int v = 0;
      if (style == 0) v = (r % 7) - 3;
      else if (style == 1) v = (r % 100 < 8) ? ((r >> 8) % 200) - 100 : 0;
      else if (style == 2) v = ((r % 1000) / 1000.0 < fall) ? (((r >> 12) & 1) ? 1 : -1) * (1 + ((r >> 16) % 3)) : 0;
      else if (style == 3) v = (x + y < 3) ? (r % 2001) - 1000 : 0;
      else if (style == 4) v = (r % 100 < 50) ? (((r >> 9) & 1) ? 1 : -1) : 0;
      else v = ((r % 1000) / 1000.0 < 0.3 * fall) ? (((r >> 12) & 1) ? 1 : -1) : 0;
      blocks[b * 1024 + y * n + x] = (int16_t)v;
    }
    if (b % 37 == 5) { for (int i = 0; i < n * n; ++i) blocks[b * 1024 + i] = 0; blocks[b * 1024] = -2; }
  }
  int16_t *dB; uint32_t *dM; int *dMeta; result *dR;
  hipMalloc(&dB, blocks.size() * 2); hipMalloc(&dM, NMODELS * 4); hipMalloc(&dMeta, meta.size() * 4); hipMalloc(&dR, nblk * sizeof(result));
  hipMemcpy(dB, blocks.data(), blocks.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dM, models.data(), NMODELS * 4, hipMemcpyHostToDevice);
  hipMemcpy(dMeta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice);
  const size_t lds_b = sizeof(lds<uint8_t>);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), lds_b, 0, dB, dM, dMeta, dR, nblk);
  std::vector<result> R(nblk);
  hipError_t e = hipMemcpy(R.data(), dR, nblk * sizeof(result), hipMemcpyDeviceToHost);
  printf("hip: %s\n", hipGetErrorString(e));
  int bad = 0;
  for (int b = 0; b < nblk; ++b)
    if (R[b].b_wave != R[b].b_serial || R[b].bad_model >= 0) {
      if (bad++ < 12) printf("block %d n %d color %d style %d: wave %.6f serial %.6f bad model %d (%08x vs %08x)\n", b, meta[2 * b], meta[2 * b + 1], b % 6, R[b].b_wave,
                             R[b].b_serial, R[b].bad_model, R[b].mw, R[b].ms);
    }
  printf("mismatching blocks: %d of %d\n", bad, nblk);
  return 0;
}
