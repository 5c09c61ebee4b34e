// dev: what the DPP controls ctu_leaf4.h relies on do on gfx950 (lane i prints the lane whose value it received)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xf, 0xf, true); }
__global__ void k(int *out)
{
  const int l = threadIdx.x;
  out[0 * 64 + l] = dpp<0x101>(l);   // row_shl:1
  out[1 * 64 + l] = dpp<0x102>(l);
  out[2 * 64 + l] = dpp<0x104>(l);
  out[3 * 64 + l] = dpp<0x105>(l);
  out[4 * 64 + l] = dpp<0x108>(l);
  out[5 * 64 + l] = dpp<0x111>(l);   // row_shr:1
  out[6 * 64 + l] = dpp<0x140>(l);   // row_mirror
  out[7 * 64 + l] = dpp<0x141>(l);   // row_half_mirror
  out[8 * 64 + l] = __builtin_amdgcn_ds_bpermute(((l + 5) & 63) << 2, l);
}
int main()
{
  int *d; hipMalloc(&d, 9 * 64 * 4); k<<<1, 64>>>(d); int h[9 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char *nm[9] = {"row_shl:1", "row_shl:2", "row_shl:4", "row_shl:5", "row_shl:8", "row_shr:1", "row_mirror", "row_half_mirror", "bpermute(l+5)"};
  for (int r = 0; r < 9; ++r) { printf("%-16s", nm[r]); for (int l = 0; l < 20; ++l) printf(" %2d", h[r * 64 + l]); printf("\n"); }
  return 0;
}
