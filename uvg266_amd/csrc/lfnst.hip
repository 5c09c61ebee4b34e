// Low-frequency non-separable transform (LFNST) on gfx950, in place on batches of coefficient TUs.
// Bit-exact with (plain C in the reference, not a strategy pointer)
//   uvg_fwd_lfnst_NxN / uvg_fwd_lfnst   src/transform.c:880-917, 965-1077
//   uvg_inv_lfnst_NxN / uvg_inv_lfnst   src/transform.c:1079-1102, 1104-1225
//   get_lfnst_intra_mode, get_transpose_flag :919-943; uvg_wide_angle_correction src/intra.c:637-658
// One wave per TU: the 16 or 48 gathered inputs go through LDS, lane j owns output j (forward: 8 or 16
// outputs of 16/48 taps; inverse: 16/48 outputs of 8 or 16 taps).  The kernels are the normative tables of
// H.266 8.7.4.3 (vvc_lfnst_tables.h, generated -- see tools/gen_lfnst_tables.py); the mode -> set rule
// (8.7.4.1) and the diagonal scans are computed, not tabulated.  The work per TU is ~800 MACs: the kernel is
// bound by its 2 x 96-byte coefficient traffic per TU, i.e. by launch latency at any realistic TU count.
#include "uvghip_common.h"
#include "vvc_lfnst_tables.h"

namespace {

__device__ __forceinline__ int lfnst_set_of_mode(int m)   // lfnst_tables.h:51-54 as ranges
{
  return m <= 1 ? 0 : m <= 12 ? 1 : m <= 23 ? 2 : m <= 44 ? 3 : m <= 55 ? 2 : 1;
}
// position k of the up-right diagonal scan of a 4x4 group -> x | y << 2
__device__ __forceinline__ int diag4_xy(int k)
{
  // diagonals s = x + y hold 1,2,3,4,3,2,1 positions, walked from the bottom-left (largest y) upwards
  int s = 0, first = 0;
  bool open_ = true;                              // stop at the first diagonal that contains k
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int len = t < 4 ? t + 1 : 7 - t;
    if (open_ && k >= first + len) { first += len; s = t + 1; } else open_ = false;
  }
  const int j = k - first;                       // j-th position on diagonal s
  const int y = (s < 4 ? s : 3) - j, x = s - y;
  return x | (y << 2);
}
// raster offset of scan position j (0..47) of the LFNST region: groups (0,0), (0,1) [below], (1,0) [right]
__device__ __forceinline__ int lfnst_scan_pos(int j, int width)
{
  const int g = j >> 4, xy = diag4_xy(j & 15);
  const int cgx = g == 2 ? 1 : 0, cgy = g == 1 ? 1 : 0;
  return (cgy * 4 + (xy >> 2)) * width + cgx * 4 + (xy & 3);
}
// raster offset of gather index k (transform.c:1021-1061 forward, :1173-1207 inverse)
__device__ __forceinline__ int lfnst_gather_pos(int k, int width, bool big, bool transpose)
{
  int x, y;
  if (!big) { if (transpose) { x = k >> 2; y = k & 3; } else { y = k >> 2; x = k & 3; } }
  else if (transpose) { if (k < 32) { x = k >> 3; y = k & 7; } else { x = 4 + ((k - 32) >> 2); y = (k - 32) & 3; } }
  else { if (k < 32) { y = k >> 3; x = k & 7; } else { y = 4 + ((k - 32) >> 2); x = (k - 32) & 3; } }
  return y * width + x;
}

__global__ void __launch_bounds__(256)
lfnst_kernel(int inverse, int16_t *__restrict__ coeffs, int width, int height, const uvghip_lfnst_tu_t *__restrict__ tus, int n)
{
  __shared__ int sIn[4][48];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tu = blockIdx.x * 4 + wave;
  const bool on = tu < n;
  const uvghip_lfnst_tu_t d = tus[on ? tu : 0];
  const bool act = on && d.lfnst_idx >= 1 && d.lfnst_idx <= 2;
  // wide-angle correction with account_for_dc_planar (intra.c:637-658), then transform.c:919-943
  int pm = d.intra_mode;
  if (d.log2_cu_width != d.log2_cu_height && pm > 1 && pm <= 66) {
    const int shift_tab[6] = {0, 6, 10, 12, 14, 15};
    const int dl = abs(d.log2_cu_width - d.log2_cu_height);
    if (d.log2_cu_width > d.log2_cu_height && pm < 2 + shift_tab[dl]) pm += 65;
    else if (d.log2_cu_height > d.log2_cu_width && pm > 66 - shift_tab[dl]) pm -= 67;
  }
  const int m = pm < 0 ? pm + 14 + 67 : (pm >= 67 ? pm + 14 : pm);
  const bool transpose = (m >= 81) || (m < 67 && m > 34);
  const bool big = width >= 8 && height >= 8;
  const int tr_size = big ? 48 : 16;
  const int zero_out = ((width == 4 && height == 4) || (width == 8 && height == 8)) ? 8 : 16;
  const int8_t *M = (big ? VVC_LFNST8 : VVC_LFNST4) + (size_t)(lfnst_set_of_mode(m) * 2 + (act ? d.lfnst_idx - 1 : 0)) * 16 * tr_size;
  int16_t *c = coeffs + (size_t)(on ? tu : 0) * width * height;

  if (!inverse) {
    if (act && lane < tr_size) sIn[wave][lane] = c[lfnst_gather_pos(lane, width, big, transpose)];
    __syncthreads();
    if (act && lane < tr_size) {
      int v = 0;
      if (lane < zero_out) {
        int acc = 0;
        for (int i = 0; i < tr_size; ++i) acc += sIn[wave][i] * M[i * 16 + lane];    // input-major table: [input][output]
        v = (int)(int16_t)((acc + 64) >> 7);
      }
      c[lfnst_scan_pos(lane, width)] = (int16_t)v;
    }
  } else {
    if (act && lane < 16) sIn[wave][lane] = c[lfnst_scan_pos(lane, width)];
    __syncthreads();
    if (act && lane < tr_size) {
      int acc = 0;
      for (int i = 0; i < zero_out; ++i) acc += sIn[wave][i] * M[lane * 16 + i];
      // transform.c:1098: the cast to coeff_t precedes the clip, so the value wraps to 16 bits
      c[lfnst_gather_pos(lane, width, big, transpose)] = (int16_t)((acc + 64) >> 7);
    }
  }
}

}  // namespace

extern "C" int uvghip_lfnst_batch(int inverse, int16_t *coeffs, int width, int height, const uvghip_lfnst_tu_t *tus, int n,
                                  void *stream)
{
  UVGHIP_REQUIRE_READY();
  auto ok = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
  if (!ok(width) || !ok(height)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  lfnst_kernel<<<(n + 3) / 4, 256, 0, uvghip_stream(stream)>>>(inverse, coeffs, width, height, tus, n);
  UVGHIP_CHECK_LAUNCH();
}
