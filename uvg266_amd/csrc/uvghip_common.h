// Shared device/host helpers for libuvg266hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uvg266_hip.h"

#define UVGHIP_WAVE 64

// ---- host side -----------------------------------------------------------
int uvghip_set_error(hipError_t e, const char *where);
bool uvghip_ready();
// a small host table -> device memory in stream order (through kernel arguments; the host never waits for the stream): context.hip
int uvghip_upload_ordered(void *dst, const void *host, size_t bytes, hipStream_t st);

#define UVGHIP_REQUIRE_READY()                                                   \
  do {                                                                           \
    if (!uvghip_ready()) return uvghip_set_error(hipErrorNotInitialized, __func__); \
  } while (0)

// every entry point that selects the pixel type from `bitdepth` refuses anything but the two depths the library is built for
#define UVGHIP_REQUIRE_DEPTH(bd)                                                   \
  do {                                                                             \
    if ((bd) != 8 && (bd) != 10) return uvghip_set_error(hipErrorInvalidValue, __func__); \
  } while (0)

#define UVGHIP_CHECK_LAUNCH()                                         \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) return uvghip_set_error(e__, __func__);    \
    return 0;                                                         \
  } while (0)

#define UVGHIP_TRY(expr)                                              \
  do {                                                                \
    hipError_t e__ = (expr);                                          \
    if (e__ != hipSuccess) return uvghip_set_error(e__, #expr);       \
  } while (0)

static inline hipStream_t uvghip_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device side -----------------------------------------------------------
template <typename PX> struct px_traits;
template <> struct px_traits<uint8_t>  { static constexpr int depth = 8;  static constexpr int maxv = 255; };
template <> struct px_traits<uint16_t> { static constexpr int depth = 10; static constexpr int maxv = 1023; };

typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint16_t u16_unaligned __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) u32x2_unaligned { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) u32x4_unaligned { uint32_t a, b, c, d; };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Load 4 consecutive pixels starting at (x,y) of a plane (no clamping, any alignment).
__device__ __forceinline__ void load4(const uint8_t *p, int (&v)[4])
{
  const uint32_t w = *reinterpret_cast<const u32_unaligned *>(p);
  v[0] = w & 0xff; v[1] = (w >> 8) & 0xff; v[2] = (w >> 16) & 0xff; v[3] = w >> 24;
}
__device__ __forceinline__ void load4(const uint16_t *p, int (&v)[4])
{
  const u32x2_unaligned w = *reinterpret_cast<const u32x2_unaligned *>(p);
  v[0] = w.a & 0xffff; v[1] = w.a >> 16; v[2] = w.b & 0xffff; v[3] = w.b >> 16;
}
__device__ __forceinline__ void load8(const uint8_t *p, int (&v)[8])
{
  const u32x2_unaligned w = *reinterpret_cast<const u32x2_unaligned *>(p);
  v[0] = w.a & 0xff; v[1] = (w.a >> 8) & 0xff; v[2] = (w.a >> 16) & 0xff; v[3] = w.a >> 24;
  v[4] = w.b & 0xff; v[5] = (w.b >> 8) & 0xff; v[6] = (w.b >> 16) & 0xff; v[7] = w.b >> 24;
}
__device__ __forceinline__ void load8(const uint16_t *p, int (&v)[8])
{
  const u32x4_unaligned w = *reinterpret_cast<const u32x4_unaligned *>(p);
  v[0] = w.a & 0xffff; v[1] = w.a >> 16; v[2] = w.b & 0xffff; v[3] = w.b >> 16;
  v[4] = w.c & 0xffff; v[5] = w.c >> 16; v[6] = w.d & 0xffff; v[7] = w.d >> 16;
}

// N consecutive pixels of row y starting at column x, with edge replication
// outside [0,W)x[0,H) (what uvg_image_calc_sad's cor/ver/hor_sad amount to).
template <typename PX, int N>
__device__ __forceinline__ void load_row_clamped(const PX *plane, int stride, int W, int H, int x, int y,
                                                 int (&v)[N])
{
  const int yy = clampi(y, 0, H - 1);
  const PX *row = plane + (size_t)yy * stride;
  if (x >= 0 && x + N <= W) {
    if constexpr (N == 8) load8(row + x, v); else load4(row + x, v);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = row[clampi(x + i, 0, W - 1)];
  }
}

template <typename PX, int N>
__device__ __forceinline__ void load_row(const PX *plane, int stride, int x, int y, int (&v)[N])
{
  const PX *row = plane + (size_t)y * stride + x;
  if constexpr (N == 8) load8(row, v); else load4(row, v);
}

// Sum over the lowest `width` lanes of each aligned lane group (width: power of 2 <= 64).
__device__ __forceinline__ int group_sum(int v, int width)
{
  for (int off = width >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
