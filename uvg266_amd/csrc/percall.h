// Per-thread host<->device staging arena used by the drop-in strategy
// functions (one call = upload + launch + download + sync).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>

struct percall_ctx {
  hipStream_t stream = nullptr;
  char *h = nullptr;   // pinned host mirror
  char *d = nullptr;   // device arena
  size_t cap = 0, used = 0;
  ~percall_ctx();

  size_t take(size_t n) { size_t o = used; used += (n + 255) & ~size_t(255); return o; }
  template <typename T> T *hp(size_t off) { return reinterpret_cast<T *>(h + off); }
  template <typename T> T *dp(size_t off) { return reinterpret_cast<T *>(d + off); }
  void upload(size_t off, size_t n)
  {
    if (hipMemcpyAsync(d + off, h + off, n, hipMemcpyHostToDevice, stream) != hipSuccess) fail("H2D");
  }
  void download(size_t off, size_t n)
  {
    if (hipMemcpyAsync(h + off, d + off, n, hipMemcpyDeviceToHost, stream) != hipSuccess) fail("D2H");
  }
  void sync() { if (hipStreamSynchronize(stream) != hipSuccess) fail("sync"); }
  void must(int rc, const char *what) { if (rc != 0) fail(what); }
  [[noreturn]] void fail(const char *what);

  // copy a strided w x h block of elements of size es into the arena as a
  // tightly packed block; returns the arena offset
  size_t stage_block(const void *src, size_t stride_elems, int w, int h, size_t es)
  {
    const size_t off = take((size_t)w * h * es);
    for (int y = 0; y < h; ++y)
      memcpy(this->h + off + (size_t)y * w * es, (const char *)src + (size_t)y * stride_elems * es, (size_t)w * es);
    return off;
  }
};

// Returns the calling thread's arena, grown to at least `bytes`, reset to empty.
// Aborts (loudly) if the device is not initialised: there is no CPU fallback.
percall_ctx *percall_get(size_t bytes);
int uvghip_do_register(void *opaque, const char *type, void *fptr);
