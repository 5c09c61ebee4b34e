// Process context, error reporting and per-thread staging for the per-call
// ("drop-in") strategy path of libuvg266hip.so.
#include "uvghip_common.h"
#include "percall.h"
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>

static std::atomic<int> g_ready{0};
static int g_device = -1;
static thread_local char t_err[256] = "";
static uvghip_register_fn g_register_fn = nullptr;

// The host encoder's registration routine (src/strategyselector.c:242).  Weak:
// resolved when the library is loaded into the encoder, absent in stand-alone use.
extern "C" int uvg_strategyselector_register(void *opaque, const char *type, const char *strategy_name,
                                              int priority, void *fptr) __attribute__((weak));

int uvghip_set_error(hipError_t e, const char *where)
{
  snprintf(t_err, sizeof t_err, "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
  return (int)e ? (int)e : -1;
}

bool uvghip_ready() { return g_ready.load(std::memory_order_acquire) != 0; }

extern "C" const char *uvghip_last_error(void) { return t_err; }
extern "C" int uvghip_abi_version(void) { return 1; }

extern "C" int uvghip_init(int device)
{
  static std::mutex m;
  std::lock_guard<std::mutex> lk(m);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_ready.store(0);
    return uvghip_set_error(e != hipSuccess ? e : hipErrorNoDevice, "uvghip_init: no HIP device");
  }
  if (device < 0 || device >= count) return uvghip_set_error(hipErrorInvalidDevice, "uvghip_init");
  UVGHIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  UVGHIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(t_err, sizeof t_err, "uvghip_init: device %d is %s, this library is built for gfx950 only",
             device, prop.gcnArchName);
    return (int)hipErrorInvalidDevice;
  }
  g_device = device;
  g_ready.store(1, std::memory_order_release);
  return 0;
}

// ---- a small host table into device memory in STREAM ORDER, through kernel arguments: hipMemcpy would have to drain the stream first (an
// earlier call on the same workspace may still read the table), and a host with independent pictures to issue must not be held up ----
namespace {
struct up_chunk { unsigned char b[3072]; };
__global__ void __launch_bounds__(256) upload_kernel(unsigned char *dst, up_chunk c, int n)
{
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = c.b[i];
}
}  // namespace
int uvghip_upload_ordered(void *dst, const void *host, size_t bytes, hipStream_t st)
{
  const unsigned char *src = static_cast<const unsigned char *>(host);
  for (size_t at = 0; at < bytes; at += sizeof(up_chunk)) {
    up_chunk c;
    const int n = (int)(bytes - at < sizeof(up_chunk) ? bytes - at : sizeof(up_chunk));
    memcpy(c.b, src + at, (size_t)n);
    hipLaunchKernelGGL(upload_kernel, dim3(1), dim3(256), 0, st, static_cast<unsigned char *>(dst) + at, c, n);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : uvghip_set_error(e, "uvghip_upload_ordered");
}

// ---- launch plans as hipGraphs: a picture's fixed kernel sequence is captured once and replayed with one call ----
extern "C" int uvghip_graph_begin(void *stream)
{
  if (!uvghip_ready()) return uvghip_set_error(hipErrorNotInitialized, __func__);
  UVGHIP_TRY(hipStreamBeginCapture(uvghip_stream(stream), hipStreamCaptureModeThreadLocal));
  return 0;
}

extern "C" int uvghip_graph_end(void *stream, void **graph_exec_out)
{
  if (!graph_exec_out) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipGraph_t g = nullptr;
  UVGHIP_TRY(hipStreamEndCapture(uvghip_stream(stream), &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return uvghip_set_error(e, "hipGraphInstantiate");
  *graph_exec_out = ge;
  return 0;
}

extern "C" int uvghip_graph_launch(void *graph_exec, void *stream)
{
  if (!graph_exec) return uvghip_set_error(hipErrorInvalidValue, __func__);
  UVGHIP_TRY(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), uvghip_stream(stream)));
  return 0;
}

extern "C" int uvghip_graph_destroy(void *graph_exec)
{
  if (graph_exec) UVGHIP_TRY(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
  return 0;
}

extern "C" void uvghip_set_register_fn(uvghip_register_fn fn) { g_register_fn = fn; }

int uvghip_do_register(void *opaque, const char *type, void *fptr)
{
  if (g_register_fn) return g_register_fn(opaque, type, UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, fptr);
  if (uvg_strategyselector_register)
    return uvg_strategyselector_register(opaque, type, UVGHIP_STRATEGY_NAME, UVGHIP_STRATEGY_PRIORITY, fptr);
  snprintf(t_err, sizeof t_err, "no uvg_strategyselector_register symbol and no uvghip_set_register_fn()");
  return 0;
}

// ---- per-thread staging (the strategy pointers are called concurrently from
// all threadqueue workers, src/threadqueue.c:275, so nothing here is shared) --
percall_ctx::~percall_ctx()
{
  if (h) (void)hipHostFree(h);
  if (d) (void)hipFree(d);
  if (stream) (void)hipStreamDestroy(stream);
}

percall_ctx *percall_get(size_t bytes)
{
  static thread_local percall_ctx ctx;
  if (!uvghip_ready()) {
    // The strategy typedefs have no error channel; a backend that
    // registered itself but lost its device must not silently return garbage.
    fprintf(stderr, "uvg266hip: strategy called without an initialised gfx950 device\n");
    abort();
  }
  if (!ctx.stream) {
    if (hipSetDevice(g_device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking) != hipSuccess) {
      fprintf(stderr, "uvg266hip: cannot create per-thread stream\n");
      abort();
    }
  }
  if (bytes > ctx.cap) {
    if (ctx.h) (void)hipHostFree(ctx.h);
    if (ctx.d) (void)hipFree(ctx.d);
    size_t cap = 1 << 20;
    while (cap < bytes) cap <<= 1;
    if (hipHostMalloc(&ctx.h, cap, hipHostMallocDefault) != hipSuccess || hipMalloc(&ctx.d, cap) != hipSuccess) {
      fprintf(stderr, "uvg266hip: cannot allocate %zu bytes of staging\n", cap);
      abort();
    }
    ctx.cap = cap;
  }
  ctx.used = 0;
  return &ctx;
}

void percall_ctx::fail(const char *what)
{
  fprintf(stderr, "uvg266hip: %s failed: %s\n", what, uvghip_last_error());
  abort();
}
