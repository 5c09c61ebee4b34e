// CTU-row bands: one picture sharded over the GPUs of a node (the reference's own row parallelism is
// encoderstate.c:1085-1189 -- one job per CTU row with a dependency on the row above).
//
//   * uvghip_band_plan: which CTU rows a rank owns and who its neighbours are (host only, no device).
//   * uvghip_comm_*: the data-path exchanges between ranks, issued with RCCL directly on the caller's HIP stream:
//       halo rows            neighbour point-to-point, ncclSend/ncclRecv inside one ncclGroupStart/End
//       recon "all-gather-v" every rank's band to every other rank (bands are uneven: 34 CTU rows over 8 ranks = 5,5,4,...),
//                            as grouped ncclSend/ncclRecv pairs -- on the xGMI full mesh each pair has its own link, so
//                            the pairwise form uses all seven links at once where a ring all-gather is per-link bound
//       ALF covariances      ncclAllReduce (int64 sum)
//     librccl is opened lazily with dlopen: a single-GPU user of the library (the per-call strategy path, the
//     batched kernels) never loads it.  In a process that already holds an RCCL (PyTorch-ROCm bundles one under the
//     same soname) that copy is the one found.
#include "uvghip_common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>

extern "C" int uvghip_band_plan(int pic_h, int nranks, int rank, uvghip_band_plan_t *out)
{
  if (!out || pic_h <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int rows = (pic_h + 63) / 64;
  if (nranks > rows) return uvghip_set_error(hipErrorInvalidValue, "uvghip_band_plan: more ranks than CTU rows");
  const int base = rows / nranks, rem = rows % nranks;
  const int r0 = rank * base + (rank < rem ? rank : rem);
  const int r1 = r0 + base + (rank < rem ? 1 : 0);
  out->rank = rank; out->nranks = nranks; out->ctu_rows = rows;
  out->ctu_row0 = r0; out->ctu_row1 = r1;
  out->y0 = r0 * 64; out->y1 = r1 * 64 < pic_h ? r1 * 64 : pic_h;
  out->up = rank > 0 ? rank - 1 : -1;
  out->down = rank + 1 < nranks ? rank + 1 : -1;
  return 0;
}

// ------------------------------------------------------------------------------------------------ RCCL ----
namespace {

struct rccl_api {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

rccl_api g_rccl;
std::mutex g_rccl_mutex;
thread_local char t_rccl_err[256];

int rccl_fail(const char *where, const char *what)
{
  snprintf(t_rccl_err, sizeof t_rccl_err, "%s: %s", where, what);
  // routed through the library's error text so uvghip_last_error() reports it
  uvghip_set_error(hipErrorUnknown, t_rccl_err);
  return (int)hipErrorUnknown;
}

int rccl_load()
{
  std::lock_guard<std::mutex> lk(g_rccl_mutex);
  if (g_rccl.handle) return 0;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return rccl_fail("uvghip_comm: dlopen(librccl)", dlerror());
#define SYM(name)                                                                  \
  g_rccl.name = reinterpret_cast<decltype(g_rccl.name)>(dlsym(h, "nccl" #name));   \
  if (!g_rccl.name) return rccl_fail("uvghip_comm: dlsym", "nccl" #name)
  SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(GroupStart); SYM(GroupEnd); SYM(Send); SYM(Recv);
  SYM(AllReduce); SYM(AllGather); SYM(GetErrorString);
#undef SYM
  g_rccl.handle = h;
  return 0;
}

#define RCCL_TRY(expr)                                                                   \
  do {                                                                                   \
    ncclResult_t r__ = (expr);                                                           \
    if (r__ != ncclSuccess) return rccl_fail(#expr, g_rccl.GetErrorString(r__));         \
  } while (0)

struct comm_ctx {
  ncclComm_t comm;
  int rank, nranks;
};

}  // namespace

static_assert(UVGHIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int uvghip_comm_unique_id(void *id_host)
{
  if (!id_host) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (int rc = rccl_load()) return rc;
  ncclUniqueId id;
  RCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id_host, &id, sizeof id);
  return 0;
}

extern "C" int uvghip_comm_create(const void *id_host, int nranks, int rank, void **comm_out)
{
  UVGHIP_REQUIRE_READY();
  if (!id_host || !comm_out || nranks <= 0 || rank < 0 || rank >= nranks) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (int rc = rccl_load()) return rc;
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof id);
  ncclComm_t c;
  RCCL_TRY(g_rccl.CommInitRank(&c, nranks, id, rank));
  *comm_out = new comm_ctx{c, rank, nranks};
  return 0;
}

extern "C" int uvghip_comm_destroy(void *comm)
{
  if (!comm) return 0;
  comm_ctx *c = static_cast<comm_ctx *>(comm);
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  delete c;
  return r == ncclSuccess ? 0 : rccl_fail("ncclCommDestroy", g_rccl.GetErrorString(r));
}

extern "C" int uvghip_comm_exchange(void *comm, const uvghip_xfer_t *xfers_host, int n, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!comm || (n > 0 && !xfers_host) || n < 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  comm_ctx *c = static_cast<comm_ctx *>(comm);
  if (n == 0) return 0;
  for (int i = 0; i < n; ++i)
    if (xfers_host[i].peer < 0 || xfers_host[i].peer >= c->nranks) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  RCCL_TRY(g_rccl.GroupStart());
  for (int i = 0; i < n; ++i) {
    const uvghip_xfer_t &x = xfers_host[i];
    ncclResult_t r = ncclSuccess;
    if (x.send_bytes) r = g_rccl.Send(x.send, x.send_bytes, ncclUint8, x.peer, c->comm, st);
    if (r == ncclSuccess && x.recv_bytes) r = g_rccl.Recv(x.recv, x.recv_bytes, ncclUint8, x.peer, c->comm, st);
    if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return rccl_fail("ncclSend/ncclRecv", g_rccl.GetErrorString(r)); }
  }
  RCCL_TRY(g_rccl.GroupEnd());
  return 0;
}

extern "C" int uvghip_comm_allreduce_i64(void *comm, int64_t *buf, size_t count, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!comm || (!buf && count)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (!count) return 0;
  comm_ctx *c = static_cast<comm_ctx *>(comm);
  RCCL_TRY(g_rccl.AllReduce(buf, buf, count, ncclInt64, ncclSum, c->comm, uvghip_stream(stream)));
  return 0;
}

extern "C" int uvghip_comm_allgather(void *comm, const void *send, void *recv, size_t bytes_per_rank, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!comm || !send || !recv) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (!bytes_per_rank) return 0;
  comm_ctx *c = static_cast<comm_ctx *>(comm);
  RCCL_TRY(g_rccl.AllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, uvghip_stream(stream)));
  return 0;
}
