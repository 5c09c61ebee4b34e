// uvghip_ctu_search_intra: the closed-loop intra search of whole pictures on the device (include/uvg266_hip.h, part 4).
// The per-CTU algorithm is csrc/ctu_core.h; this file is the launch: one workgroup per CTU, handed out in an order in which every
// CTU comes after its left and upper neighbour (the WPP dependencies, src/encoderstate.c:1160-1167), pictures interleaved so that
// the wavefronts of many pictures fill the device together.  A workgroup takes the next CTU of that order (a ticket from an atomic
// counter -- so a waiting workgroup only ever waits for CTUs that are already running or done), spins on its neighbours' "done"
// flags, runs the CTU, publishes its outputs (device-scope release) and raises its own flag.
#include "uvghip_common.h"
#include "ctu_core.h"
#include <vector>
#include <mutex>
#include <cstring>
#include <new>

namespace {

struct pic_dev {
  const void *src_y, *src_u, *src_v;
  void *rec_y, *rec_u, *rec_v;
  uvghip_scu_t *cu;
  int16_t *coeff;
  uint32_t *models;
  int src_stride, src_stride_c, rec_stride, rec_stride_c, cu_stride, pad;
};

struct launch_args {
  ctu::params P;
  const pic_dev *pics;
  const int32_t *order;       // [ticket] = pic << 16 | cy << 8 | cx
  int32_t *ticket;            // the counter
  int32_t *done;              // [pic * ctus + cy * wc + cx]
  ctu::scratch *scratch;      // per slot: a running workgroup owns one (far fewer slots than CTUs: the memory stays cache-resident)
  uint32_t *slots;            // bitmap of the slots in use
  int n_slots;
  int wc, hc, n_ctus;
  int32_t *simd_load;         // [XCD * 256 + (SE, SH, CU)][4]: walkers resident per SIMD of every CU
  int row0;                   // first CTU row of this launch (a band of a picture sharded by CTU rows): the row above it is complete in the buffers
};

// Four workgroups per CU at 8 bit is 40 960 B each: the dynamic image below + 4 304 B of function-scope tables (the build's .usage file
// shows "LDS Size [bytes/block]: 4304").  One more word and the device holds three workgroups per CU instead of four (-25 %).
static_assert(sizeof(ctu::lds<uint8_t>) + 4304 <= 40960, "the 8-bit LDS image of a CTU no longer fits four workgroups per CU");
static_assert(!ctu::lds_cfg<uint16_t>::slim || sizeof(ctu::lds<uint16_t>) + 4304 <= 40960, "the slim 10-bit LDS image of a CTU no longer fits four workgroups per CU");

// PERSIST: a small grid of workgroups that take CTU after CTU (uvghip_ctu_plan_set_grid) -- for a few pictures whose search runs BESIDE
// another kernel (the I pictures of a clip beside the in-flight P / B launch, which takes whole CUs): the launch then holds G workgroup
// slots instead of one per CTU, most of them waiting.  Same CTUs, same order, same results.
template <typename PX, bool PERSIST>
__global__ void __launch_bounds__(256, 4) ctu_search_kernel(launch_args A)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ctu::lds<PX> *S = reinterpret_cast<ctu::lds<PX> *>(smem);
  __shared__ int s_ticket;
  __shared__ int s_slot;
  __shared__ int s_load;               // index of this workgroup's walker's counter in A.simd_load
  __shared__ int s_simd[4];            // the SIMD every wave sits on as the workgroup starts
#if defined(CTU_POISON_LDS)          // debug builds (make EXTRA=-DCTU_POISON_LDS=0xA5): nothing may depend on what the LDS held before
  for (unsigned i = threadIdx.x; i < sizeof(ctu::lds<PX>); i += 256) smem[i] = (unsigned char)(CTU_POISON_LDS);
  __syncthreads();
#endif
  if ((threadIdx.x & 63) == 0) s_simd[threadIdx.x >> 6] = (int)((__builtin_amdgcn_s_getreg(63492) >> 4) & 3);      // HW_REG_HW_ID [5:4]
  __syncthreads();
  if (threadIdx.x == 0) {
    s_ticket = atomicAdd(A.ticket, 1);
    // claim a scratch slot: more slots than workgroups can ever be resident, so a free bit always exists
    const int words = A.n_slots >> 5;
    int got = -1;
    for (int i = s_ticket % words; got < 0; i = (i + 1 == words ? 0 : i + 1)) {
      const uint32_t cur = __hip_atomic_load(&A.slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0xffffffffu) continue;
      const int bit = __ffs((int)~cur) - 1;
      const uint32_t prev = atomicOr(&A.slots[i], 1u << bit);
      if (!(prev & (1u << bit))) got = i * 32 + bit;
    }
    s_slot = got;
    // Which wave walks the CTU: the one that sits on the SIMD carrying the fewest walkers of the workgroups resident on this CU (a
    // walker is the one wave that never idles; the four waves of a workgroup normally start on the four SIMDs).  Counters per CU in
    // the workspace.  A hint for speed only: nothing depends on where a wave really runs (ctu_core.h CTU_WAVE).
    {
      const uint32_t hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(63508) & 7u;      // HW_REG_HW_ID, HW_REG_XCC_ID
      int32_t *const c = A.simd_load + (size_t)((xcc << 8) | ((hw >> 8) & 0xffu)) * 4;
      int best = s_ticket & 3, lo = __hip_atomic_load(&c[s_simd[best]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int k = 1; k < 4; ++k) {
        const int w = (s_ticket + k) & 3, v = __hip_atomic_load(&c[s_simd[w]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < lo) { lo = v; best = w; }
      }
      atomicAdd(&c[s_simd[best]], 1);
      S->rot = best;
      s_load = (int)(&c[s_simd[best]] - A.simd_load);
    }
  }
  __syncthreads();
  for (;;) {
  const int ticket = s_ticket;
  if (PERSIST && ticket >= A.n_ctus) break;
  const int32_t o = A.order[ticket];
  const int pic = o >> 16, cy = (o >> 8) & 0xff, cx = o & 0xff;
  const int ctus = A.wc * A.hc, k = cy * A.wc + cx;
  int32_t *done = A.done + (size_t)pic * ctus;
  if (threadIdx.x == 0) {
    // Relaxed polls with a growing nap (an acquire per poll would invalidate this CU's L1 and the XCD's L2 under the other workgroups
    // every microsecond), then ONE agent-scope acquire: the L1 is the CU's, so the other waves' loads after the barrier are behind it.
    int naps = 1;
    const int32_t *deps[2] = {cx > 0 ? &done[k - 1] : nullptr, cy > A.row0 ? &done[k - A.wc] : nullptr};
    for (int d = 0; d < 2; ++d)
      while (deps[d] && __hip_atomic_load(deps[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
        if (naps < 8) naps <<= 1;
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  const pic_dev &D = A.pics[pic];
  ctu::job<PX> J;
  J.P = A.P;
  J.src_y = (const PX *)D.src_y; J.src_u = (const PX *)D.src_u; J.src_v = (const PX *)D.src_v;
  J.src_stride = D.src_stride; J.src_stride_c = D.src_stride_c;
  J.rec_y = (PX *)D.rec_y; J.rec_u = (PX *)D.rec_u; J.rec_v = (PX *)D.rec_v;
  J.rec_stride = D.rec_stride; J.rec_stride_c = D.rec_stride_c;
  J.cu_tab = D.cu; J.cu_stride = D.cu_stride;
  J.coeff = D.coeff + (size_t)k * 6144;
  J.models_out = D.models + (size_t)k * 3 * ctu::NMODELS;
  J.models_in = cx > 0 ? D.models + ((size_t)(k - 1) * 3 + 2) * ctu::NMODELS
                       : (cy > 0 ? D.models + ((size_t)((cy - 1) * A.wc) * 3 + 2) * ctu::NMODELS : nullptr);
  J.W = A.scratch + s_slot;
  J.x = cx * 64; J.y = cy * 64;
  ctu::run_ctu(S, J);
  __syncthreads();          // every wave's stores are complete (workgroup-scope release) ...
  if (threadIdx.x == 0) {
    // ... and ONE agent-scope release writes the XCD's L2 back (it is shared by the waves): a fence per wave did that four times over
    __hip_atomic_store(&done[k], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (PERSIST) s_ticket = atomicAdd(A.ticket, 1);
  }
  if (!PERSIST) break;
  __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAnd(&A.slots[s_slot >> 5], ~(1u << (s_slot & 31)));          // the scratch is free again
    atomicSub(&A.simd_load[s_load], 1);
  }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Profiling knob (tools/dev/pmc_icache.sh): UVGHIP_CTU_LDS_PAD=<bytes> pads the dynamic LDS of the search kernel so that fewer
// workgroups fit a CU (occupancy experiments: 1 / 2 / 4 workgroups per CU).  Unset = 0; results never depend on it.
size_t lds_pad()
{
  static const size_t pad = [] { const char *e = getenv("UVGHIP_CTU_LDS_PAD"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 && v < 120 * 1024 ? v : 0); }();
  return pad;
}

// Scratch slots: a workgroup claims one while it runs.  2048 = 256 CUs x 8 is more than the device can hold of this kernel
// (4 per CU); small jobs take one slot per CTU, rounded up to whole bitmap words.
enum { MAX_SLOTS = 2048 };
struct ws_layout { size_t ticket, slots, simd_load, done, order, pics, scratch, total; int n_slots; };
ws_layout layout(int n_pictures, int pic_w, int pic_h)
{
  const size_t ctus = (size_t)((pic_w + 63) / 64) * ((pic_h + 63) / 64), total = ctus * n_pictures;
  ws_layout L;
  L.n_slots = (int)(total < MAX_SLOTS ? align_up(total, 32) : MAX_SLOTS);
  L.ticket = 0;
  L.slots = 256;
  L.simd_load = L.slots + MAX_SLOTS / 8;
  L.done = L.simd_load + 8 * 256 * 4 * sizeof(int32_t);
  L.order = align_up(L.done + total * 4, 256);          // [0, order): zeroed before every run
  L.pics = align_up(L.order + total * 4, 256);
  L.scratch = align_up(L.pics + (size_t)n_pictures * sizeof(pic_dev), 256);
  L.total = L.scratch + (size_t)L.n_slots * sizeof(ctu::scratch);
  return L;
}

}  // namespace

extern "C" size_t uvghip_ctu_search_workspace_bytes(int n_pictures, int pic_w, int pic_h)
{
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0) return 0;
  return layout(n_pictures, pic_w, pic_h).total;
}

// A plan: the validated configuration, the hand-out order and the picture table uploaded once; a run is a memset of the
// counters and one launch -- nothing on the host waits, so launches of several plans on several streams overlap (the thin
// start of one plan's wavefronts fills the device while another's drain).
struct uvghip_ctu_plan {
  launch_args A;
  int bitdepth, total;
  int grid;                   // 0: a workgroup per CTU; else that many persistent workgroups (uvghip_ctu_plan_set_grid)
  size_t counters;            // bytes of (ticket, done flags) at the head of the workspace
  unsigned char *ws;
};

extern "C" int uvghip_ctu_plan_create(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, int n_pictures,
                                      void *workspace, uvghip_ctu_plan_t **plan_out)
{
  return uvghip_ctu_plan_create_rows(bitdepth, params, pictures, n_pictures, 0, params ? (params->pic_h + 63) / 64 : 0, workspace, plan_out);
}

// A band of CTU rows [ctu_row0, ctu_row1) of every picture (one picture sharded over GPUs by CTU rows): the same search,
// released in the same order, with the row above the band taken as complete -- the caller has put its last sample line, its last row of
// side information and the models of its first CTU into the pictures' buffers (the halo a band receives from the band above,
// encoderstate.c:196-323, 966-975: hor_buf_search, the cu_array row, the WPP context hand-over).
extern "C" int uvghip_ctu_plan_create_rows(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, int n_pictures,
                                           int ctu_row0, int ctu_row1, void *workspace, uvghip_ctu_plan_t **plan_out)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  static_assert(sizeof(uvghip_ctu_params_t) == sizeof(ctu::params), "uvghip_ctu_params_t mirrors ctu::params");
  if (!params || !pictures || n_pictures <= 0 || !workspace || !plan_out) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const uvghip_ctu_params_t &p = *params;
  if (p.pic_w <= 0 || p.pic_h <= 0 || (p.pic_w & 7) || (p.pic_h & 7) || p.pic_w > 64 * 255 || p.pic_h > 64 * 255 || n_pictures > 32767)
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_plan_create: picture size");
  if (p.wpp != 1 || p.depth_min < 1 || p.depth_max > 4 || p.depth_min > p.depth_max || p.rough_levels < 2 || p.rough_levels > 3 || p.qp < 0 || p.qp > 63 ||
      p.qp_c < 0 || p.qp_c > 63 || !(p.lambda > 0) || p.rd < 0 || p.rd > 1)
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_plan_create: configuration outside the supported subset");
  // combine_intra_cus tries the first child's mode at EVERY depth without a search of its own (search.c:2082-2143); the kernel builds the
  // candidate at depth 0 only, which is all there is with pu-depth-intra starting at 1 (tests/test_ctu_emulation.py, other ranges)
  if (p.combine_intra_cus && p.depth_min > 1)
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_plan_create: combine_intra_cus with pu-depth-intra starting below depth 1 is not supported");
  const int wc = (p.pic_w + 63) / 64, hc = (p.pic_h + 63) / 64, ctus = wc * hc;
  if (ctu_row0 < 0 || ctu_row1 > hc || ctu_row0 >= ctu_row1) return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_plan_create_rows: CTU row range");
  const int total = wc * (ctu_row1 - ctu_row0) * n_pictures;
  const ws_layout L = layout(n_pictures, p.pic_w, p.pic_h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  // hand-out order: wavefront index first, pictures interleaved inside a wavefront
  std::vector<int32_t> order;
  order.reserve(total);
  for (int d = 0; d < wc + hc - 1; ++d)
    for (int pic = 0; pic < n_pictures; ++pic)
      for (int cy = ctu_row0; cy < ctu_row1; ++cy) {
        const int cx = d - cy;
        if (cx >= 0 && cx < wc) order.push_back(pic << 16 | cy << 8 | cx);
      }
  std::vector<pic_dev> pics(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_ctu_picture_t &q = pictures[i];
    if (!q.src_y || !q.src_u || !q.src_v || !q.rec_y || !q.rec_u || !q.rec_v || !q.cu || !q.coeff || !q.models || q.cu_stride < wc * 16)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_plan_create: picture descriptor");
    if (q.src_stride < p.pic_w || q.rec_stride < p.pic_w || q.src_stride_c < p.pic_w / 2 || q.rec_stride_c < p.pic_w / 2)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_plan_create: a sample stride is smaller than the picture");
    pics[i] = pic_dev{q.src_y, q.src_u, q.src_v, q.rec_y, q.rec_u, q.rec_v, q.cu, q.coeff, q.models,
                      q.src_stride, q.src_stride_c, q.rec_stride, q.rec_stride_c, q.cu_stride, 0};
  }
  UVGHIP_TRY(hipMemcpy(ws + L.order, order.data(), (size_t)total * 4, hipMemcpyHostToDevice));
  UVGHIP_TRY(hipMemcpy(ws + L.pics, pics.data(), (size_t)n_pictures * sizeof(pic_dev), hipMemcpyHostToDevice));
  uvghip_ctu_plan *pl = new (std::nothrow) uvghip_ctu_plan;
  if (!pl) return uvghip_set_error(hipErrorOutOfMemory, __func__);
  memcpy(&pl->A.P, params, sizeof pl->A.P);
  pl->A.pics = reinterpret_cast<const pic_dev *>(ws + L.pics);
  pl->A.order = reinterpret_cast<const int32_t *>(ws + L.order);
  pl->A.ticket = reinterpret_cast<int32_t *>(ws + L.ticket);
  pl->A.done = reinterpret_cast<int32_t *>(ws + L.done);
  pl->A.scratch = reinterpret_cast<ctu::scratch *>(ws + L.scratch);
  pl->A.slots = reinterpret_cast<uint32_t *>(ws + L.slots);
  pl->A.simd_load = reinterpret_cast<int32_t *>(ws + L.simd_load);
  pl->A.n_slots = L.n_slots;
  pl->A.wc = wc; pl->A.hc = hc; pl->A.n_ctus = total; pl->A.row0 = ctu_row0;
  pl->bitdepth = bitdepth; pl->total = total; pl->counters = L.order; pl->ws = ws; pl->grid = 0;
  const size_t lds = (bitdepth == 8 ? sizeof(ctu::lds<uint8_t>) : sizeof(ctu::lds<uint16_t>)) + lds_pad();
  const hipError_t e = bitdepth == 8
      ? hipFuncSetAttribute(reinterpret_cast<const void *>(&ctu_search_kernel<uint8_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
      : hipFuncSetAttribute(reinterpret_cast<const void *>(&ctu_search_kernel<uint16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { delete pl; return uvghip_set_error(e, "uvghip_ctu_plan_create: dynamic LDS size"); }
  const hipError_t e2 = bitdepth == 8
      ? hipFuncSetAttribute(reinterpret_cast<const void *>(&ctu_search_kernel<uint8_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
      : hipFuncSetAttribute(reinterpret_cast<const void *>(&ctu_search_kernel<uint16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e2 != hipSuccess) { delete pl; return uvghip_set_error(e2, "uvghip_ctu_plan_create: dynamic LDS size"); }
  *plan_out = pl;
  return 0;
}

// A run in two halves, for a caller that lets ANOTHER stream's kernel wait for this plan's per-CTU flags (pictures in flight behind an I
// picture, uvghip_loop_pb_run_inflight_ext): reset -- the counters and flags back to zero, in stream order; the other stream waits for
// an event recorded behind it -- then launch.  uvghip_ctu_plan_run is the two in a row.
extern "C" int uvghip_ctu_plan_reset(uvghip_ctu_plan_t *pl, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  UVGHIP_TRY(hipMemsetAsync(pl->ws, 0, pl->counters, uvghip_stream(stream)));
  return 0;
}
extern "C" int uvghip_ctu_plan_launch(uvghip_ctu_plan_t *pl, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  const size_t lds = (pl->bitdepth == 8 ? sizeof(ctu::lds<uint8_t>) : sizeof(ctu::lds<uint16_t>)) + lds_pad();
  if (pl->grid > 0 && pl->grid < pl->total) {
    if (pl->bitdepth == 8) hipLaunchKernelGGL((ctu_search_kernel<uint8_t, true>), dim3(pl->grid), dim3(256), lds, st, pl->A);
    else hipLaunchKernelGGL((ctu_search_kernel<uint16_t, true>), dim3(pl->grid), dim3(256), lds, st, pl->A);
  } else {
    if (pl->bitdepth == 8) hipLaunchKernelGGL((ctu_search_kernel<uint8_t, false>), dim3(pl->total), dim3(256), lds, st, pl->A);
    else hipLaunchKernelGGL((ctu_search_kernel<uint16_t, false>), dim3(pl->total), dim3(256), lds, st, pl->A);
  }
  UVGHIP_CHECK_LAUNCH();
}
extern "C" int uvghip_ctu_plan_run(uvghip_ctu_plan_t *pl, void *stream)
{
  if (int rc = uvghip_ctu_plan_reset(pl, stream)) return rc;
  return uvghip_ctu_plan_launch(pl, stream);
}
// ... max_workgroups > 0: the launch is that many persistent workgroups (0: one per CTU, the default)
extern "C" int uvghip_ctu_plan_set_grid(uvghip_ctu_plan_t *pl, int max_workgroups)
{
  if (!pl || max_workgroups < 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  pl->grid = max_workgroups;
  return 0;
}
// ... the per-CTU "searched" flags [picture][ctu] (DEVICE memory; zero after the reset, 1 when the CTU's outputs are published)
extern "C" const int32_t *uvghip_ctu_plan_done_flags(const uvghip_ctu_plan_t *pl) { return pl ? pl->A.done : nullptr; }

extern "C" void uvghip_ctu_plan_destroy(uvghip_ctu_plan_t *pl) { delete pl; }

// one-shot form: plan, run, wait, drop the plan
extern "C" int uvghip_ctu_search_intra(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, int n_pictures,
                                       void *workspace, void *stream)
{
  uvghip_ctu_plan_t *pl = nullptr;
  int rc = uvghip_ctu_plan_create(bitdepth, params, pictures, n_pictures, workspace, &pl);
  if (rc) return rc;
  rc = uvghip_ctu_plan_run(pl, stream);
  if (!rc) {
    const hipError_t e = hipStreamSynchronize(uvghip_stream(stream));
    if (e != hipSuccess) rc = uvghip_set_error(e, "uvghip_ctu_search_intra");
  }
  uvghip_ctu_plan_destroy(pl);
  return rc;
}
