// uvghip_ctu_search_pb: the closed-loop CTU search of P / B pictures on the device (include/uvg266_hip.h, part 8).  The per-CTU
// algorithm is csrc/ctu_pb.h (on csrc/ctu_core.h built with CTU_PB); this file is the launch: one wave per CTU, handed out in an order
// in which every CTU comes after its left, upper and upper-RIGHT neighbour -- the upper-right one because the deblocking side effect
// of a CTU (ctu_pb.h deblock_zeroes_unused_vectors) rewrites stored vectors in the two 4-columns left of it, which the CTU below
// those columns reads as its neighbour row (the reference, single-threaded in the configuration the goldens pin, runs CTUs in raster
// order).  Pictures of one call are independent of each other (pictures of different sequences, or of one sequence's different
// temporal positions whose references are complete); their wavefronts interleave.
#define CTU_PB 1
#include "uvghip_common.h"
#include "ctu_pb.h"
#include <vector>
#include <cstring>
#include <new>

namespace {

struct pb_pic_dev {
  ctu::params P;
  ctu::pb_job B;
  const void *src_y, *src_u, *src_v;
  void *rec_y, *rec_u, *rec_v;
  uvghip_scu_t *cu;
  int16_t *coeff;
  uint32_t *models, *models_inter;
  int src_stride, src_stride_c, rec_stride, rec_stride_c, cu_stride, pad;
};

struct pb_launch_args {
  const pb_pic_dev *pics;
  const int32_t *order;       // [ticket] = pic << 16 | cy << 8 | cx
  int32_t *ticket;
  int32_t *done;              // [pic * ctus + cy * wc + cx]
  ctu::scratch *scratch;
  uint32_t *slots;
  int n_slots;
  int wc, hc, n_ctus;
};

// four workgroups (= four waves: the kernel's registers allow one per SIMD) per CU at 8 bit: 160 KB / 4 incl. the 4288 bytes of static tables
static_assert(sizeof(ctu::lds<uint8_t>) + 4288 <= 40960, "the 8-bit LDS image of a P / B CTU no longer fits four workgroups per CU");
template <typename PX>
__global__ void __launch_bounds__(64) ctu_search_pb_kernel(pb_launch_args A)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ctu::lds<PX> *S = reinterpret_cast<ctu::lds<PX> *>(smem);
  __shared__ int s_ticket;
  __shared__ int s_slot;
#if defined(CTU_POISON_LDS)
  for (unsigned i = threadIdx.x; i < sizeof(ctu::lds<PX>); i += 64) smem[i] = (unsigned char)(CTU_POISON_LDS);
  __syncthreads();
#endif
  // a scratch slot for the workgroup's lifetime
  if (threadIdx.x == 0) {
    const int words = A.n_slots >> 5;
    int got = -1;
    for (int i = (int)(blockIdx.x % (unsigned)words); got < 0; i = (i + 1 == words ? 0 : i + 1)) {
      const uint32_t cur = __hip_atomic_load(&A.slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0xffffffffu) continue;
      const int bit = __ffs((int)~cur) - 1;
      const uint32_t prev = atomicOr(&A.slots[i], 1u << bit);
      if (!(prev & (1u << bit))) got = i * 32 + bit;
    }
    s_slot = got;
    S->rot = 0;       // one wave: role 0 (ctu_core.h CTU_WAVE)
  }
  // The workgroup takes CTU after CTU in the hand-out order until none is left.  The grid is a few workgroups per CTU of a picture's
  // widest diagonal (uvghip_ctu_search_pb), not one per CTU: a workgroup waiting for its neighbours holds a quarter of a CU, and with
  // several calls in flight (pictures of a random-access GOP on their own streams) the waiting ones of one picture would keep the
  // runnable ones of another off the device.  No deadlock at any grid size: a ticket's neighbours hold smaller tickets, which are
  // done or in the hands of running workgroups.
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(A.ticket, 1);
    __syncthreads();
    const int ticket = s_ticket;
    if (ticket >= A.n_ctus) break;
    const int32_t o = A.order[ticket];
    const int pic = o >> 16, cy = (o >> 8) & 0xff, cx = o & 0xff;
    const int ctus = A.wc * A.hc, k = cy * A.wc + cx;
    int32_t *done = A.done + (size_t)pic * ctus;
    if (threadIdx.x == 0) {
      int naps = 1;
      const int32_t *deps[3] = {cx > 0 ? &done[k - 1] : nullptr, cy > 0 ? &done[k - A.wc] : nullptr, cy > 0 && cx + 1 < A.wc ? &done[k - A.wc + 1] : nullptr};
      for (int d = 0; d < 3; ++d)
        while (deps[d] && __hip_atomic_load(deps[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
          if (naps < 8) naps <<= 1;
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const pb_pic_dev &D = A.pics[pic];
    ctu::job<PX> J;
    J.P = D.P;
    J.src_y = (const PX *)D.src_y; J.src_u = (const PX *)D.src_u; J.src_v = (const PX *)D.src_v;
    J.src_stride = D.src_stride; J.src_stride_c = D.src_stride_c;
    J.rec_y = (PX *)D.rec_y; J.rec_u = (PX *)D.rec_u; J.rec_v = (PX *)D.rec_v;
    J.rec_stride = D.rec_stride; J.rec_stride_c = D.rec_stride_c;
    J.cu_tab = D.cu; J.cu_stride = D.cu_stride;
    J.coeff = D.coeff + (size_t)k * 6144;
    J.models_out = D.models + (size_t)k * 3 * ctu::NMODELS;
    J.pbm_out = D.models_inter + (size_t)k * 3 * 18;
    const int from = cx > 0 ? k - 1 : (cy > 0 ? (cy - 1) * A.wc : -1);
    J.models_in = from >= 0 ? D.models + ((size_t)from * 3 + 2) * ctu::NMODELS : nullptr;
    J.pbm_in = from >= 0 ? D.models_inter + ((size_t)from * 3 + 2) * 18 : nullptr;
    J.slice_type = D.B.slice_type; J.init_qp = D.B.frame_qp;
    J.pb = &D.B;
    J.W = A.scratch + s_slot;
    J.x = cx * 64; J.y = cy * 64;
    ctu::run_ctu_pb(S, J);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&done[k], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) atomicAnd(&A.slots[s_slot >> 5], ~(1u << (s_slot & 31)));
}

// The hand-out order, written on the device (stream-ordered: no host copy to wait for): index cx + 2 * cy first -- left, upper and
// upper-right neighbours all have a smaller one --, pictures interleaved.  One block.
__global__ void __launch_bounds__(256) pb_order_kernel(int32_t *order, int wc, int hc, int n_pictures)
{
  __shared__ int base[768];
  const int nd = wc + 2 * (hc - 1);
  if (threadIdx.x == 0) {
    int at = 0;
    for (int d = 0; d < nd; ++d) {
      const int lo = d - (wc - 1) > 0 ? (d - (wc - 1) + 1) / 2 : 0, hi = d / 2 < hc - 1 ? d / 2 : hc - 1;
      base[d] = at;
      at += (hi >= lo ? hi - lo + 1 : 0) * n_pictures;
    }
  }
  __syncthreads();
  for (int d = 0; d < nd; ++d) {
    const int lo = d - (wc - 1) > 0 ? (d - (wc - 1) + 1) / 2 : 0, hi = d / 2 < hc - 1 ? d / 2 : hc - 1, cnt = hi >= lo ? hi - lo + 1 : 0;
    for (int i = threadIdx.x; i < cnt * n_pictures; i += blockDim.x) {
      const int pic = i / cnt, cy = lo + (i - pic * cnt);
      order[base[d] + i] = pic << 16 | cy << 8 | (d - 2 * cy);
    }
  }
}
// A small host table into device memory in stream order, through kernel arguments: hipMemcpy would have to drain the stream first (an
// earlier call on this workspace may still read the table), which stalls a host that has independent pictures to issue.
struct pb_chunk { unsigned char b[3072]; };
__global__ void __launch_bounds__(256) pb_upload_kernel(unsigned char *dst, pb_chunk c, int n)
{
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = c.b[i];
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
enum { MAX_SLOTS = 2048 };
struct ws_layout { size_t ticket, slots, done, hmvp, order, pics, scratch, total; int n_slots; };
ws_layout layout(int n_pictures, int pic_w, int pic_h)
{
  const size_t hc = (size_t)((pic_h + 63) / 64), ctus = (size_t)((pic_w + 63) / 64) * hc, total = ctus * n_pictures;
  ws_layout L;
  L.n_slots = (int)(total < MAX_SLOTS ? align_up(total, 32) : MAX_SLOTS);
  L.ticket = 0;
  L.slots = 256;
  L.done = L.slots + MAX_SLOTS / 8;
  L.hmvp = align_up(L.done + total * 4, 256);
  L.order = align_up(L.hmvp + (size_t)n_pictures * hc * 41 * 4, 256);        // [0, order): zeroed before every run
  L.pics = align_up(L.order + total * 4, 256);
  L.scratch = align_up(L.pics + (size_t)n_pictures * sizeof(pb_pic_dev), 256);
  L.total = L.scratch + (size_t)L.n_slots * sizeof(ctu::scratch);
  return L;
}

}  // namespace

#if defined(CTU_PROFILE)
// development builds: where the scratch slots (whose tails hold the phase counters) sit in the workspace
extern "C" __attribute__((visibility("default"))) size_t uvghip_ctu_search_pb_debug_scratch(int n_pictures, int pic_w, int pic_h, size_t *slot_bytes, int *n_slots)
{
  const ws_layout L = layout(n_pictures, pic_w, pic_h);
  *slot_bytes = sizeof(ctu::scratch); *n_slots = L.n_slots;
  return L.scratch;
}
#endif

extern "C" size_t uvghip_ctu_search_pb_workspace_bytes(int n_pictures, int pic_w, int pic_h)
{
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0) return 0;
  return layout(n_pictures, pic_w, pic_h).total;
}

extern "C" int uvghip_ctu_search_pb(int bitdepth, const uvghip_ctu_pb_picture_t *pictures, int n_pictures, void *workspace, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  static_assert(sizeof(uvghip_ctu_params_t) == sizeof(ctu::params), "uvghip_ctu_params_t mirrors ctu::params");
  if (!pictures || n_pictures <= 0 || !workspace || n_pictures > 32767) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const uvghip_ctu_params_t &p0 = pictures[0].params;
  if (p0.pic_w <= 0 || p0.pic_h <= 0 || (p0.pic_w & 7) || (p0.pic_h & 7) || p0.pic_w > 64 * 255 || p0.pic_h > 64 * 255)
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: picture size");
  const int wc = (p0.pic_w + 63) / 64, hc = (p0.pic_h + 63) / 64, ctus = wc * hc, total = ctus * n_pictures;
  const ws_layout L = layout(n_pictures, p0.pic_w, p0.pic_h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  std::vector<pb_pic_dev> pics(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_ctu_pb_picture_t &q = pictures[i];
    const uvghip_ctu_params_t &p = q.params;
    if (p.pic_w != p0.pic_w || p.pic_h != p0.pic_h) return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: the pictures of a call share one size");
    if (p.wpp != 1 || p.depth_min < 1 || p.depth_max != 4 || p.depth_min > p.depth_max || p.rough_levels < 2 || p.rough_levels > 3 || p.qp < 0 || p.qp > 63 ||
        p.qp_c < 0 || p.qp_c > 63 || !(p.lambda > 0) || !(p.lambda_sqrt > 0) || p.rd < 0 || p.rd > 1)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: configuration outside the supported subset");
    if ((q.slice_type != 0 && q.slice_type != 1) || q.n_refs < 1 || q.n_refs > 16 || q.l_size[0] < 1 || q.l_size[0] > 8 || q.l_size[1] < 0 || q.l_size[1] > 8 ||
        (q.slice_type == 1 && q.l_size[1] != 0) || q.depth_inter_min != 0 || q.depth_inter_max != 3 || q.max_merge < 5 || q.max_merge > 6 || q.fme_level < 0 ||
        q.fme_level > 4 || q.merge_level < 2)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: slice state outside the supported subset");
    const uvghip_ctu_picture_t &c = q.pic;
    if (!c.src_y || !c.src_u || !c.src_v || !c.rec_y || !c.rec_u || !c.rec_v || !c.cu || !c.coeff || !c.models || c.cu_stride < wc * 16 || !q.inter4 ||
        !q.models_inter || c.src_stride < p.pic_w || c.rec_stride < p.pic_w || c.src_stride_c < p.pic_w / 2 || c.rec_stride_c < p.pic_w / 2 ||
        q.ref_stride < p.pic_w || q.ref_stride_c < p.pic_w / 2 || q.ref_motion_stride < wc * 16)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: picture descriptor");
    pb_pic_dev &d = pics[i];
    memset(&d, 0, sizeof d);
    memcpy(&d.P, &p, sizeof d.P);
    ctu::pb_job &B = d.B;
    B.slice_type = q.slice_type; B.poc = q.poc; B.n_refs = q.n_refs;
    for (int k = 0; k < 16; ++k) {
      B.ref_pocs[k] = q.ref_pocs[k]; B.l[0][k] = q.l[0][k]; B.l[1][k] = q.l[1][k];
      if (k < q.n_refs && (!q.ref_y[k] || !q.ref_u[k] || !q.ref_v[k] || !q.ref_motion[k]))
        return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: a reference picture is missing");
      B.ref_y[k] = q.ref_y[k]; B.ref_u[k] = q.ref_u[k]; B.ref_v[k] = q.ref_v[k]; B.ref_cu[k] = q.ref_motion[k];
    }
    for (int l = 0; l < 2; ++l)
      for (int k = 0; k < q.l_size[l]; ++k)
        if (q.l[l][k] < 0 || q.l[l][k] >= q.n_refs) return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: a reference list entry is out of range");
    B.l_size[0] = q.l_size[0]; B.l_size[1] = q.l_size[1];
    B.tmvp = q.tmvp; B.max_merge = q.max_merge; B.merge_level = q.merge_level; B.frame_qp = q.frame_qp;
    B.bipred = q.bipred; B.fme_level = q.fme_level; B.early_skip = q.early_skip; B.depth_inter_min = q.depth_inter_min; B.depth_inter_max = q.depth_inter_max;
    B.ref_stride = q.ref_stride; B.ref_stride_c = q.ref_stride_c; B.ref_cu_stride = q.ref_motion_stride;
    if (q.inflight_margin < 0 || q.inflight_margin > 64) return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: inflight_margin");
    B.inflight_margin = q.inflight_margin;
    B.inter4 = q.inter4; B.trees = q.trees; B.motion_out = q.motion_out;
    B.hmvp_rows = reinterpret_cast<int32_t *>(ws + L.hmvp) + (size_t)i * hc * 41;
    d.src_y = c.src_y; d.src_u = c.src_u; d.src_v = c.src_v; d.rec_y = c.rec_y; d.rec_u = c.rec_u; d.rec_v = c.rec_v;
    d.cu = c.cu; d.coeff = c.coeff; d.models = c.models; d.models_inter = q.models_inter;
    d.src_stride = c.src_stride; d.src_stride_c = c.src_stride_c; d.rec_stride = c.rec_stride; d.rec_stride_c = c.rec_stride_c; d.cu_stride = c.cu_stride;
  }
  // the hand-out order and the pictures' descriptors, in stream order (nothing here waits for the stream: a caller with independent
  // pictures to issue -- api.LowDelayLoop.run(in_flight) -- is not held up by this one's references)
  hipStream_t st = uvghip_stream(stream);
  hipLaunchKernelGGL(pb_order_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<int32_t *>(ws + L.order), wc, hc, n_pictures);
  {
    const unsigned char *src = reinterpret_cast<const unsigned char *>(pics.data());
    const size_t bytes = (size_t)n_pictures * sizeof(pb_pic_dev);
    for (size_t at = 0; at < bytes; at += sizeof(pb_chunk)) {
      pb_chunk c;
      const int n = (int)(bytes - at < sizeof(pb_chunk) ? bytes - at : sizeof(pb_chunk));
      memcpy(c.b, src + at, (size_t)n);
      hipLaunchKernelGGL(pb_upload_kernel, dim3(1), dim3(256), 0, st, ws + L.pics + at, c, n);
    }
  }
  pb_launch_args A;
  A.pics = reinterpret_cast<const pb_pic_dev *>(ws + L.pics);
  A.order = reinterpret_cast<const int32_t *>(ws + L.order);
  A.ticket = reinterpret_cast<int32_t *>(ws + L.ticket);
  A.done = reinterpret_cast<int32_t *>(ws + L.done);
  A.scratch = reinterpret_cast<ctu::scratch *>(ws + L.scratch);
  A.slots = reinterpret_cast<uint32_t *>(ws + L.slots);
  A.n_slots = L.n_slots;
  A.wc = wc; A.hc = hc; A.n_ctus = total;
  const size_t lds = bitdepth == 8 ? sizeof(ctu::lds<uint8_t>) : sizeof(ctu::lds<uint16_t>);
  const hipError_t e = bitdepth == 8
      ? hipFuncSetAttribute(reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
      : hipFuncSetAttribute(reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return uvghip_set_error(e, "uvghip_ctu_search_pb: dynamic LDS size");
  UVGHIP_TRY(hipMemsetAsync(ws, 0, L.order, st));
  // workgroups: twice the CTUs a picture's wavefront can have in progress (the widest diagonal of cx + 2 cy), per picture
  const int width = (wc + 1) / 2 < hc ? (wc + 1) / 2 : hc;
  const long long want = 2LL * width * n_pictures;
  const int grid = (int)(want < total ? want : total);
  if (bitdepth == 8) hipLaunchKernelGGL(ctu_search_pb_kernel<uint8_t>, dim3(grid), dim3(64), lds, st, A);
  else hipLaunchKernelGGL(ctu_search_pb_kernel<uint16_t>, dim3(grid), dim3(64), lds, st, A);
  UVGHIP_CHECK_LAUNCH();
}
