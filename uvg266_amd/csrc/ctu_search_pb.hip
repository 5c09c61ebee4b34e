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
#include "ctu_filter.h"
#include <vector>
#include <cstring>
#include <cstdlib>
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
  // pictures in flight (uvghip_ctu_search_pb_inflight): the pictures of this call whose output this one reads, its depth in that DAG
  int n_wait, level;
  int wait_pic[16];
  // a picture whose SEARCH runs elsewhere (an I picture in the all-intra launch beside this call): its per-CTU "searched" flags; the
  // kernel waits for them and runs the picture's filter stage only
  const int32_t *ext_done;
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
  // pictures in flight: the filter stage of every picture (NULL: search only, the pictures of the call are independent), its flags
  const ctuf::filt_pic *fpics;
  int32_t *sao_done, *final_done;   // [pic * ctus + cy * wc + cx]
  unsigned long long *times;        // CTU_PROFILE builds: per CTU the wall clock (100 MHz, the same on every CU) at ticket, start of the search, end of the search, end of the filters
};

// four workgroups (= four waves: the kernel's registers allow one per SIMD) per CU at 8 bit: 160 KB / 4 incl. the 4288 bytes of static tables
static_assert(ctu::pb_lds_bytes<uint8_t>(1) + 4288 <= 40960, "the 8-bit LDS image of a P / B CTU no longer fits four workgroups per CU");
// NT = 64: one wave walks the CTU (four workgroups per CU: many independent pictures side by side).  NT = 128: the walk's wave + the leaf
// wave that takes the 4x4 CUs of every 8x8 area (ctu_pb.h post_leaves) -- a shorter CTU for pictures in flight behind each other, where
// the CTU's latency, not the device's occupancy, sets the pace.
static_assert(sizeof(ctuf::filt_lds<uint8_t>) <= ctu::pb_lds_bytes<uint8_t>(1) && sizeof(ctuf::filt_lds<uint16_t>) <= ctu::pb_lds_bytes<uint16_t>(1), "the filter job works inside the CTU's LDS image");
// NT = 192: ... + the depth wave that evaluates the 32x32 / 16x16 CUs while the walk is in their children (ctu_pb.h post_eval_pb).
// NT = 256: ... two depth waves, one for the 16x16 CUs and one for the 32x32 CUs -- a CU, four SIMDs, one CTU.
template <typename PX, int NT>
__global__ void __launch_bounds__(NT) ctu_search_pb_kernel(pb_launch_args A)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ctu::lds<PX> *S = reinterpret_cast<ctu::lds<PX> *>(smem);
  __shared__ int s_ticket;
  __shared__ int s_slot;
#if defined(CTU_POISON_LDS)
  for (unsigned i = threadIdx.x; i < ctu::pb_lds_bytes<PX>(NT / 64); i += NT) smem[i] = (unsigned char)(CTU_POISON_LDS);
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
#if defined(CTU_PROFILE)
    const unsigned long long t_ticket = wall_clock64();
#endif
    const int32_t o = A.order[ticket];
    const int pic = o >> 16, cy = (o >> 8) & 0xff, cx = o & 0xff;
    const int ctus = A.wc * A.hc, k = cy * A.wc + cx;
    int32_t *done = A.done + (size_t)pic * ctus;
    const int32_t *const ext = A.fpics ? A.pics[pic].ext_done : nullptr;
    if (threadIdx.x == 0) {
      int naps = 1;
      // (an externally searched picture: that launch's own order has the CTU's neighbours done before the CTU)
      const int32_t *deps[3] = {ext ? &ext[k] : (cx > 0 ? &done[k - 1] : nullptr), !ext && cy > 0 ? &done[k - A.wc] : nullptr,
                                !ext && cy > 0 && cx + 1 < A.wc ? &done[k - A.wc + 1] : nullptr};
      for (int d = 0; d < 3; ++d)
        while (deps[d] && __hip_atomic_load(deps[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
          if (naps < 8) naps <<= 1;
        }
      // a picture in flight behind its references.  The reference encoder lets CTU (x, y) wait for CTU (x + 2, y + 1) of the reference
      // (encoderstate.c:1103-1112, max_inter_ref_lcu = {1, 1}); what the vectors can actually reach under fracmv_within_tile (mv_within:
      // ly <= 1, lx + ly <= 2, the filters' delay in the margin) is final earlier: the staircase of CTUs (x + 2 + j, y - j), j >= 0, and
      // CTU (x + 1, y + 1), with everything left of and above them.  final_done is monotone over exactly that shape (a CTU's flag is
      // set after its left, upper and upper-right neighbour's, ctu_filter.h), so ONE flag says it: (x + 1, y + 1), or (x + 2, y) in the
      // last row.  Same pictures, same stream -- a shorter wait: a picture follows its reference four diagonals behind, not five.
      if (A.fpics) {
        const pb_pic_dev &Dw = A.pics[pic];
        const int kd = cy + 1 < A.hc ? (cy + 1) * A.wc + (cx + 1 < A.wc ? cx + 1 : A.wc - 1) : cy * A.wc + (cx + 2 < A.wc ? cx + 2 : A.wc - 1);
        for (int r = 0; r < Dw.n_wait; ++r) ctuf::wait_set(A.final_done + (size_t)Dw.wait_pic[r] * ctus + kd);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
#if defined(CTU_PROFILE)
    const unsigned long long t_start = wall_clock64();
#endif
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
    if (!ext) ctu::run_ctu_pb(S, J);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&done[k], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#if defined(CTU_PROFILE)
    const unsigned long long t_searched = wall_clock64();
#endif
    if (A.fpics) {
      // the CTU's in-loop filters while its right and lower neighbours search on (ctu_filter.h); the LDS image is free until the next CTU
      ctuf::filt_ctu F;
      F.rec_y = D.rec_y; F.rec_u = D.rec_u; F.rec_v = D.rec_v; F.src_y = D.src_y; F.src_u = D.src_u; F.src_v = D.src_v;
      F.rec_stride = D.rec_stride; F.rec_stride_c = D.rec_stride_c; F.src_stride = D.src_stride; F.src_stride_c = D.src_stride_c;
      F.scu = D.cu; F.scu_stride = D.cu_stride;
      F.W = D.P.pic_w; F.H = D.P.pic_h; F.cx = cx; F.cy = cy; F.wc = A.wc; F.hc = A.hc;
      F.sao_done = A.sao_done + (size_t)pic * ctus; F.final_done = A.final_done + (size_t)pic * ctus;
      ctuf::filter_ctu<PX>(smem, A.fpics[pic], F);
      if (threadIdx.x == 0) S->rot = 0;
    }
#if defined(CTU_PROFILE)
    if (threadIdx.x == 0) {
      unsigned long long *t = A.times + ((size_t)pic * ctus + k) * 4;
      t[0] = t_ticket; t[1] = t_start; t[2] = t_searched; t[3] = wall_clock64();
    }
#endif
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
// ... for pictures in flight: the key of a CTU is its index cx + 2 cy plus LAG times its picture's depth in the call's reference DAG.  A CTU
// waits for CTU (x + 1, y + 1) of the pictures it reads -- index + 3, depth at least one less: a smaller key, so every wait is for a
// smaller ticket (no deadlock at any grid size).  cnt / cur: [keys] zeroed; any order inside a key will do.  One block.
enum { LAG = 4 };
__global__ void __launch_bounds__(256) pb_order_levels_kernel(int32_t *order, int32_t *cnt, int32_t *cur, const pb_pic_dev *pics, int wc, int hc, int n_pictures, int n_keys)
{
  const int nd = wc + 2 * (hc - 1);
  for (int p = threadIdx.x; p < n_pictures; p += blockDim.x) {
    const int lv = pics[p].level;
    for (int d = 0; d < nd; ++d) {
      const int lo = d - (wc - 1) > 0 ? (d - (wc - 1) + 1) / 2 : 0, hi = d / 2 < hc - 1 ? d / 2 : hc - 1;
      if (hi >= lo) atomicAdd(&cnt[d + LAG * lv], hi - lo + 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int at = 0;
    for (int k = 0; k < n_keys; ++k) { const int c = cnt[k]; cnt[k] = at; at += c; }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < n_pictures; p += blockDim.x) {
    const int lv = pics[p].level;
    for (int d = 0; d < nd; ++d) {
      const int lo = d - (wc - 1) > 0 ? (d - (wc - 1) + 1) / 2 : 0, hi = d / 2 < hc - 1 ? d / 2 : hc - 1;
      if (hi < lo) continue;
      const int key = d + LAG * lv, at = cnt[key] + atomicAdd(&cur[key], hi - lo + 1);
      for (int cy = lo; cy <= hi; ++cy) order[at + cy - lo] = p << 16 | cy << 8 | (d - 2 * cy);
    }
  }
}
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
enum { MAX_SLOTS = 2048 };
struct ws_layout { size_t ticket, slots, done, hmvp, sao_done, final_done, key_cnt, key_cur, order, pics, filt, times, scratch, total; int n_slots, n_keys; };
ws_layout layout(int n_pictures, int pic_w, int pic_h)
{
  const size_t wc = (size_t)((pic_w + 63) / 64), hc = (size_t)((pic_h + 63) / 64), ctus = wc * hc, total = ctus * n_pictures;
  ws_layout L;
  L.n_slots = (int)(total < MAX_SLOTS ? align_up(total, 32) : MAX_SLOTS);
  L.n_keys = (int)(wc + 2 * (hc - 1)) + LAG * n_pictures;
  L.ticket = 0;
  L.slots = 256;
  L.done = L.slots + MAX_SLOTS / 8;
  L.hmvp = align_up(L.done + total * 4, 256);
  L.sao_done = align_up(L.hmvp + (size_t)n_pictures * hc * 41 * 4, 256);
  L.final_done = L.sao_done + total * 4;
  L.key_cnt = L.final_done + total * 4;
  L.key_cur = L.key_cnt + (size_t)L.n_keys * 4;
  L.order = align_up(L.key_cur + (size_t)L.n_keys * 4, 256);                  // [0, order): zeroed before every run
  L.pics = align_up(L.order + total * 4, 256);
  L.filt = align_up(L.pics + (size_t)n_pictures * sizeof(pb_pic_dev), 256);
  L.times = align_up(L.filt + (size_t)n_pictures * sizeof(ctuf::filt_pic), 256);
#if defined(CTU_PROFILE)
  L.scratch = align_up(L.times + total * 32, 256);
#else
  L.scratch = L.times;
#endif
  L.total = L.scratch + (size_t)L.n_slots * sizeof(ctu::scratch);
  return L;
}

}  // namespace

// where the in-flight launch raises a picture's per-CTU "final" flags (picture i at [i * ctus]): a consumer of the finished pictures that runs
// beside the launch waits on them (uvghip_loop_plan_run_coder_behind).  The launch zeroes them in stream order before its kernel.
extern "C" const int32_t *uvghip_ctu_search_pb_inflight_final_flags(int n_pictures, int pic_w, int pic_h, const void *workspace)
{
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !workspace) return nullptr;
  return reinterpret_cast<const int32_t *>(static_cast<const unsigned char *>(workspace) + layout(n_pictures, pic_w, pic_h).final_done);
}

#if defined(CTU_PROFILE)
// development builds: where the scratch slots (whose tails hold the phase counters) sit in the workspace
extern "C" __attribute__((visibility("default"))) size_t uvghip_ctu_search_pb_debug_scratch(int n_pictures, int pic_w, int pic_h, size_t *slot_bytes, int *n_slots)
{
  const ws_layout L = layout(n_pictures, pic_w, pic_h);
  *slot_bytes = sizeof(ctu::scratch); *n_slots = L.n_slots;
  return L.scratch;
}
// ... and the per-CTU timestamps [picture][ctu][4] (uint64 ticks of the 100 MHz wall clock: ticket taken, search started, search done, filters done)
extern "C" __attribute__((visibility("default"))) size_t uvghip_ctu_search_pb_debug_times(int n_pictures, int pic_w, int pic_h) { return layout(n_pictures, pic_w, pic_h).times; }
#endif

extern "C" size_t uvghip_ctu_search_pb_workspace_bytes(int n_pictures, int pic_w, int pic_h)
{
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0) return 0;
  return layout(n_pictures, pic_w, pic_h).total;
}

namespace {
int search_pb(int bitdepth, const uvghip_ctu_pb_picture_t *pictures, int n_pictures, const uvghip_pb_filter_t *filters, const int32_t *ref_in_call,
              const int32_t *const *searched_flags, int other_workgroups, void *workspace, void *stream);
}
extern "C" int uvghip_ctu_search_pb(int bitdepth, const uvghip_ctu_pb_picture_t *pictures, int n_pictures, void *workspace, void *stream)
{
  return search_pb(bitdepth, pictures, n_pictures, nullptr, nullptr, nullptr, 0, workspace, stream);
}
extern "C" size_t uvghip_ctu_search_pb_inflight_workspace_bytes(int n_pictures, int pic_w, int pic_h) { return uvghip_ctu_search_pb_workspace_bytes(n_pictures, pic_w, pic_h); }
extern "C" int uvghip_ctu_search_pb_inflight(int bitdepth, const uvghip_ctu_pb_picture_t *pictures, const uvghip_pb_filter_t *filters, const int32_t *ref_in_call,
                                             int n_pictures, void *workspace, void *stream)
{
  if (!filters || !ref_in_call) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return search_pb(bitdepth, pictures, n_pictures, filters, ref_in_call, nullptr, 0, workspace, stream);
}
extern "C" int uvghip_ctu_search_pb_inflight_ext(int bitdepth, const uvghip_ctu_pb_picture_t *pictures, const uvghip_pb_filter_t *filters, const int32_t *ref_in_call,
                                                 const int32_t *const *searched_flags, int other_workgroups, int n_pictures, void *workspace, void *stream)
{
  if (!filters || !ref_in_call || !searched_flags || other_workgroups < 0 || other_workgroups > 512) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return search_pb(bitdepth, pictures, n_pictures, filters, ref_in_call, searched_flags, other_workgroups, workspace, stream);
}

namespace {
int search_pb(int bitdepth, const uvghip_ctu_pb_picture_t *pictures, int n_pictures, const uvghip_pb_filter_t *filters, const int32_t *ref_in_call,
              const int32_t *const *searched_flags, int other_workgroups, void *workspace, void *stream)
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
  std::vector<ctuf::filt_pic> filt(filters ? n_pictures : 0);
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_ctu_pb_picture_t &q = pictures[i];
    const uvghip_ctu_params_t &p = q.params;
    if (p.pic_w != p0.pic_w || p.pic_h != p0.pic_h) return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: the pictures of a call share one size");
    if (p.wpp != 1 || p.depth_min < 1 || p.depth_max != 4 || p.depth_min > p.depth_max || p.rough_levels < 2 || p.rough_levels > 3 || p.qp < 0 || p.qp > 63 ||
        p.qp_c < 0 || p.qp_c > 63 || !(p.lambda > 0) || !(p.lambda_sqrt > 0) || p.rd < 0 || p.rd > 1)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb: configuration outside the supported subset");
    if (searched_flags && searched_flags[i]) {
      // a picture searched elsewhere (an I picture in uvghip_ctu_plan_launch beside this call): the filter stage only, behind that launch's flags
      const uvghip_ctu_picture_t &c = q.pic;
      if (!filters || !c.src_y || !c.src_u || !c.src_v || !c.rec_y || !c.rec_u || !c.rec_v || !c.cu || c.cu_stride < wc * 16 || c.src_stride < p.pic_w || c.rec_stride < p.pic_w ||
          c.src_stride_c < p.pic_w / 2 || c.rec_stride_c < p.pic_w / 2 || q.slice_type != 2 || p.qp_c != p.qp)
        return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb_inflight_ext: an externally searched picture");
      const uvghip_pb_filter_t &f = filters[i];
      if (!f.dbk_y || !f.dbk_u || !f.dbk_v || !f.out_y || !f.out_u || !f.out_v || f.dbk_stride < p.pic_w || f.dbk_stride_c < p.pic_w / 2 || f.out_stride < p.pic_w ||
          f.out_stride_c < p.pic_w / 2 || f.sao_type < 0 || f.sao_type > 3 || (f.sao_type && (!f.sao_info || !f.sao_models)))
        return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb_inflight_ext: filter stage");
      pb_pic_dev &d = pics[i];
      memset(&d, 0, sizeof d);
      memcpy(&d.P, &p, sizeof d.P);
      d.src_y = c.src_y; d.src_u = c.src_u; d.src_v = c.src_v; d.rec_y = c.rec_y; d.rec_u = c.rec_u; d.rec_v = c.rec_v; d.cu = c.cu;
      d.src_stride = c.src_stride; d.src_stride_c = c.src_stride_c; d.rec_stride = c.rec_stride; d.rec_stride_c = c.rec_stride_c; d.cu_stride = c.cu_stride;
      d.ext_done = searched_flags[i];
      ctuf::filt_pic &g = filt[i];
      g.dbk_y = f.dbk_y; g.dbk_u = f.dbk_u; g.dbk_v = f.dbk_v; g.out_y = f.out_y; g.out_u = f.out_u; g.out_v = f.out_v;
      g.dbk_stride = f.dbk_stride; g.dbk_stride_c = f.dbk_stride_c; g.out_stride = f.out_stride; g.out_stride_c = f.out_stride_c;
      g.sao_info = f.sao_info; g.sao_models = f.sao_models; g.lambda = p.lambda; g.sao_type = f.sao_type; g.slice_type = 2; g.qp = p.qp; g.is_b = 0;
      continue;
    }
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
    if (filters) {
      // the picture's references inside this call (pictures are in coding order: a reference is an earlier entry), its depth in that DAG
      const uvghip_pb_filter_t &f = filters[i];
      if (!f.dbk_y || !f.dbk_u || !f.dbk_v || !f.out_y || !f.out_u || !f.out_v || f.dbk_stride < p.pic_w || f.dbk_stride_c < p.pic_w / 2 || f.out_stride < p.pic_w ||
          f.out_stride_c < p.pic_w / 2 || f.sao_type < 0 || f.sao_type > 3 || (f.sao_type && (!f.sao_info || !f.sao_models)))
        return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb_inflight: filter stage");
      if (p.qp_c != p.qp || p.qp != q.frame_qp) return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb_inflight: qp_c != qp or params.qp != frame_qp");
      d.n_wait = 0; d.level = 0;
      for (int k = 0; k < q.n_refs; ++k) {
        const int r = ref_in_call[(size_t)i * 16 + k];
        if (r < 0) continue;
        if (r >= i) return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb_inflight: a reference inside the call must be an earlier picture of it");
        const uvghip_pb_filter_t &fr = filters[r];
        const bool r_ext = searched_flags && searched_flags[r];          // (an I picture: its motion table is the caller's constant "intra everywhere")
        if (q.ref_y[k] != fr.out_y || q.ref_u[k] != fr.out_u || q.ref_v[k] != fr.out_v || (!r_ext && q.ref_motion[k] != pictures[r].motion_out) || q.ref_stride != fr.out_stride ||
            q.ref_stride_c != fr.out_stride_c)
          return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb_inflight: ref_in_call names a picture whose output is not this reference");
        // a reference still being coded: the vectors must stay inside what is final in it (fracmv_within_tile's margin: 1 + the filters' delay)
        if (q.inflight_margin != (fr.sao_type ? 11 : 9)) return uvghip_set_error(hipErrorInvalidValue, "uvghip_ctu_search_pb_inflight: inflight_margin must be 11 (SAO) / 9 with a reference in flight");
        bool seen = false;
        for (int j = 0; j < d.n_wait; ++j) seen = seen || d.wait_pic[j] == r;
        if (!seen) d.wait_pic[d.n_wait++] = r;
        if (pics[r].level + 1 > d.level) d.level = pics[r].level + 1;
      }
      ctuf::filt_pic &g = filt[i];
      g.dbk_y = f.dbk_y; g.dbk_u = f.dbk_u; g.dbk_v = f.dbk_v; g.out_y = f.out_y; g.out_u = f.out_u; g.out_v = f.out_v;
      g.dbk_stride = f.dbk_stride; g.dbk_stride_c = f.dbk_stride_c; g.out_stride = f.out_stride; g.out_stride_c = f.out_stride_c;
      g.sao_info = f.sao_info; g.sao_models = f.sao_models; g.lambda = p.lambda; g.sao_type = f.sao_type; g.slice_type = q.slice_type; g.qp = p.qp; g.is_b = q.slice_type == 0;
    }
  }
  // the hand-out order and the pictures' descriptors, in stream order (nothing here waits for the stream: a caller with independent
  // pictures to issue -- api.LowDelayLoop.run(in_flight) -- is not held up by this one's references)
  hipStream_t st = uvghip_stream(stream);
  UVGHIP_TRY(hipMemsetAsync(ws, 0, L.order, st));
  if (int rc = uvghip_upload_ordered(ws + L.pics, pics.data(), (size_t)n_pictures * sizeof(pb_pic_dev), st)) return rc;
  if (filters) {
    if (int rc = uvghip_upload_ordered(ws + L.filt, filt.data(), (size_t)n_pictures * sizeof(ctuf::filt_pic), st)) return rc;
    hipLaunchKernelGGL(pb_order_levels_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<int32_t *>(ws + L.order), reinterpret_cast<int32_t *>(ws + L.key_cnt),
                       reinterpret_cast<int32_t *>(ws + L.key_cur), reinterpret_cast<const pb_pic_dev *>(ws + L.pics), wc, hc, n_pictures, L.n_keys);
  } else hipLaunchKernelGGL(pb_order_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<int32_t *>(ws + L.order), wc, hc, n_pictures);
  pb_launch_args A;
  A.pics = reinterpret_cast<const pb_pic_dev *>(ws + L.pics);
  A.order = reinterpret_cast<const int32_t *>(ws + L.order);
  A.ticket = reinterpret_cast<int32_t *>(ws + L.ticket);
  A.done = reinterpret_cast<int32_t *>(ws + L.done);
  A.scratch = reinterpret_cast<ctu::scratch *>(ws + L.scratch);
  A.slots = reinterpret_cast<uint32_t *>(ws + L.slots);
  A.n_slots = L.n_slots;
  A.wc = wc; A.hc = hc; A.n_ctus = total;
  A.fpics = filters ? reinterpret_cast<const ctuf::filt_pic *>(ws + L.filt) : nullptr;
  A.sao_done = reinterpret_cast<int32_t *>(ws + L.sao_done);
  A.final_done = reinterpret_cast<int32_t *>(ws + L.final_done);
  A.times = reinterpret_cast<unsigned long long *>(ws + L.times);
  // pictures in flight: three waves per CTU (UVGHIP_PB_WAVES=1 / 2 / 3 overrides, development)
  int waves = filters ? 4 : 1;
  if (const char *e = getenv("UVGHIP_PB_WAVES")) waves = e[0] >= '1' && e[0] <= '4' ? e[0] - '0' : waves;
  const size_t lds = bitdepth == 8 ? ctu::pb_lds_bytes<uint8_t>(waves) : ctu::pb_lds_bytes<uint16_t>(waves);
  const void *fn8[4] = {reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint8_t, 64>), reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint8_t, 128>),
                        reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint8_t, 192>), reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint8_t, 256>)};
  const void *fn10[4] = {reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint16_t, 64>), reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint16_t, 128>),
                         reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint16_t, 192>), reinterpret_cast<const void *>(&ctu_search_pb_kernel<uint16_t, 256>)};
  const hipError_t e = hipFuncSetAttribute(bitdepth == 8 ? fn8[waves - 1] : fn10[waves - 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return uvghip_set_error(e, "uvghip_ctu_search_pb: dynamic LDS size");
  // workgroups: twice the CTUs a picture's wavefront can have in progress (the widest diagonal of cx + 2 cy), per picture.  Pictures in
  // flight behind each other: a picture follows its reference LAG diagonals later, so a chain holds a picture's CTUs / LAG in progress
  // whatever its length -- a device's worth of workgroups is plenty (the rest would only wait)
  const int width = (wc + 1) / 2 < hc ? (wc + 1) / 2 : hc;
  long long want = 2LL * width * n_pictures;
  if (filters && want > (waves >= 3 ? 256 : 1024 / waves)) want = waves >= 3 ? 256 : 1024 / waves;
  // a launch that runs BESIDE this one and is waited for (externally searched pictures): this launch's workgroups take whole CUs and
  // must leave that one its own -- four of its workgroups fit a CU
  if (searched_flags && other_workgroups > 0 && waves >= 3) { const int room = 256 - (other_workgroups + 3) / 4; if (want > room) want = room > 16 ? room : 16; }
  const int grid = (int)(want < total ? want : total);
  if (bitdepth == 8) {
    if (waves == 4) hipLaunchKernelGGL((ctu_search_pb_kernel<uint8_t, 256>), dim3(grid), dim3(256), lds, st, A);
    else if (waves == 3) hipLaunchKernelGGL((ctu_search_pb_kernel<uint8_t, 192>), dim3(grid), dim3(192), lds, st, A);
    else if (waves == 2) hipLaunchKernelGGL((ctu_search_pb_kernel<uint8_t, 128>), dim3(grid), dim3(128), lds, st, A);
    else hipLaunchKernelGGL((ctu_search_pb_kernel<uint8_t, 64>), dim3(grid), dim3(64), lds, st, A);
  } else {
    if (waves == 4) hipLaunchKernelGGL((ctu_search_pb_kernel<uint16_t, 256>), dim3(grid), dim3(256), lds, st, A);
    else if (waves == 3) hipLaunchKernelGGL((ctu_search_pb_kernel<uint16_t, 192>), dim3(grid), dim3(192), lds, st, A);
    else if (waves == 2) hipLaunchKernelGGL((ctu_search_pb_kernel<uint16_t, 128>), dim3(grid), dim3(128), lds, st, A);
    else hipLaunchKernelGGL((ctu_search_pb_kernel<uint16_t, 64>), dim3(grid), dim3(64), lds, st, A);
  }
  UVGHIP_CHECK_LAUNCH();
}
}  // namespace
