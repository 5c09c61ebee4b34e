// uvghip_filter_pictures: the in-loop filters of a whole group of searched pictures in ONE launch -- a workgroup per CTU runs the per-CTU
// filter stage of ctu_filter.h (deblocking of what the CTU completes, its SAO statistics and decision, SAO into the output picture: what
// encoder_state_worker_encode_lcu_search does after uvg_search_lcu, src/encoderstate.c:841-853, with the pictures' reconstruction left
// unfiltered).  It replaces the chain of whole-picture kernels the loop plans strung behind the search (per picture: three snapshot
// copies, two snapshot deblocking passes, three SAO statistics launches, two deblocking passes, three SAO apply launches; per group the
// two decision kernels: ~9 000 launches for 224 pictures, each taking turns with the other group's search on a full device) -- the
// same pictures, decisions and models (tests/test_gpu_sao_decide.py, the whole-picture goldens).
//
// The stage is the P / B kernel's (where it runs inside the persistent search kernel, pictures in flight); here it is a kernel of its
// own because the I-picture search kernel is built for four workgroups per CU at 128 registers, under which the deblocking segment
// functions spill (3.8 M cycles per CTU inside that kernel against 0.17 M here).  CTUs are handed out in wavefront order (a ticket per
// workgroup), so the SAO decision's chain -- a CTU reads the models its left neighbour left, the first CTU of a row those of the first
// CTU of the row above, and both neighbours' decisions as merge candidates -- only ever waits for CTUs that are running or done.
#include "uvghip_common.h"
#define CTUF_CALL __attribute__((always_inline))
#include "ctu_filter.h"
#include <vector>

namespace {

struct fpic_dev {
  ctuf::filt_pic F;
  const void *rec_y, *rec_u, *rec_v, *src_y, *src_u, *src_v;
  const uvghip_scu_t *scu;
  int32_t rec_stride, rec_stride_c, src_stride, src_stride_c, scu_stride, pad;
};

struct filter_args {
  const fpic_dev *pics;
  int32_t *ticket, *sao_done, *final_done;
  int wc, hc, n_pictures, W, H;
  const int32_t *searched;      // BEHIND: [picture][ctu], the search's "done" flags (uvghip_ctu_plan_done_flags)
};

// BEHIND: a small grid of persistent workgroups that take CTU after CTU and wait for the SEARCH's flag of each -- the filters of a group
// BESIDE its search launch (uvghip_loop_plan_run_overlapped), a CTU filtered as soon as it is searched instead of after the last CTU of the
// launch.  The grid is capped by the caller so that the search, launched first, keeps the device: a waiting workgroup holds its slot.
template <typename PX, bool BEHIND>
__global__ void __launch_bounds__(256) ctu_filter_kernel(filter_args A)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_ticket;
  for (;;) {
  if (threadIdx.x == 0) s_ticket = atomicAdd(A.ticket, 1);
  __syncthreads();
  // ticket -> (diagonal cx + cy, picture, row): every CTU comes after its left and upper neighbour; pictures interleaved
  const int wc = A.wc, hc = A.hc, ctus = wc * hc;
  if (BEHIND && s_ticket >= ctus * A.n_pictures) break;
  int t = s_ticket, d = 0;
  for (;; ++d) {
    const int lo = d - (wc - 1) > 0 ? d - (wc - 1) : 0, hi = d < hc - 1 ? d : hc - 1, cnt = (hi - lo + 1) * A.n_pictures;
    if (t < cnt) break;
    t -= cnt;
  }
  const int lo = d - (wc - 1) > 0 ? d - (wc - 1) : 0, hi = d < hc - 1 ? d : hc - 1, per = hi - lo + 1;
  const int pic = t / per, cy = lo + (t - pic * per), cx = d - cy;
  const fpic_dev &D = A.pics[pic];
  ctuf::filt_ctu F;
  F.rec_y = D.rec_y; F.rec_u = D.rec_u; F.rec_v = D.rec_v; F.src_y = D.src_y; F.src_u = D.src_u; F.src_v = D.src_v;
  F.rec_stride = D.rec_stride; F.rec_stride_c = D.rec_stride_c; F.src_stride = D.src_stride; F.src_stride_c = D.src_stride_c;
  F.scu = D.scu; F.scu_stride = D.scu_stride;
  F.W = A.W; F.H = A.H; F.cx = cx; F.cy = cy; F.wc = wc; F.hc = hc;
  F.sao_done = A.sao_done + (size_t)pic * ctus; F.final_done = A.final_done + (size_t)pic * ctus;
  if (BEHIND) {
    // the CTU's own flag: its left / upper / upper-left neighbours, whose samples and side information the stage reads too, were searched before it
    if (threadIdx.x == 0) { ctuf::wait_set(&A.searched[(size_t)pic * ctus + cy * wc + cx]); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    __syncthreads();
  }
  ctuf::filter_ctu<PX>(smem, D.F, F);
  if (!BEHIND) break;
  __syncthreads();          // the LDS image and s_ticket are free again
  }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct fl_layout { size_t ticket, sao_done, final_done, pics, total; };
fl_layout layout(int n, int w, int h)
{
  const size_t ctus = (size_t)((w + 63) / 64) * ((h + 63) / 64), total = ctus * n;
  fl_layout L;
  L.ticket = 0;
  L.sao_done = 256;
  L.final_done = L.sao_done + total * 4;
  L.pics = align_up(L.final_done + total * 4, 256);          // [0, pics): zeroed before every run
  L.total = L.pics + (size_t)n * sizeof(fpic_dev);
  return L;
}

}  // namespace

extern "C" size_t uvghip_filter_pictures_workspace_bytes(int n_pictures, int pic_w, int pic_h)
{
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0) return 0;
  return layout(n_pictures, pic_w, pic_h).total;
}

// The picture table is written once per workspace (prepare: synchronous copy, as uvghip_slice_rows_prepare), a run is a memset of the
// flags and one launch.
extern "C" int uvghip_filter_pictures_prepare(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_ctu_picture_t *pictures, const uvghip_pb_filter_t *filters,
                                              int n_pictures, int slice_type, void *workspace)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || !pictures || !filters || n_pictures <= 0 || !workspace || slice_type < 0 || slice_type > 2) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int w = params->pic_w, h = params->pic_h;
  if (w <= 0 || h <= 0 || (w & 7) || (h & 7)) return uvghip_set_error(hipErrorInvalidValue, "uvghip_filter_pictures_prepare: picture size");
  if (params->qp_c != params->qp) return uvghip_set_error(hipErrorInvalidValue, "uvghip_filter_pictures_prepare: qp_c != qp needs a chroma QP table");
  const int wc = (w + 63) / 64;
  std::vector<fpic_dev> pd(n_pictures);
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_ctu_picture_t &p = pictures[i];
    const uvghip_pb_filter_t &f = filters[i];
    if (!p.src_y || !p.src_u || !p.src_v || !p.rec_y || !p.rec_u || !p.rec_v || !p.cu || p.cu_stride < wc * 16 || p.src_stride < w || p.rec_stride < w ||
        p.src_stride_c < w / 2 || p.rec_stride_c < w / 2)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_filter_pictures_prepare: picture descriptor");
    if (!f.dbk_y || !f.dbk_u || !f.dbk_v || !f.out_y || !f.out_u || !f.out_v || f.dbk_stride < w || f.dbk_stride_c < w / 2 || f.out_stride < w || f.out_stride_c < w / 2 ||
        f.sao_type < 0 || f.sao_type > 3 || (f.sao_type && (!f.sao_info || !f.sao_models)))
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_filter_pictures_prepare: filter stage");
    fpic_dev &d = pd[i];
    ctuf::filt_pic &g = d.F;
    g.dbk_y = f.dbk_y; g.dbk_u = f.dbk_u; g.dbk_v = f.dbk_v; g.out_y = f.out_y; g.out_u = f.out_u; g.out_v = f.out_v;
    g.dbk_stride = f.dbk_stride; g.dbk_stride_c = f.dbk_stride_c; g.out_stride = f.out_stride; g.out_stride_c = f.out_stride_c;
    g.sao_info = f.sao_info; g.sao_models = f.sao_models; g.lambda = params->lambda; g.sao_type = f.sao_type; g.slice_type = slice_type; g.qp = params->qp;
    g.is_b = slice_type == 0;
    d.rec_y = p.rec_y; d.rec_u = p.rec_u; d.rec_v = p.rec_v; d.src_y = p.src_y; d.src_u = p.src_u; d.src_v = p.src_v; d.scu = p.cu;
    d.rec_stride = p.rec_stride; d.rec_stride_c = p.rec_stride_c; d.src_stride = p.src_stride; d.src_stride_c = p.src_stride_c; d.scu_stride = p.cu_stride; d.pad = 0;
  }
  const fl_layout L = layout(n_pictures, w, h);
  UVGHIP_TRY(hipMemcpy(static_cast<unsigned char *>(workspace) + L.pics, pd.data(), pd.size() * sizeof(fpic_dev), hipMemcpyHostToDevice));
  return 0;
}

// The run in two halves for a caller that lets the stage run BESIDE the search that feeds it (and a coder behind the stage's own flags):
// reset -- ticket and flags to zero, in stream order -- then uvghip_filter_pictures_run_behind on a stream that waits for the reset: at most
// max_workgroups persistent workgroups, each CTU waiting for searched[picture][ctu] (the search plan's flags).
extern "C" int uvghip_filter_pictures_reset(int n_pictures, int pic_w, int pic_h, void *workspace, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !workspace) return uvghip_set_error(hipErrorInvalidValue, __func__);
  UVGHIP_TRY(hipMemsetAsync(workspace, 0, layout(n_pictures, pic_w, pic_h).pics, uvghip_stream(stream)));
  return 0;
}
// ... the stage's per-CTU "final" flags [picture][ctu] (DEVICE memory): 1 when the CTU's part of the output picture and its SAO decision are published
extern "C" const int32_t *uvghip_filter_pictures_final_flags(int n_pictures, int pic_w, int pic_h, const void *workspace)
{
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !workspace) return nullptr;
  return reinterpret_cast<const int32_t *>(static_cast<const unsigned char *>(workspace) + layout(n_pictures, pic_w, pic_h).final_done);
}
extern "C" int uvghip_filter_pictures_run_behind(int bitdepth, int n_pictures, int pic_w, int pic_h, void *workspace, const int32_t *searched, int max_workgroups, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !workspace || !searched || max_workgroups < 1) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const fl_layout L = layout(n_pictures, pic_w, pic_h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  hipStream_t st = uvghip_stream(stream);
  filter_args A;
  A.pics = reinterpret_cast<const fpic_dev *>(ws + L.pics);
  A.ticket = reinterpret_cast<int32_t *>(ws + L.ticket);
  A.sao_done = reinterpret_cast<int32_t *>(ws + L.sao_done);
  A.final_done = reinterpret_cast<int32_t *>(ws + L.final_done);
  A.wc = (pic_w + 63) / 64; A.hc = (pic_h + 63) / 64; A.n_pictures = n_pictures; A.W = pic_w; A.H = pic_h;
  A.searched = searched;
  const int total = A.wc * A.hc * n_pictures, grid = total < max_workgroups ? total : max_workgroups;
  if (bitdepth == 8) hipLaunchKernelGGL((ctu_filter_kernel<uint8_t, true>), dim3(grid), dim3(256), sizeof(ctuf::filt_lds<uint8_t>), st, A);
  else hipLaunchKernelGGL((ctu_filter_kernel<uint16_t, true>), dim3(grid), dim3(256), sizeof(ctuf::filt_lds<uint16_t>), st, A);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_filter_pictures_run(int bitdepth, int n_pictures, int pic_w, int pic_h, void *workspace, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !workspace) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const fl_layout L = layout(n_pictures, pic_w, pic_h);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  hipStream_t st = uvghip_stream(stream);
  UVGHIP_TRY(hipMemsetAsync(ws, 0, L.pics, st));
  filter_args A;
  A.searched = nullptr;
  A.pics = reinterpret_cast<const fpic_dev *>(ws + L.pics);
  A.ticket = reinterpret_cast<int32_t *>(ws + L.ticket);
  A.sao_done = reinterpret_cast<int32_t *>(ws + L.sao_done);
  A.final_done = reinterpret_cast<int32_t *>(ws + L.final_done);
  A.wc = (pic_w + 63) / 64; A.hc = (pic_h + 63) / 64; A.n_pictures = n_pictures; A.W = pic_w; A.H = pic_h;
  const int grid = A.wc * A.hc * n_pictures;
  if (bitdepth == 8) hipLaunchKernelGGL((ctu_filter_kernel<uint8_t, false>), dim3(grid), dim3(256), sizeof(ctuf::filt_lds<uint8_t>), st, A);
  else hipLaunchKernelGGL((ctu_filter_kernel<uint16_t, false>), dim3(grid), dim3(256), sizeof(ctuf::filt_lds<uint16_t>), st, A);
  UVGHIP_CHECK_LAUNCH();
}
