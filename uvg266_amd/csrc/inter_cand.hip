// uvghip_merge_cand_batch / uvghip_amvp_cand_batch: the merge and AMVP candidate lists of n inter CUs (include/uvg266_hip.h, part 7),
// one lane per call over tables in global memory.  The derivation itself is inter_cand_dev.h.
#include "uvghip_common.h"
#include "inter_cand_dev.h"

namespace {

struct global_tab {                 // the lcu_t's side information of one call: 17 * 17 + 1 entries of 8 ints, modified in place
  int32_t *p;
  __device__ icand::unit &at(int i) { return *reinterpret_cast<icand::unit *>(p + (size_t)i * 8); }
};
struct global_col {
  const int32_t *p;
  __device__ icand::col_unit at(int i) const
  {
    const int32_t *o = p + (size_t)i * 8;
    icand::col_unit c;
    c.type = o[0]; c.mv[0][0] = o[1]; c.mv[0][1] = o[2]; c.mv[1][0] = o[3]; c.mv[1][1] = o[4]; c.dir = o[5]; c.poc[0] = o[6]; c.poc[1] = o[7];
    return c;
  }
};

// the 64-int call record (the layout tools/refcheck/ctu_dump.c writes and the oracle reads): [1..4] x, y, width, height of the CU; [5] POC;
// [6] slice type (0 = B); [7..8] picture size; [9] tmvp; [10] max merge candidates; [11] log2 parallel merge level; [12] wpp; [13]
// references in use, [14..29] their POCs; [30..31] list sizes, [32..39] / [40..47] L0 / L1; [49] the CU's split tree; [50] list and
// [51..52] reference indices (AMVP)
__device__ void ctx_of(const int32_t *c, icand::frame_ctx &f)
{
  f.x = c[1]; f.y = c[2]; f.w = c[3]; f.h = c[4]; f.poc = c[5]; f.is_b = c[6] == 0; f.pic_w = c[7]; f.pic_h = c[8];
  f.tmvp = c[9]; f.max_cands = c[10]; f.mer_level = c[11]; f.wpp = c[12]; f.n_refs = c[13];
  for (int i = 0; i < 16; ++i) f.ref_pocs[i] = c[14 + i];
  f.l_size[0] = c[30]; f.l_size[1] = c[31];
  for (int i = 0; i < 8; ++i) { f.l[0][i] = c[32 + i]; f.l[1][i] = c[40 + i]; }
  f.split_tree = (uint32_t)c[49];
}

// a lane's working set (the call's context, the list being built) lives in the workgroup's LDS, one slot per lane: indexed arrays in
// registers would go to scratch
struct lane_slot { icand::frame_ctx f; icand::merge_cand mc[6]; icand::amvp_ws ws; int32_t ref_idx[2], out[4]; };

__global__ void merge_cand_kernel(const int32_t *ctx, int32_t *lcu, const int32_t *col, long col_stride, const int32_t *hmvp, int n, int32_t *cands,
                                  int32_t *counts)
{
  __shared__ lane_slot slots[64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  icand::frame_ctx &f = slots[threadIdx.x].f;
  icand::merge_cand *mc = slots[threadIdx.x].mc;
  ctx_of(ctx + (size_t)i * 64, f);
  global_tab tab{lcu + (size_t)i * (icand::TCW * icand::TCW + 1) * 8};
  global_col c{col + (size_t)i * col_stride};
  const int k = icand::merge_candidates(f, tab, c, hmvp + (size_t)i * 41, mc);
  int32_t *o = cands + (size_t)i * 42;
  for (int j = 0; j < 6; ++j) {
    o[7 * j] = mc[j].dir; o[7 * j + 1] = mc[j].ref[0]; o[7 * j + 2] = mc[j].ref[1];
    o[7 * j + 3] = mc[j].mv[0][0]; o[7 * j + 4] = mc[j].mv[0][1]; o[7 * j + 5] = mc[j].mv[1][0]; o[7 * j + 6] = mc[j].mv[1][1];
  }
  counts[i] = k;
}

__global__ void amvp_cand_kernel(const int32_t *ctx, int32_t *lcu, const int32_t *col, long col_stride, const int32_t *hmvp, int n, int32_t *mv_cand)
{
  __shared__ lane_slot slots[64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t *c = ctx + (size_t)i * 64;
  icand::frame_ctx &f = slots[threadIdx.x].f;
  ctx_of(c, f);
  global_tab tab{lcu + (size_t)i * (icand::TCW * icand::TCW + 1) * 8};
  global_col cl{col + (size_t)i * col_stride};
  int32_t *ref_idx = slots[threadIdx.x].ref_idx, *out = slots[threadIdx.x].out;
  ref_idx[0] = c[51]; ref_idx[1] = c[52];
  icand::amvp_candidates(f, tab, cl, hmvp + (size_t)i * 41, c[50], ref_idx, out, &slots[threadIdx.x].ws);
  for (int j = 0; j < 4; ++j) mv_cand[(size_t)i * 4 + j] = out[j];
}

}  // namespace

extern "C" int uvghip_merge_cand_batch(const int32_t *ctx, int32_t *lcu, const int32_t *col, long col_stride, const int32_t *hmvp, int n, int32_t *cands,
                                       int32_t *counts, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!ctx || !lcu || !col || !hmvp || !cands || !counts || n < 0 || col_stride < 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n == 0) return 0;
  hipLaunchKernelGGL(merge_cand_kernel, dim3((n + 63) / 64), dim3(64), 0, uvghip_stream(stream), ctx, lcu, col, col_stride, hmvp, n, cands, counts);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_amvp_cand_batch(const int32_t *ctx, int32_t *lcu, const int32_t *col, long col_stride, const int32_t *hmvp, int n, int32_t *mv_cand,
                                      void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!ctx || !lcu || !col || !hmvp || !mv_cand || n < 0 || col_stride < 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n == 0) return 0;
  hipLaunchKernelGGL(amvp_cand_kernel, dim3((n + 63) / 64), dim3(64), 0, uvghip_stream(stream), ctx, lcu, col, col_stride, hmvp, n, mv_cand);
  UVGHIP_CHECK_LAUNCH();
}
