// Deblocking filter on gfx950 (src/filter.c; plain C upstream, no strategy slot).
// Whole-picture kernel, launched twice: all vertical edges, then all horizontal
// edges.  The reference's per-CTU schedule with its 8-column delay
// (filter.c:1341-1380) produces the same picture because VVC sizes the filters
// so that edges of one direction never touch each other's samples
// (filter.c:587-644); DESIGN.md states the argument and the dev-time
// cross-check verifies it against uvg_filter_deblock_lcu.
//
// Bit-exact with
//   boundary strength, beta/tc           filter.c:696-698,734-822, tables :47-60
//   luma decisions + strong/weak/large   filter.c:127-198,406-585,827-1009
//   chroma                               filter.c:205-257,1036-1194
//   edge / grid selection                filter.c:1207-1300
//
// One thread owns one 4-sample edge segment (and the co-located 2-sample chroma
// segment).  Lanes of a wave take consecutive 4x4 units of a row, so the up to
// 16 samples per line that a segment reads are contiguous across lanes for
// horizontal edges and 4..16-byte runs for vertical edges.  Only samples a
// filter really modifies are written back: a neighbouring edge 4 samples away
// may be modifying the rest concurrently.
#include "uvghip_common.h"
#include "deblock_dev.h"


template <typename PX>
__global__ void __launch_bounds__(256)
deblock_pass_kernel(PX *__restrict__ y, int y_stride, PX *__restrict__ u, PX *__restrict__ v, int c_stride, int width,
                    int height, const uvghip_scu_t *__restrict__ scu, int scu_stride, dbk_cfg cfg, int dir_hor, int unit_row0)
{
  // blockIdx.z = 0: luma segments, one thread per 4x4 unit.  blockIdx.z = 1: the chroma segments (8-sample chroma grid =
  // every fourth unit across the edge direction) on their own threads, so that no thread carries a luma and two chroma
  // filters in one dependent chain.
  int ux = blockIdx.x * blockDim.x + threadIdx.x;         // 4x4 unit column
  int uy = unit_row0 + blockIdx.y;                        // 4x4 unit row (bands start at unit_row0)
  const bool chroma = blockIdx.z != 0;
  if (chroma) {
    if (!u) return;
    if (dir_hor) { if (uy & 3) return; }                  // chroma rows of the grid: luma unit rows 0, 4, 8, ...
    else ux *= 4;                                         // chroma columns: compacted, unit columns 0, 4, 8, ...
  }
  const int bx = ux * 4, by = uy * 4;
  if (bx >= width || by >= height) return;
  if ((!dir_hor && bx == 0) || (dir_hor && by == 0)) return;
  const int bit = dir_hor ? 2 : 1;
  const uvghip_scu_t *c = scu + uy * scu_stride + ux;
  // single tree (filter.c:1284-1289 -> :1247-1254): a chroma segment is filtered where the unit has a luma edge AND a
  // chroma TU edge.  A dual tree walks a second cu_array for chroma (:1290-1292), which uvghip_scu_t does not model:
  // the header states chroma_edges must be a subset of luma_edges.
  const int le = c->luma_edges, ce = c->chroma_edges;
  if (!(le & bit)) return;
  bool keep_p = false;
  if (cfg.snapshot) {
    if (dir_hor && (bx & 63) >= 56 && bx < width - 8) return;      // "the last 8 pixels will be deblocked when processing the next LCU"
    keep_p = ((dir_hor ? by : bx) & 63) == 0;                       // the CTU on the other side has not seen this edge yet
  }
  if (!chroma) luma_segment<PX>(y, y_stride, scu, scu_stride, bx, by, dir_hor != 0, cfg, keep_p);
  else if (ce & bit) chroma_segment<PX>(u, v, c_stride, scu, scu_stride, bx >> 1, by >> 1, dir_hor != 0, cfg, keep_p);
}

static int deblock_launch(int bitdepth, void *y, int y_stride, void *u, void *v, int c_stride, int width, int height,
                          const uvghip_scu_t *scu, int scu_stride, int beta_offset_div2, int tc_offset_div2, int slice_is_b,
                          int frame_qp, const int8_t *chroma_qp_map_host, int row0, int row1, int passes, hipStream_t st,
                          const char *who, int snapshot = 0)
{
  if (bitdepth != 8 && bitdepth != 10) return uvghip_set_error(hipErrorInvalidValue, who);
  if (width <= 0 || height <= 0 || (width & 3) || (height & 3) || row0 < 0 || row1 > height || row0 >= row1 ||
      (row0 & 3) || (row1 & 3) || !(passes & 3))
    return uvghip_set_error(hipErrorInvalidValue, who);
  dbk_cfg cfg;
  cfg.beta_offset_div2 = beta_offset_div2; cfg.tc_offset_div2 = tc_offset_div2; cfg.slice_is_b = slice_is_b; cfg.frame_qp = frame_qp;
  cfg.has_qp_map = chroma_qp_map_host != nullptr;
  cfg.snapshot = snapshot;
  for (int i = 0; i < 64; ++i) cfg.qp_map[i] = chroma_qp_map_host ? chroma_qp_map_host[i] : 0;
  const int ux = width / 4;
  constexpr int DBK_THREADS = 64;     // one wave per workgroup: 2160 workgroups at 1080p spread evenly over the 256 CUs (256-thread groups: 540)
  for (int dir_hor = 0; dir_hor < 2; ++dir_hor) {
    if (!(passes & (1 << dir_hor))) continue;
    // vertical edges: the segments of unit rows [row0/4, row1/4).  Horizontal edges: the edges at y = row0 .. row1
    // INCLUSIVE of the band's lower boundary (when it is not the picture's): both neighbours filter a shared boundary
    // edge, each into its own copy of the rows around it.
    const int u0 = row0 / 4, u1 = dir_hor && row1 < height ? row1 / 4 + 1 : row1 / 4;
    dim3 grid((ux + DBK_THREADS - 1) / DBK_THREADS, u1 - u0, u ? 2 : 1);
    if (bitdepth == 8)
      deblock_pass_kernel<uint8_t><<<grid, DBK_THREADS, 0, st>>>((uint8_t *)y, y_stride, (uint8_t *)u, (uint8_t *)v, c_stride, width, height, scu, scu_stride, cfg, dir_hor, u0);
    else
      deblock_pass_kernel<uint16_t><<<grid, DBK_THREADS, 0, st>>>((uint16_t *)y, y_stride, (uint16_t *)u, (uint16_t *)v, c_stride, width, height, scu, scu_stride, cfg, dir_hor, u0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return uvghip_set_error(e, who);
  }
  return 0;
}

extern "C" int uvghip_deblock_frame(int bitdepth, void *y, int y_stride, void *u, void *v, int c_stride, int width, int height,
                                    const uvghip_scu_t *scu, int scu_stride, int beta_offset_div2, int tc_offset_div2,
                                    int slice_is_b, int frame_qp, const int8_t *chroma_qp_map_host, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return deblock_launch(bitdepth, y, y_stride, u, v, c_stride, width, height, scu, scu_stride, beta_offset_div2, tc_offset_div2,
                        slice_is_b, frame_qp, chroma_qp_map_host, 0, height, 3, uvghip_stream(stream), __func__);
}

extern "C" int uvghip_deblock_band(int bitdepth, void *y, int y_stride, void *u, void *v, int c_stride, int width, int height,
                                   const uvghip_scu_t *scu, int scu_stride, int beta_offset_div2, int tc_offset_div2,
                                   int slice_is_b, int frame_qp, const int8_t *chroma_qp_map_host, int row0, int row1,
                                   int passes, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return deblock_launch(bitdepth, y, y_stride, u, v, c_stride, width, height, scu, scu_stride, beta_offset_div2, tc_offset_div2,
                        slice_is_b, frame_qp, chroma_qp_map_host, row0, row1, passes, uvghip_stream(stream), __func__);
}

extern "C" int uvghip_deblock_frame_sao_snapshot(int bitdepth, void *y, int y_stride, void *u, void *v, int c_stride, int width, int height,
                                                 const uvghip_scu_t *scu, int scu_stride, int beta_offset_div2, int tc_offset_div2,
                                                 int slice_is_b, int frame_qp, const int8_t *chroma_qp_map_host, void *stream)
{
  UVGHIP_REQUIRE_READY();
  return deblock_launch(bitdepth, y, y_stride, u, v, c_stride, width, height, scu, scu_stride, beta_offset_div2, tc_offset_div2,
                        slice_is_b, frame_qp, chroma_qp_map_host, 0, height, 3, uvghip_stream(stream), __func__, 1);
}
