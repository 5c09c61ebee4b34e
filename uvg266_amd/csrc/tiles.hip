// uvghip_tiles_plan_*: all-intra pictures cut into TILES (--tiles CxR --wpp): the encoder's tile states (encoder_state_t of type
// ENCODER_STATE_TYPE_TILE, src/encoder_state-ctors_dtors.c; every tile with its own sub-image of the frame, its own cu_array view,
// its own CABAC start and its own WPP rows) as independent rectangles of the device's closed loop.
//
// What a tile is in the reference (all checked against runs of the encoder, tests/golden/ref_tiles_*.npz):
//   * the search, the deblocking, the SAO decision and SAO itself see `state->tile->frame`, a sub-image: a tile's left / upper edge is
//     a PICTURE edge to them (no neighbours, no references, no filtering across: pps_loop_filter_across_tiles_enabled_flag = 0,
//     src/encoder_state-bitstream.c:788);
//   * every tile starts from freshly initialised context models, its WPP rows synchronise inside the tile;
//   * the slice data is the tiles' substreams in tile raster order, each tile's rows in order; the slice header's entry points list
//     them all (encoder_state_entry_points_explore, :977-1007); everything else of the slice header is the one-tile header;
//   * the decoded picture hash covers the whole picture.
// So a tile IS a picture of its own size whose planes are views into the frame's (origin + the frame's stride), and the plan below is
// host code only: one uvghip_loop_plan per tile SIZE (a uniform grid has at most four), holding that size's tiles of all pictures as its
// "pictures"; the plans run beside each other on their own streams -- the tiles of ONE picture are that many WPP wavefronts in flight
// (1080p in 2 x 2 tiles: four wavefronts of 29 diagonals instead of one of 62).
#include "uvghip_common.h"
#include <new>
#include <vector>
#include <cstring>
#include <cstdlib>

struct uvghip_tiles_plan {
  int bitdepth, n, w, h, cols, rows, sao_type;
  std::vector<uvghip_rect_t> tiles;          // in samples, raster order of the tiles
  std::vector<int> first_ctu;                // per tile: the tile-scan address of its first CTU (ctbAddrRsToTs of its corner)
  struct size_class {
    int w, h;
    std::vector<int> ids;                    // the tiles of this size
    uvghip_loop_plan_t *plan;
    hipStream_t st;
    hipEvent_t done;
  };
  std::vector<size_class> classes;
  std::vector<int> cls_of, slot_of;          // per tile: its class and its index among the class's tiles (-1: a tile another device owns)
  std::vector<uint8_t> owned;                // per tile
  bool all_owned;
  std::vector<uvghip_loop_picture_t> pics;
  hipEvent_t fork;
  uint32_t *sums;                            // device: [n][3]
  uint8_t *host_rows;                        // pinned staging of the group's substreams (grown on demand)
  size_t host_cap;
  uint8_t *dev_rows;                         // the same bytes gathered on the device first: ONE download instead of one per substream
  size_t dev_cap;
  struct piece { const uint8_t *src; unsigned long long dst; unsigned long long len; };
  piece *host_tab, *dev_tab;                 // the gather's table (pinned / device), grown on demand
  size_t tab_cap;
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the grid from the columns' widths and the rows' heights in CTUs (encoder->tiles_col_width / tiles_row_height)
int grid_of(int pic_w, int pic_h, const std::vector<int> &colw, const std::vector<int> &rowh, std::vector<uvghip_rect_t> &tiles, std::vector<int> &first_ctu)
{
  const int wc = (pic_w + 63) / 64, hc = (pic_h + 63) / 64, cols = (int)colw.size(), rows = (int)rowh.size();
  if (cols < 1 || rows < 1 || cols > wc || rows > hc || cols >= 48 || rows >= 48) return 1;      // (MAX_TILES_PER_DIM, global.h:297 with cfg.c:314; the encoder refuses more tiles than CTUs, encoder.c:405-412)
  int sw = 0, sh = 0;
  for (int v : colw) { if (v < 1) return 1; sw += v; }
  for (int v : rowh) { if (v < 1) return 1; sh += v; }
  if (sw != wc || sh != hc) return 1;
  tiles.resize((size_t)cols * rows);
  first_ctu.resize((size_t)cols * rows);
  int at = 0, y0 = 0;
  for (int r = 0; r < rows; ++r) {
    const int th = rowh[r];
    int x0 = 0;
    for (int c = 0; c < cols; ++c) {
      const int tw = colw[c];
      uvghip_rect_t &t = tiles[(size_t)r * cols + c];
      t.x = x0 * 64; t.y = y0 * 64;
      t.w = (x0 + tw) * 64 > pic_w ? pic_w - x0 * 64 : tw * 64;
      t.h = (y0 + th) * 64 > pic_h ? pic_h - y0 * 64 : th * 64;
      first_ctu[(size_t)r * cols + c] = at;
      at += tw * th;
      x0 += tw;
    }
    y0 += th;
  }
  return 0;
}

// encoder.c:445-451: uniform spacing, column i is (i + 1) * W / n - i * W / n CTUs wide
void uniform(int n_ctus, int parts, std::vector<int> &out)
{
  out.clear();
  for (int i = 0; i < parts; ++i) out.push_back((i + 1) * n_ctus / parts - i * n_ctus / parts);
}
int grid(int pic_w, int pic_h, int cols, int rows, std::vector<uvghip_rect_t> &tiles, std::vector<int> &first_ctu)
{
  const int wc = (pic_w + 63) / 64, hc = (pic_h + 63) / 64;
  if (cols < 1 || rows < 1 || cols > wc || rows > hc) return 1;
  std::vector<int> colw, rowh;
  uniform(wc, cols, colw); uniform(hc, rows, rowh);
  return grid_of(pic_w, pic_h, colw, rowh, tiles, first_ctu);
}

struct class_key { int w, h, count; };
void classes_of(const std::vector<uvghip_rect_t> &tiles, const uint8_t *owned, std::vector<class_key> &keys, std::vector<int> &cls_of, std::vector<int> &slot_of)
{
  cls_of.assign(tiles.size(), -1); slot_of.assign(tiles.size(), -1);
  for (size_t t = 0; t < tiles.size(); ++t) {
    if (owned && !owned[t]) continue;
    size_t k = 0;
    while (k < keys.size() && (keys[k].w != tiles[t].w || keys[k].h != tiles[t].h)) ++k;
    if (k == keys.size()) keys.push_back(class_key{tiles[t].w, tiles[t].h, 0});
    cls_of[t] = (int)k; slot_of[t] = keys[k].count++;
  }
}

// a block per substream: its bytes from the size class's row slot to its place in the packed buffer
__global__ void __launch_bounds__(256) tile_gather_kernel(const uvghip_tiles_plan::piece *__restrict__ tab, uint8_t *__restrict__ packed)
{
  const uvghip_tiles_plan::piece p = tab[blockIdx.x];
  uint8_t *dst = packed + p.dst;
  if ((((size_t)p.src | (size_t)dst) & 15) == 0) {
    const uint4 *s4 = reinterpret_cast<const uint4 *>(p.src);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    const unsigned long long n4 = p.len >> 4;
    for (unsigned long long i = threadIdx.x; i < n4; i += blockDim.x) d4[i] = s4[i];
    for (unsigned long long i = (n4 << 4) + threadIdx.x; i < p.len; i += blockDim.x) dst[i] = p.src[i];
  } else {
    for (unsigned long long i = threadIdx.x; i < p.len; i += blockDim.x) dst[i] = p.src[i];
  }
}

// pieces (host vector) -> pl->host_rows[0, total): the table up, one gather launch, one download; the caller waits for the stream
int gather_to_host(uvghip_tiles_plan *pl, const std::vector<uvghip_tiles_plan::piece> &pieces, size_t total, hipStream_t st)
{
  if (pieces.empty()) return 0;
  if (total > pl->host_cap) {
    if (pl->host_rows) { UVGHIP_TRY(hipHostFree(pl->host_rows)); pl->host_rows = nullptr; pl->host_cap = 0; }
    const size_t want = total + total / 4 + 4096;
    UVGHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pl->host_rows), want, hipHostMallocDefault));
    pl->host_cap = want;
  }
  if (total > pl->dev_cap) {
    if (pl->dev_rows) { UVGHIP_TRY(hipFree(pl->dev_rows)); pl->dev_rows = nullptr; pl->dev_cap = 0; }
    const size_t want = total + total / 4 + 4096;
    UVGHIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->dev_rows), want));
    pl->dev_cap = want;
  }
  if (pieces.size() > pl->tab_cap) {
    if (pl->host_tab) { UVGHIP_TRY(hipHostFree(pl->host_tab)); pl->host_tab = nullptr; }
    if (pl->dev_tab) { UVGHIP_TRY(hipFree(pl->dev_tab)); pl->dev_tab = nullptr; }
    pl->tab_cap = 0;
    const size_t want = pieces.size() + pieces.size() / 4 + 64;
    UVGHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pl->host_tab), want * sizeof(uvghip_tiles_plan::piece), hipHostMallocDefault));
    UVGHIP_TRY(hipMalloc(reinterpret_cast<void **>(&pl->dev_tab), want * sizeof(uvghip_tiles_plan::piece)));
    pl->tab_cap = want;
  }
  memcpy(pl->host_tab, pieces.data(), pieces.size() * sizeof(uvghip_tiles_plan::piece));          // (the previous call's table is no longer in use: every call ends with a wait)
  UVGHIP_TRY(hipMemcpyAsync(pl->dev_tab, pl->host_tab, pieces.size() * sizeof(uvghip_tiles_plan::piece), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(tile_gather_kernel, dim3((unsigned)pieces.size()), dim3(256), 0, st, pl->dev_tab, pl->dev_rows);
  UVGHIP_TRY(hipGetLastError());
  UVGHIP_TRY(hipMemcpyAsync(pl->host_rows, pl->dev_rows, total, hipMemcpyDeviceToHost, st));
  return 0;
}

void destroy(uvghip_tiles_plan *pl)
{
  for (auto &c : pl->classes) {
    if (c.plan) uvghip_loop_plan_destroy(c.plan);
    if (c.st) (void)hipStreamDestroy(c.st);
    if (c.done) (void)hipEventDestroy(c.done);
  }
  if (pl->fork) (void)hipEventDestroy(pl->fork);
  if (pl->host_rows) (void)hipHostFree(pl->host_rows);
  if (pl->dev_rows) (void)hipFree(pl->dev_rows);
  if (pl->host_tab) (void)hipHostFree(pl->host_tab);
  if (pl->dev_tab) (void)hipFree(pl->dev_tab);
  delete pl;
}

}  // namespace

// A HOST function (works without a device): the uniform tile grid of --tiles <cols>x<rows> (encoder.c:445-451, 480-510): tiles[cols * rows]
// in samples, raster order; first_ctu[cols * rows] (may be NULL): the tile-scan address of every tile's first CTU.
extern "C" int uvghip_tile_grid(int pic_w, int pic_h, int cols, int rows, uvghip_rect_t *tiles, int32_t *first_ctu)
{
  if (pic_w <= 0 || pic_h <= 0 || !tiles) return uvghip_set_error(hipErrorInvalidValue, __func__);
  std::vector<uvghip_rect_t> t;
  std::vector<int> f;
  if (grid(pic_w, pic_h, cols, rows, t, f)) return uvghip_set_error(hipErrorInvalidValue, "uvghip_tile_grid: more tiles than CTUs in a dimension (or none)");
  for (size_t i = 0; i < t.size(); ++i) { tiles[i] = t[i]; if (first_ctu) first_ctu[i] = f[i]; }
  return 0;
}

// ... of --tiles-width-split / --tiles-height-split (encoder.c:452-478): the columns' widths and the rows' heights in CTUs
// (encoder->tiles_col_width[] / tiles_row_height[]; they must add up to the picture's).  HOST function.
extern "C" int uvghip_tile_grid_split(int pic_w, int pic_h, const int32_t *col_ctus, int cols, const int32_t *row_ctus, int rows, uvghip_rect_t *tiles, int32_t *first_ctu)
{
  if (pic_w <= 0 || pic_h <= 0 || !tiles || !col_ctus || !row_ctus || cols < 1 || rows < 1) return uvghip_set_error(hipErrorInvalidValue, __func__);
  std::vector<uvghip_rect_t> t;
  std::vector<int> f;
  if (grid_of(pic_w, pic_h, std::vector<int>(col_ctus, col_ctus + cols), std::vector<int>(row_ctus, row_ctus + rows), t, f))
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_tile_grid_split: the columns / rows do not add up to the picture's CTUs (or an empty one, or too many)");
  for (size_t i = 0; i < t.size(); ++i) { tiles[i] = t[i]; if (first_ctu) first_ctu[i] = f[i]; }
  return 0;
}

extern "C" size_t uvghip_tiles_workspace_bytes(int bitdepth, int n_pictures, int pic_w, int pic_h, int cols, int rows)
{
  return uvghip_tiles_workspace_bytes_owned(bitdepth, n_pictures, pic_w, pic_h, cols, rows, nullptr);
}

extern "C" size_t uvghip_tiles_workspace_bytes_owned(int bitdepth, int n_pictures, int pic_w, int pic_h, int cols, int rows, const uint8_t *owned)
{
  const int wc = (pic_w + 63) / 64, hc = (pic_h + 63) / 64;
  if (pic_w <= 0 || pic_h <= 0 || cols < 1 || rows < 1 || cols > wc || rows > hc) return 0;
  std::vector<int> colw, rowh;
  uniform(wc, cols, colw); uniform(hc, rows, rowh);
  return uvghip_tiles_workspace_bytes_split(bitdepth, n_pictures, pic_w, pic_h, colw.data(), cols, rowh.data(), rows, owned);
}

extern "C" size_t uvghip_tiles_workspace_bytes_split(int bitdepth, int n_pictures, int pic_w, int pic_h, const int32_t *col_ctus, int cols, const int32_t *row_ctus, int rows,
                                                     const uint8_t *owned)
{
  if ((bitdepth != 8 && bitdepth != 10) || n_pictures <= 0 || pic_w <= 0 || pic_h <= 0 || !col_ctus || !row_ctus || cols < 1 || rows < 1) return 0;
  std::vector<uvghip_rect_t> t;
  std::vector<int> f, cls_of, slot_of;
  if (grid_of(pic_w, pic_h, std::vector<int>(col_ctus, col_ctus + cols), std::vector<int>(row_ctus, row_ctus + rows), t, f)) return 0;
  std::vector<class_key> keys;
  classes_of(t, owned, keys, cls_of, slot_of);
  if (keys.empty()) return 0;
  size_t at = align_up((size_t)n_pictures * 3 * sizeof(uint32_t), 256);
  for (const class_key &k : keys) at += align_up(uvghip_loop_workspace_bytes(bitdepth, n_pictures * k.count, k.w, k.h), 256);
  return at;
}

// pictures: the WHOLE pictures (planes with their strides; cu with cu_stride >= 16 * CTUs per picture row); coeff / models hold the CTUs in
// TILE-SCAN order (the order of the bitstream: tile after tile, raster inside a tile).  params: the whole picture's (pic_w, pic_h).
extern "C" int uvghip_tiles_plan_create(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_loop_picture_t *pictures, int n_pictures, int tile_cols, int tile_rows,
                                        int sao_type, void *workspace, uvghip_tiles_plan_t **plan_out)
{
  return uvghip_tiles_plan_create_owned(bitdepth, params, pictures, n_pictures, tile_cols, tile_rows, nullptr, sao_type, workspace, plan_out);
}

// ... of the tiles this device OWNS (owned[cols * rows], raster order; NULL = all): the tiles of a picture over the devices of a node.  Tiles
// share nothing -- no halo in the search, none in the filters (pps_loop_filter_across_tiles_enabled_flag = 0) --, so the only exchange of an
// all-intra picture is at its end: the substreams and the three words of the checksum (uvghip_tiles_plan_substreams) go to whoever writes
// the NAL units.  The planes, tables and the workspace layout are those of the whole picture on every device; a device touches its tiles' parts.
extern "C" int uvghip_tiles_plan_create_owned(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_loop_picture_t *pictures, int n_pictures, int tile_cols,
                                              int tile_rows, const uint8_t *owned, int sao_type, void *workspace, uvghip_tiles_plan_t **plan_out)
{
  if (!params) return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int wc = (params->pic_w + 63) / 64, hc = (params->pic_h + 63) / 64;
  if (tile_cols < 1 || tile_rows < 1 || tile_cols > wc || tile_rows > hc)
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_create: more tiles than CTUs in a dimension (or none)");
  std::vector<int> colw, rowh;
  uniform(wc, tile_cols, colw); uniform(hc, tile_rows, rowh);
  return uvghip_tiles_plan_create_split(bitdepth, params, pictures, n_pictures, colw.data(), tile_cols, rowh.data(), tile_rows, owned, sao_type, workspace, plan_out);
}

// ... with the columns' widths and the rows' heights given in CTUs: --tiles-width-split / --tiles-height-split (encoder.c:452-478)
extern "C" int uvghip_tiles_plan_create_split(int bitdepth, const uvghip_ctu_params_t *params, const uvghip_loop_picture_t *pictures, int n_pictures, const int32_t *col_ctus,
                                              int tile_cols, const int32_t *row_ctus, int tile_rows, const uint8_t *owned, int sao_type, void *workspace,
                                              uvghip_tiles_plan_t **plan_out)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || !pictures || n_pictures <= 0 || !workspace || !plan_out || !col_ctus || !row_ctus || tile_cols < 1 || tile_rows < 1) return uvghip_set_error(hipErrorInvalidValue, __func__);
  uvghip_tiles_plan *pl = new (std::nothrow) uvghip_tiles_plan;
  if (!pl) return uvghip_set_error(hipErrorOutOfMemory, __func__);
  pl->bitdepth = bitdepth; pl->n = n_pictures; pl->w = params->pic_w; pl->h = params->pic_h; pl->cols = tile_cols; pl->rows = tile_rows; pl->sao_type = sao_type;
  pl->fork = nullptr; pl->host_rows = nullptr; pl->host_cap = 0; pl->dev_rows = nullptr; pl->dev_cap = 0; pl->host_tab = nullptr; pl->dev_tab = nullptr; pl->tab_cap = 0;
  if (grid_of(pl->w, pl->h, std::vector<int>(col_ctus, col_ctus + tile_cols), std::vector<int>(row_ctus, row_ctus + tile_rows), pl->tiles, pl->first_ctu)) {
    delete pl;
    return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_create: the tile columns / rows do not add up to the picture's CTUs (or an empty one, or too many)");
  }
  const int wc = (pl->w + 63) / 64;
  for (int i = 0; i < n_pictures; ++i) {
    const uvghip_ctu_picture_t &s = pictures[i].search;
    if (!s.src_y || !s.src_u || !s.src_v || !s.rec_y || !s.rec_u || !s.rec_v || !s.cu || !s.coeff || !s.models || !pictures[i].out_y || !pictures[i].out_u || !pictures[i].out_v ||
        s.src_stride < pl->w || s.rec_stride < pl->w || s.src_stride_c < pl->w / 2 || s.rec_stride_c < pl->w / 2 || s.cu_stride < wc * 16 || pictures[i].out_stride < pl->w ||
        pictures[i].out_stride_c < pl->w / 2) { delete pl; return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_create: a picture's planes, tables or strides"); }
  }
  pl->pics.assign(pictures, pictures + n_pictures);
  std::vector<class_key> keys;
  pl->owned.assign(pl->tiles.size(), 1);
  pl->all_owned = true;
  if (owned) for (size_t t = 0; t < pl->tiles.size(); ++t) { pl->owned[t] = owned[t] != 0; pl->all_owned = pl->all_owned && owned[t]; }
  classes_of(pl->tiles, pl->owned.data(), keys, pl->cls_of, pl->slot_of);
  if (keys.empty()) { delete pl; return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_create_owned: this device owns no tile"); }
  pl->classes.resize(keys.size());
  for (size_t k = 0; k < keys.size(); ++k) { pl->classes[k].w = keys[k].w; pl->classes[k].h = keys[k].h; pl->classes[k].plan = nullptr; pl->classes[k].st = nullptr; pl->classes[k].done = nullptr; }
  for (size_t t = 0; t < pl->tiles.size(); ++t) if (pl->owned[t]) pl->classes[pl->cls_of[t]].ids.push_back((int)t);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  pl->sums = reinterpret_cast<uint32_t *>(ws);
  size_t at = align_up((size_t)n_pictures * 3 * sizeof(uint32_t), 256);
  const size_t b = bitdepth == 8 ? 1 : 2;
  for (auto &c : pl->classes) {
    const int per = (int)c.ids.size();
    std::vector<uvghip_loop_picture_t> sub((size_t)n_pictures * per);
    for (int i = 0; i < n_pictures; ++i)
      for (int j = 0; j < per; ++j) {
        const uvghip_rect_t &t = pl->tiles[c.ids[j]];
        uvghip_loop_picture_t q = pictures[i];
        uvghip_ctu_picture_t &s = q.search;
        auto luma = [&](const void *p, int stride) { return (void *)((unsigned char *)p + ((size_t)t.y * stride + t.x) * b); };
        auto chroma = [&](const void *p, int stride) { return (void *)((unsigned char *)p + ((size_t)(t.y / 2) * stride + t.x / 2) * b); };
        s.src_y = luma(s.src_y, s.src_stride); s.src_u = chroma(s.src_u, s.src_stride_c); s.src_v = chroma(s.src_v, s.src_stride_c);
        s.rec_y = luma(s.rec_y, s.rec_stride); s.rec_u = chroma(s.rec_u, s.rec_stride_c); s.rec_v = chroma(s.rec_v, s.rec_stride_c);
        q.out_y = luma(q.out_y, q.out_stride); q.out_u = chroma(q.out_u, q.out_stride_c); q.out_v = chroma(q.out_v, q.out_stride_c);
        s.cu = s.cu + (size_t)(t.y / 4) * s.cu_stride + t.x / 4;
        s.coeff = s.coeff + (size_t)pl->first_ctu[c.ids[j]] * 6144;
        s.models = s.models + (size_t)pl->first_ctu[c.ids[j]] * 3 * UVGHIP_CTU_MODELS;
        sub[(size_t)i * per + j] = q;
      }
    uvghip_ctu_params_t p = *params;
    p.pic_w = c.w; p.pic_h = c.h;
    if (int rc = uvghip_loop_plan_create(bitdepth, &p, sub.data(), n_pictures * per, sao_type, ws + at, &c.plan)) { destroy(pl); return rc; }
    at += align_up(uvghip_loop_workspace_bytes(bitdepth, n_pictures * per, c.w, c.h), 256);
    hipError_t e = hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.done, hipEventDisableTiming);
    if (e != hipSuccess) { destroy(pl); return uvghip_set_error(e, "uvghip_tiles_plan_create: streams"); }
  }
  hipError_t e = hipEventCreateWithFlags(&pl->fork, hipEventDisableTiming);
  if (e != hipSuccess) { destroy(pl); return uvghip_set_error(e, "uvghip_tiles_plan_create: events"); }
  *plan_out = pl;
  return 0;
}

// Search + filters + slice data of every tile of every picture; returns at once.  The size classes run beside each other on the plan's own
// streams, forked from and joined to `stream`: work enqueued on `stream` afterwards sees all of it.
extern "C" int uvghip_tiles_plan_run(uvghip_tiles_plan_t *pl, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  // up to two size classes: each class's filters and coder BESIDE its search (uvghip_loop_plan_run_overlapped; it is the serial order by
  // itself when the class's wavefronts fill the device) -- six streams in all, within what the runtime carries side by side
  auto run_class = (pl->classes.size() <= 2 || getenv("UVGHIP_TILES_OVERLAP_ALL")) ? uvghip_loop_plan_run_overlapped : uvghip_loop_plan_run;
  if (pl->classes.size() == 1) return run_class(pl->classes[0].plan, stream);
  // classes 1.. on the plan's own streams, class 0 on the caller's: a uniform grid's four classes then take four streams, what the
  // runtime's hardware queues carry side by side by default (GPU_MAX_HW_QUEUES = 4; a fifth stream shares a queue with another and the
  // two launches run one after the other: measured, 4 x 4 tiles of one 1080p picture 219 -> 119 ms; uvg266_amd/__init__.py raises the default to 8)
  UVGHIP_TRY(hipEventRecord(pl->fork, st));
  for (size_t k = 1; k < pl->classes.size(); ++k) {
    auto &c = pl->classes[k];
    UVGHIP_TRY(hipStreamWaitEvent(c.st, pl->fork, 0));
    if (int rc = run_class(c.plan, c.st)) return rc;
    UVGHIP_TRY(hipEventRecord(c.done, c.st));
  }
  if (int rc = run_class(pl->classes[0].plan, stream)) return rc;
  for (size_t k = 1; k < pl->classes.size(); ++k) UVGHIP_TRY(hipStreamWaitEvent(st, pl->classes[k].done, 0));
  return 0;
}

extern "C" int uvghip_tiles_plan_layout(const uvghip_tiles_plan_t *pl, int *n_tiles, int *n_classes, int *n_substreams)
{
  if (!pl) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n_tiles) *n_tiles = (int)pl->tiles.size();
  if (n_classes) *n_classes = (int)pl->classes.size();
  if (n_substreams) {
    int s = 0;
    for (const uvghip_rect_t &t : pl->tiles) s += (t.h + 63) / 64;
    *n_substreams = s;
  }
  return 0;
}

// Where tile `tile` (raster order) of picture `picture` lives: the loop plan of its size and its picture index there -- for
// uvghip_loop_plan_results / _slice_data on a tile's own decisions and substreams.
extern "C" int uvghip_tiles_plan_tile(const uvghip_tiles_plan_t *pl, int picture, int tile, uvghip_loop_plan_t **plan, int *index, uvghip_rect_t *rect, int *first_ctu)
{
  if (!pl || picture < 0 || picture >= pl->n || tile < 0 || tile >= (int)pl->tiles.size()) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (!pl->owned[tile]) return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_tile: a tile another device owns");
  const auto &c = pl->classes[pl->cls_of[tile]];
  if (plan) *plan = c.plan;
  if (index) *index = picture * (int)c.ids.size() + pl->slot_of[tile];
  if (rect) *rect = pl->tiles[tile];
  if (first_ctu) *first_ctu = pl->first_ctu[tile];
  return 0;
}

// The NAL units of pictures [first, first + count) of the group after a run, as pictures first_poc, first_poc + 1, ... of the stream: per
// picture the slice NAL -- the one-tile header, the entry points of ALL substreams, the tiles' rows in tile raster order -- and the hash SEI
// of the whole output picture, one after the other into `out` (lens[i] bytes each).  Waits for the stream (the bytes are host memory):
// twice for the whole range.  Behind the encoder's parameter sets (its PPS carries the tile grid, :768-791) these bytes complete the .266.
extern "C" int uvghip_tiles_plan_nals(uvghip_tiles_plan_t *pl, int first, int count, int first_poc, uint8_t *out, size_t cap, size_t *lens, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl || first < 0 || count <= 0 || first + count > pl->n || first_poc < 0 || !lens || (!out && cap)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (!pl->all_owned) return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_nals: the plan holds a part of the tiles (uvghip_tiles_plan_substreams + uvghip_write_picture_nals)");
  hipStream_t st = uvghip_stream(stream);
  for (int i = first; i < first + count; ++i) {
    const uvghip_loop_picture_t &q = pl->pics[i];
    if (int rc = uvghip_picture_checksum(pl->bitdepth, q.out_y, q.out_stride, q.out_u, q.out_v, q.out_stride_c, pl->w, pl->h, pl->sums + 3 * (size_t)i, stream)) return rc;
  }
  std::vector<uint32_t> sums((size_t)3 * count);
  UVGHIP_TRY(hipMemcpyAsync(sums.data(), pl->sums + 3 * (size_t)first, sums.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  // every class's row lengths of the range (a class's pictures are [picture][tile of the class]: the range is contiguous)
  struct view { const uint8_t *rows; const int32_t *row_bytes; int row_cap, hc, per; std::vector<int32_t> nb; };
  std::vector<view> v(pl->classes.size());
  for (size_t k = 0; k < pl->classes.size(); ++k) {
    if (int rc = uvghip_loop_plan_slice_data(pl->classes[k].plan, &v[k].rows, &v[k].row_bytes, &v[k].row_cap, &v[k].hc)) return rc;
    v[k].per = (int)pl->classes[k].ids.size();
    v[k].nb.resize((size_t)count * v[k].per * v[k].hc);
    UVGHIP_TRY(hipMemcpyAsync(v[k].nb.data(), v[k].row_bytes + (size_t)first * v[k].per * v[k].hc, v[k].nb.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  }
  UVGHIP_TRY(hipStreamSynchronize(st));
  // a picture's substreams side by side at one pitch (the longest of the picture), pictures one after the other
  int n_sub = 0;
  for (const uvghip_rect_t &t : pl->tiles) n_sub += (t.h + 63) / 64;
  std::vector<int32_t> nb((size_t)count * n_sub);
  std::vector<size_t> base((size_t)count + 1), pitch(count);
  size_t at = 0;
  for (int i = 0; i < count; ++i) {
    int s = 0, longest = 1;
    for (size_t t = 0; t < pl->tiles.size(); ++t) {
      const view &c = v[pl->cls_of[t]];
      const int32_t *src = c.nb.data() + ((size_t)i * c.per + pl->slot_of[t]) * c.hc;
      for (int r = 0; r < c.hc; ++r, ++s) {
        if (src[r] <= 0 || src[r] > c.row_cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_nals: a row overflowed its slot (or the plan has not run)");
        nb[(size_t)i * n_sub + s] = src[r];
        if (src[r] > longest) longest = src[r];
      }
    }
    base[i] = at; pitch[i] = ((size_t)longest + 15) & ~(size_t)15;
    at += pitch[i] * n_sub;
  }
  base[count] = at;
  std::vector<uvghip_tiles_plan::piece> pieces;
  pieces.reserve((size_t)count * n_sub);
  for (int i = 0; i < count; ++i) {
    int s = 0;
    for (size_t t = 0; t < pl->tiles.size(); ++t) {
      const view &c = v[pl->cls_of[t]];
      const size_t idx = (size_t)(first + i) * c.per + pl->slot_of[t];
      for (int r = 0; r < c.hc; ++r, ++s)
        pieces.push_back(uvghip_tiles_plan::piece{c.rows + (idx * c.hc + r) * (size_t)c.row_cap, (unsigned long long)(base[i] + (size_t)s * pitch[i]),
                                                  (unsigned long long)nb[(size_t)i * n_sub + s]});
    }
  }
  if (int rc = gather_to_host(pl, pieces, at, st)) return rc;
  UVGHIP_TRY(hipStreamSynchronize(st));
  size_t used = 0;
  for (int i = 0; i < count; ++i) {
    size_t len = 0;
    if (int rc = uvghip_write_picture_nals(first_poc + i, pl->sao_type != 0, pl->host_rows + base[i], pitch[i], nb.data() + (size_t)i * n_sub, n_sub, sums.data() + 3 * (size_t)i,
                                           out ? out + used : nullptr, cap > used ? cap - used : 0, &len)) return rc;
    lens[i] = len;
    used += len;
  }
  return 0;
}

// What this device contributes to the NAL units of pictures [first, first + count) after a run, in HOST memory: lens[count][n_substreams] --
// the length of every substream of the picture in the order of the bitstream, 0 for those of tiles another device owns --, the owned
// substreams' bytes one after the other in that order (pictures one after the other; *used bytes, an error beyond cap), and
// sums[count][3]: the checksum terms of the owned tiles of the output picture (uvghip_picture_checksum_rect).  Over all devices the lengths
// and the sums ADD UP to the picture's (every substream has one owner); whoever has them all calls uvghip_write_picture_nals.  Waits for
// the stream.
extern "C" int uvghip_tiles_plan_substreams(uvghip_tiles_plan_t *pl, int first, int count, int32_t *lens, uint8_t *bytes, size_t cap, size_t *used, uint32_t *sums,
                                            void *stream)
{
  UVGHIP_REQUIRE_READY();
  if (!pl || first < 0 || count <= 0 || first + count > pl->n || !lens || !used || !sums || (!bytes && cap)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  UVGHIP_TRY(hipMemsetAsync(pl->sums + 3 * (size_t)first, 0, (size_t)count * 3 * sizeof(uint32_t), st));
  for (int i = first; i < first + count; ++i) {
    const uvghip_loop_picture_t &q = pl->pics[i];
    for (size_t t = 0; t < pl->tiles.size(); ++t) {
      if (!pl->owned[t]) continue;
      const uvghip_rect_t &r = pl->tiles[t];
      if (int rc = uvghip_picture_checksum_rect(pl->bitdepth, q.out_y, q.out_stride, q.out_u, q.out_v, q.out_stride_c, r.x, r.y, r.w, r.h, pl->sums + 3 * (size_t)i, stream)) return rc;
    }
  }
  UVGHIP_TRY(hipMemcpyAsync(sums, pl->sums + 3 * (size_t)first, (size_t)count * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  struct view { const uint8_t *rows; const int32_t *row_bytes; int row_cap, hc, per; std::vector<int32_t> nb; };
  std::vector<view> v(pl->classes.size());
  for (size_t k = 0; k < pl->classes.size(); ++k) {
    if (int rc = uvghip_loop_plan_slice_data(pl->classes[k].plan, &v[k].rows, &v[k].row_bytes, &v[k].row_cap, &v[k].hc)) return rc;
    v[k].per = (int)pl->classes[k].ids.size();
    v[k].nb.resize((size_t)count * v[k].per * v[k].hc);
    UVGHIP_TRY(hipMemcpyAsync(v[k].nb.data(), v[k].row_bytes + (size_t)first * v[k].per * v[k].hc, v[k].nb.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  }
  UVGHIP_TRY(hipStreamSynchronize(st));
  int n_sub = 0;
  for (const uvghip_rect_t &t : pl->tiles) n_sub += (t.h + 63) / 64;
  size_t at = 0;
  std::vector<uvghip_tiles_plan::piece> pieces;
  for (int i = 0; i < count; ++i) {
    int s = 0;
    for (size_t t = 0; t < pl->tiles.size(); ++t) {
      const int hc = (pl->tiles[t].h + 63) / 64;
      if (!pl->owned[t]) { for (int r = 0; r < hc; ++r, ++s) lens[(size_t)i * n_sub + s] = 0; continue; }
      const view &c = v[pl->cls_of[t]];
      const int32_t *src = c.nb.data() + ((size_t)i * c.per + pl->slot_of[t]) * c.hc;
      const size_t idx = (size_t)(first + i) * c.per + pl->slot_of[t];
      for (int r = 0; r < c.hc; ++r, ++s) {
        if (src[r] <= 0 || src[r] > c.row_cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_substreams: a row overflowed its slot (or the plan has not run)");
        lens[(size_t)i * n_sub + s] = src[r];
        pieces.push_back(uvghip_tiles_plan::piece{c.rows + (idx * c.hc + r) * (size_t)c.row_cap, (unsigned long long)at, (unsigned long long)src[r]});
        at += (size_t)src[r];
      }
    }
  }
  *used = at;
  if (at > cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_tiles_plan_substreams: the output buffer is too small (see *used)");
  if (int rc = gather_to_host(pl, pieces, at, st)) return rc;
  UVGHIP_TRY(hipStreamSynchronize(st));
  if (at) memcpy(bytes, pl->host_rows, at);
  return 0;
}

extern "C" void uvghip_tiles_plan_destroy(uvghip_tiles_plan_t *pl)
{
  if (pl) destroy(pl);
}
