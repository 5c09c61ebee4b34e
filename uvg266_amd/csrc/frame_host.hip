// uvghip_frame_pool_*: the closed loop for all-intra pictures handed over in HOST memory one by one -- what uvg_encode_one_frame
// (src/encoderstate.c:2051-2091) gives a frame-level backend and what it wants back:
//   in:   the source planes of state->tile->frame->source (uvg_picture: y / u / v, stride; src/uvg266.h:566-600) and the frame-level
//         parameters the per-CTU worker would read (uvghip_ctu_params_t);
//   out:  the picture the encoder returns and hashes (frame->rec after the in-loop filters: add_checksum,
//         src/encoder_state-bitstream.c:1420-1492) and the substream of every WPP row -- the bytes the row's leaf state would hold in
//         `stream` when encoder_state_worker_encode_lcu_bitstream has coded its last CTU (src/encoderstate.c:862-976), emulation
//         prevention included, ready for uvg_bitstream_move / the slice header's entry points (:977-1007, :1494-1511).
// Host code only.  A SLOT is the device memory of one picture in flight (the encoder has cfg.owf + 1 main states); frames that are begun
// collect in a GROUP, and a group is one uvghip_loop_plan launch (search -> deblocking -> SAO -> slice data of all its pictures) on a
// stream of its own.  Why groups and not a launch per frame: measured inside the reference's CLI (profiles/r06_frame_dropin_time.txt),
// five single-picture launches beside each other run 2.5 x as fast as one after the other, seventeen run 15 x SLOWER -- every launch puts
// all its workgroups on the device at once, most of them waiting for their wavefront, and beyond the hardware queues the waiting ones of
// one launch keep the running ones of another off the device.  Inside one launch the release order interleaves the pictures' wavefronts
// (ctu_search.hip), which is what bench.py's 224-picture launches rely on.
// The reference-side caller: csrc/shim/frame-hip.c, run by tests/test_gpu_dropin_frame.py.
#include "uvghip_common.h"
#include <chrono>
#include <mutex>
#include <thread>
#include <new>
#include <vector>
#include <cstring>
#include <cstdlib>

namespace {

enum { FREE = 0, PENDING = 1, LAUNCHED = 2 };
constexpr int MAX_GROUPS = 8;

struct slot_t {
  uint8_t *src = nullptr, *rec = nullptr, *out = nullptr;      // device: tight planes Y, U, V one after the other
  void *cu = nullptr, *coeff = nullptr, *models = nullptr;
  uint8_t *host_src = nullptr, *host_out = nullptr;            // pinned staging of the source / the output picture
  uint8_t *host_rows = nullptr;                                // pinned: the rows' bytes one after the other (grown on demand)
  int32_t *host_row_bytes = nullptr;                           // pinned [hc]: this picture's row lengths, handed to the caller
  size_t host_rows_cap = 0;
  int state = FREE, group = -1, index = -1;
};

struct group_t {
  uvghip_loop_plan_t *plan = nullptr;
  uvghip_tiles_plan_t *tplan = nullptr;                        // a pool for tiled frames: the group's launch is a tiles plan
  std::vector<int32_t> lens;                                   // tiled: [k][n_sub] substream lengths, the bytes of the group's pictures one
  std::vector<uint8_t> bytes;                                  //        after the other, where each picture's begin
  std::vector<size_t> pic_off;
  void *ws = nullptr;
  hipStream_t st = nullptr;
  int32_t *host_row_bytes = nullptr;                           // pinned [group_max][hc]
  std::vector<int> slots, planned;                             // the slots of this launch in plan order; those the plan was made for
  uvghip_ctu_params_t P, planned_P;
  const uint8_t *d_rows = nullptr;
  const int32_t *d_row_bytes = nullptr;
  int row_cap = 0, unfinished = 0;
  bool launched = false, waited = false;
};

// rows of `bytes` bytes between a tight plane and a plane with a stride
void copy_rows(uint8_t *dst, size_t dst_pitch, const uint8_t *src, size_t src_pitch, size_t bytes, int rows)
{
  if (dst_pitch == bytes && src_pitch == bytes) { memcpy(dst, src, bytes * rows); return; }
  for (int y = 0; y < rows; ++y) memcpy(dst + y * dst_pitch, src + y * src_pitch, bytes);
}

}  // namespace

struct uvghip_frame_pool {
  std::mutex m;                                                // the slots' and groups' states
  std::mutex fin;                                              // finish() calls one at a time (a group's download is shared by its frames); begin() goes on beside them
  int device = 0, bitdepth = 0, sao_type = 0, w = 0, h = 0, wc = 0, hc = 0, group_max = 1, open = -1;
  size_t b = 1, ysz = 0, csz = 0, psz = 0;
  std::vector<slot_t> slots;
  group_t groups[MAX_GROUPS];
  std::vector<int32_t> col_ctus, row_ctus;                     // tiled frames: the grid in CTUs (empty: one loop plan per group)
  int n_sub = 0;                                               // substreams of a picture = leaf states of the encoder (hc without tiles)
  bool tiled() const { return !col_ctus.empty(); }
};

namespace {

void release(uvghip_frame_pool *p)
{
  for (group_t &g : p->groups) {
    if (g.st) (void)hipStreamSynchronize(g.st);
    if (g.plan) uvghip_loop_plan_destroy(g.plan);
    if (g.tplan) uvghip_tiles_plan_destroy(g.tplan);
    if (g.st) (void)hipStreamDestroy(g.st);
    if (g.ws) (void)hipFree(g.ws);
    if (g.host_row_bytes) (void)hipHostFree(g.host_row_bytes);
  }
  for (slot_t &s : p->slots) {
    void *dev[] = {s.src, s.rec, s.out, s.cu, s.coeff, s.models};
    for (void *q : dev) if (q) (void)hipFree(q);
    void *host[] = {s.host_src, s.host_out, s.host_rows, s.host_row_bytes};
    for (void *q : host) if (q) (void)hipHostFree(q);
  }
  delete p;
}

// slot -> the loop plan's picture descriptor
void describe(const uvghip_frame_pool *p, const slot_t &s, uvghip_loop_picture_t &q)
{
  memset(&q, 0, sizeof q);
  q.search.src_y = s.src; q.search.src_u = s.src + p->ysz; q.search.src_v = s.src + p->ysz + p->csz;
  q.search.src_stride = p->w; q.search.src_stride_c = p->w / 2;
  q.search.rec_y = s.rec; q.search.rec_u = s.rec + p->ysz; q.search.rec_v = s.rec + p->ysz + p->csz;
  q.search.rec_stride = p->w; q.search.rec_stride_c = p->w / 2;
  q.search.cu = static_cast<uvghip_scu_t *>(s.cu); q.search.cu_stride = p->wc * 16;
  q.search.coeff = static_cast<int16_t *>(s.coeff); q.search.models = static_cast<uint32_t *>(s.models);
  q.out_y = s.out; q.out_u = s.out + p->ysz; q.out_v = s.out + p->ysz + p->csz; q.out_stride = p->w; q.out_stride_c = p->w / 2;
}

// a group's stream, workspace (for group_max pictures) and row-length staging: made when the group object is first used, kept for good
int ready_group(uvghip_frame_pool *p, group_t &g)
{
  if (g.st) return 0;
  const size_t ws = p->tiled() ? uvghip_tiles_workspace_bytes_split(p->bitdepth, p->group_max, p->w, p->h, p->col_ctus.data(), (int)p->col_ctus.size(), p->row_ctus.data(),
                                                                    (int)p->row_ctus.size(), nullptr)
                               : uvghip_loop_workspace_bytes(p->bitdepth, p->group_max, p->w, p->h);
  if (!ws) return uvghip_set_error(hipErrorInvalidValue, "uvghip_frame_pool: no workspace size for this picture / grid");
  UVGHIP_TRY(hipMalloc(&g.ws, ws));
  UVGHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.host_row_bytes), (size_t)p->group_max * p->hc * sizeof(int32_t), hipHostMallocDefault));
  UVGHIP_TRY(hipStreamCreateWithFlags(&g.st, hipStreamNonBlocking));
  return 0;
}

// the open group becomes one launch: plan (kept when the same slots with the same parameters come again), run, the downloads behind it
int launch(uvghip_frame_pool *p, int gi)
{
  group_t &g = p->groups[gi];
  const int k = (int)g.slots.size();
  if ((!g.plan && !g.tplan) || g.planned != g.slots || memcmp(&g.planned_P, &g.P, sizeof g.P)) {
    if (g.plan) { uvghip_loop_plan_destroy(g.plan); g.plan = nullptr; }
    if (g.tplan) { uvghip_tiles_plan_destroy(g.tplan); g.tplan = nullptr; }
    std::vector<uvghip_loop_picture_t> pics(k);
    for (int i = 0; i < k; ++i) describe(p, p->slots[g.slots[i]], pics[i]);
    if (p->tiled()) {
      if (int rc = uvghip_tiles_plan_create_split(p->bitdepth, &g.P, pics.data(), k, p->col_ctus.data(), (int)p->col_ctus.size(), p->row_ctus.data(), (int)p->row_ctus.size(),
                                                  nullptr, p->sao_type, g.ws, &g.tplan)) return rc;
      int n_tiles = 0, n_classes = 0, n_sub = 0;
      if (int rc = uvghip_tiles_plan_layout(g.tplan, &n_tiles, &n_classes, &n_sub)) return rc;
      if (n_sub != p->n_sub) return uvghip_set_error(hipErrorInvalidValue, "uvghip_frame_pool: the tiles plan's substreams are not the grid's rows");
    } else {
      if (int rc = uvghip_loop_plan_create(p->bitdepth, &g.P, pics.data(), k, p->sao_type, g.ws, &g.plan)) return rc;
      int n_rows = 0;
      if (int rc = uvghip_loop_plan_slice_data(g.plan, &g.d_rows, &g.d_row_bytes, &g.row_cap, &n_rows)) return rc;
      if (n_rows != p->hc) return uvghip_set_error(hipErrorInvalidValue, "uvghip_frame_pool: the plan's rows are not the picture's CTU rows");
    }
    g.planned = g.slots; g.planned_P = g.P;
  }
  if (int rc = p->tiled() ? uvghip_tiles_plan_run(g.tplan, g.st) : uvghip_loop_plan_run(g.plan, g.st)) return rc;
  for (int i = 0; i < k; ++i) {
    slot_t &s = p->slots[g.slots[i]];
    UVGHIP_TRY(hipMemcpyAsync(s.host_out, s.out, p->psz, hipMemcpyDeviceToHost, g.st));
    s.state = LAUNCHED;
  }
  if (!p->tiled()) UVGHIP_TRY(hipMemcpyAsync(g.host_row_bytes, g.d_row_bytes, (size_t)k * p->hc * sizeof(int32_t), hipMemcpyDeviceToHost, g.st));
  g.launched = true; g.waited = false; g.unfinished = k;
  if (p->open == gi) p->open = -1;
  return 0;
}

}  // namespace

extern "C" int uvghip_frame_pool_create(int bitdepth, const uvghip_ctu_params_t *params, int sao_type, int n_slots, int group_max, uvghip_frame_pool_t **out)
{
  return uvghip_frame_pool_create_tiles(bitdepth, params, sao_type, n_slots, group_max, nullptr, 0, nullptr, 0, out);
}

extern "C" int uvghip_frame_pool_create_tiles(int bitdepth, const uvghip_ctu_params_t *params, int sao_type, int n_slots, int group_max, const int32_t *col_ctus, int cols,
                                              const int32_t *row_ctus, int rows, uvghip_frame_pool_t **out)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || !out || params->pic_w <= 0 || params->pic_h <= 0 || (params->pic_w & 7) || (params->pic_h & 7) || n_slots < 1 || n_slots > 256 || group_max < 1)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const bool tiled = cols * rows > 1;
  if (tiled && (!col_ctus || !row_ctus || cols < 1 || rows < 1)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  uvghip_frame_pool *p = new (std::nothrow) uvghip_frame_pool();
  if (!p) return uvghip_set_error(hipErrorOutOfMemory, __func__);
  if (tiled) { p->col_ctus.assign(col_ctus, col_ctus + cols); p->row_ctus.assign(row_ctus, row_ctus + rows); }
  p->bitdepth = bitdepth; p->sao_type = sao_type; p->w = params->pic_w; p->h = params->pic_h;
  p->wc = (p->w + 63) / 64; p->hc = (p->h + 63) / 64; p->group_max = group_max < n_slots ? group_max : n_slots;
  p->b = bitdepth == 8 ? 1 : 2; p->ysz = (size_t)p->w * p->h * p->b; p->csz = p->ysz / 4; p->psz = p->ysz + 2 * p->csz;
  p->n_sub = p->hc;
  if (tiled) {                                                  // every tile's CTU rows (the grid itself is checked by the tiles plan)
    p->n_sub = 0;
    for (int r = 0; r < rows; ++r) p->n_sub += row_ctus[r] * cols;
  }
  const size_t ctus = (size_t)p->wc * p->hc, cu_bytes = (size_t)p->hc * 16 * p->wc * 16 * sizeof(uvghip_scu_t);
  hipError_t err = hipGetDevice(&p->device);
  p->slots.resize(n_slots);
  auto dev = [&](void **q, size_t bytes, bool zero) {
    if (err == hipSuccess) err = hipMalloc(q, bytes);
    if (err == hipSuccess && zero) err = hipMemset(*q, 0, bytes);
  };
  auto host = [&](void **q, size_t bytes) { if (err == hipSuccess) err = hipHostMalloc(q, bytes, hipHostMallocDefault); };
  for (slot_t &s : p->slots) {
    dev(reinterpret_cast<void **>(&s.src), p->psz, false);
    dev(reinterpret_cast<void **>(&s.rec), p->psz, true);
    dev(reinterpret_cast<void **>(&s.out), p->psz, false);
    dev(&s.cu, cu_bytes, true);
    dev(&s.coeff, ctus * 6144 * sizeof(int16_t), false);
    dev(&s.models, ctus * 3 * UVGHIP_CTU_MODELS * sizeof(uint32_t), false);
    host(reinterpret_cast<void **>(&s.host_src), p->psz);
    host(reinterpret_cast<void **>(&s.host_out), p->psz);
    host(reinterpret_cast<void **>(&s.host_row_bytes), (size_t)p->n_sub * sizeof(int32_t));
    s.host_rows_cap = p->psz / 4 + 4096;                          // grown in finish() by the rare picture that needs more
    host(reinterpret_cast<void **>(&s.host_rows), s.host_rows_cap);
  }
  if (err == hipSuccess) err = hipDeviceSynchronize();            // the memsets above ran on the null stream, the groups' streams do not wait for it
  if (err != hipSuccess) { release(p); return uvghip_set_error(err, __func__); }
  // a plan for one picture now: a configuration the loop plan refuses (sao_type 0, qp_c != qp, anything but the medium / slow settings) is
  // refused here and not at the first frame's launch
  int rc = ready_group(p, p->groups[0]);
  if (!rc) {
    uvghip_loop_picture_t q;
    describe(p, p->slots[0], q);
    if (tiled) {
      uvghip_tiles_plan_t *probe = nullptr;
      rc = uvghip_tiles_plan_create_split(bitdepth, params, &q, 1, col_ctus, cols, row_ctus, rows, nullptr, sao_type, p->groups[0].ws, &probe);
      if (!rc) uvghip_tiles_plan_destroy(probe);
    } else {
      uvghip_loop_plan_t *probe = nullptr;
      rc = uvghip_loop_plan_create(bitdepth, params, &q, 1, sao_type, p->groups[0].ws, &probe);
      if (!rc) uvghip_loop_plan_destroy(probe);
    }
  }
  if (rc) { release(p); return rc; }
  *out = p;
  return 0;
}

extern "C" int uvghip_frame_pool_begin(uvghip_frame_pool_t *p, int slot, const uvghip_ctu_params_t *params, const void *src_y, const void *src_u, const void *src_v,
                                       int src_stride, int src_stride_c)
{
  UVGHIP_REQUIRE_READY();
  if (!p || !params || !src_y || !src_u || !src_v || slot < 0 || slot >= (int)p->slots.size() || src_stride < p->w || src_stride_c < p->w / 2)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (params->pic_w != p->w || params->pic_h != p->h) return uvghip_set_error(hipErrorInvalidValue, "uvghip_frame_pool_begin: the picture size is the pool's for good");
  UVGHIP_TRY(hipSetDevice(p->device));
  slot_t &s = p->slots[slot];
  {
    std::lock_guard<std::mutex> lock(p->m);
    if (s.state != FREE) return uvghip_set_error(hipErrorNotReady, "uvghip_frame_pool_begin: the slot's previous picture has not been finished");
  }
  // staged outside the lock: the slot is the caller's until it is PENDING
  const size_t b = p->b;
  copy_rows(s.host_src, p->w * b, static_cast<const uint8_t *>(src_y), src_stride * b, p->w * b, p->h);
  copy_rows(s.host_src + p->ysz, p->w / 2 * b, static_cast<const uint8_t *>(src_u), src_stride_c * b, p->w / 2 * b, p->h / 2);
  copy_rows(s.host_src + p->ysz + p->csz, p->w / 2 * b, static_cast<const uint8_t *>(src_v), src_stride_c * b, p->w / 2 * b, p->h / 2);
  std::lock_guard<std::mutex> lock(p->m);
  if (p->open >= 0 && memcmp(&p->groups[p->open].P, params, sizeof *params)) {       // another QP / lambda: a plan binds one set of parameters
    if (int rc = launch(p, p->open)) return rc;
  }
  if (p->open < 0) {
    for (int i = 0; i < MAX_GROUPS && p->open < 0; ++i) if (!p->groups[i].launched && p->groups[i].slots.empty()) p->open = i;
    if (p->open < 0) return uvghip_set_error(hipErrorNotReady, "uvghip_frame_pool_begin: every group is in flight (finish frames in the order they were begun)");
    if (int rc = ready_group(p, p->groups[p->open])) { p->open = -1; return rc; }
    p->groups[p->open].P = *params;
  }
  group_t &g = p->groups[p->open];
  UVGHIP_TRY(hipMemcpyAsync(s.src, s.host_src, p->psz, hipMemcpyHostToDevice, g.st));
  s.state = PENDING; s.group = p->open; s.index = (int)g.slots.size();
  g.slots.push_back(slot);
  if ((int)g.slots.size() >= p->group_max) return launch(p, p->open);
  return 0;
}

extern "C" int uvghip_frame_pool_finish(uvghip_frame_pool_t *p, int slot, void *out_y, void *out_u, void *out_v, int out_stride, int out_stride_c,
                                        const uint8_t **rows, const int32_t **row_bytes, int *n_rows)
{
  UVGHIP_REQUIRE_READY();
  if (!p || slot < 0 || slot >= (int)p->slots.size() || !out_y || !out_u || !out_v || out_stride < p->w || out_stride_c < p->w / 2 || !rows || !row_bytes || !n_rows)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  UVGHIP_TRY(hipSetDevice(p->device));        // finish() may run on another thread than begin() (the encoder's bitstream job)
  slot_t &s = p->slots[slot];
  std::lock_guard<std::mutex> serial(p->fin);
  std::unique_lock<std::mutex> lock(p->m);
  if (s.state == FREE) return uvghip_set_error(hipErrorNotReady, "uvghip_frame_pool_finish: no picture has been begun in this slot");
  if (s.state == PENDING) {
    // the frame is asked for while its group still collects.  The encoder's bitstream job for the FIRST frame of a clip becomes runnable the
    // moment the frame is begun (nothing to wait for), with the encoder still reading the next frames: give the group a moment to grow --
    // as long as frames keep arriving (a new one within 6 ms, 60 ms at most) -- instead of launching one picture alone (0.47 s for 1080p)
    for (int quiet = 0, waited_ms = 0; s.state == PENDING && quiet < 3 && waited_ms < 60; waited_ms += 2) {
      const size_t before = p->groups[s.group].slots.size();
      lock.unlock();
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      lock.lock();
      quiet = s.state == PENDING && p->groups[s.group].slots.size() == before ? quiet + 1 : 0;
    }
    if (s.state == PENDING) {                 // (begin() launches a group that fills up meanwhile)
      if (int rc = launch(p, s.group)) return rc;
    }
  }
  group_t &g = p->groups[s.group];
  const bool wait = !g.waited;
  hipStream_t st = g.st;
  lock.unlock();                              // begin() of the next frames goes on while this one waits
  const bool tiled = p->tiled();
  const int n_sub = p->n_sub;
  if (wait && !tiled) UVGHIP_TRY(hipStreamSynchronize(st));
  if (wait && tiled) {
    // every substream of every picture of the group in one call (it waits for the stream): lengths, bytes in the order of the bitstream
    const int k = (int)g.slots.size();
    g.lens.resize((size_t)k * n_sub);
    if (g.bytes.size() < (size_t)k * p->psz) g.bytes.resize((size_t)k * p->psz);
    std::vector<uint32_t> sums((size_t)3 * k);
    size_t used = 0;
    if (int rc = uvghip_tiles_plan_substreams(g.tplan, 0, k, g.lens.data(), g.bytes.data(), g.bytes.size(), &used, sums.data(), st)) return rc;
    g.pic_off.assign((size_t)k + 1, 0);
    for (int i = 0; i < k; ++i) {
      size_t n = 0;
      for (int r = 0; r < n_sub; ++r) n += (size_t)g.lens[(size_t)i * n_sub + r];
      g.pic_off[i + 1] = g.pic_off[i] + n;
    }
  }
  lock.lock();
  g.waited = true;
  size_t total = 0;
  for (int r = 0; r < n_sub; ++r) {
    const int nb = tiled ? g.lens[(size_t)s.index * n_sub + r] : g.host_row_bytes[(size_t)s.index * p->hc + r];
    if (nb <= 0 || (!tiled && nb > g.row_cap)) return uvghip_set_error(hipErrorInvalidValue, "uvghip_frame_pool_finish: a row overflowed its slot");
    s.host_row_bytes[r] = nb;
    total += nb;
  }
  if (total > s.host_rows_cap) {
    if (s.host_rows) { UVGHIP_TRY(hipHostFree(s.host_rows)); s.host_rows = nullptr; s.host_rows_cap = 0; }
    const size_t want = total + total / 2 + 4096;
    UVGHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.host_rows), want, hipHostMallocDefault));
    s.host_rows_cap = want;
  }
  size_t at = 0;
  for (int r = 0; r < p->hc && !tiled; ++r) {
    const int nb = s.host_row_bytes[r];
    UVGHIP_TRY(hipMemcpyAsync(s.host_rows + at, g.d_rows + ((size_t)s.index * p->hc + r) * g.row_cap, nb, hipMemcpyDeviceToHost, st));
    at += nb;
  }
  if (tiled) memcpy(s.host_rows, g.bytes.data() + g.pic_off[s.index], total);
  lock.unlock();
  const size_t b = p->b;
  copy_rows(static_cast<uint8_t *>(out_y), out_stride * b, s.host_out, p->w * b, p->w * b, p->h);
  copy_rows(static_cast<uint8_t *>(out_u), out_stride_c * b, s.host_out + p->ysz, p->w / 2 * b, p->w / 2 * b, p->h / 2);
  copy_rows(static_cast<uint8_t *>(out_v), out_stride_c * b, s.host_out + p->ysz + p->csz, p->w / 2 * b, p->w / 2 * b, p->h / 2);
  if (!tiled) UVGHIP_TRY(hipStreamSynchronize(st));       // the rows (the group's stream carries nothing else before all its frames are finished)
  lock.lock();
  s.state = FREE; s.group = -1; s.index = -1;
  if (--g.unfinished == 0) { g.launched = false; g.slots.clear(); }
  *rows = s.host_rows; *row_bytes = s.host_row_bytes; *n_rows = n_sub;
  return 0;
}

extern "C" void uvghip_frame_pool_destroy(uvghip_frame_pool_t *p)
{
  if (!p) return;
  (void)hipSetDevice(p->device);
  release(p);
}
