// uvghip_frame_encoder_*: the closed loop for ONE all-intra picture handed over in HOST memory -- what uvg_encode_one_frame
// (src/encoderstate.c:2051-2091) gives a frame-level backend and what it wants back:
//   in:   the source planes of state->tile->frame->source (uvg_picture: y / u / v, stride; src/uvg266.h:566-600) and the frame-level
//         parameters the per-CTU worker would read (uvghip_ctu_params_t);
//   out:  the picture the encoder returns and hashes (frame->rec after the in-loop filters: add_checksum,
//         src/encoder_state-bitstream.c:1420-1492) and the substream of every WPP row -- the bytes the row's leaf state would hold in
//         `stream` when encoder_state_worker_encode_lcu_bitstream has coded its last CTU (src/encoderstate.c:862-976), emulation
//         prevention included, ready for uvg_bitstream_move / the slice header's entry points (:977-1007, :1494-1511).
// Host code only: device buffers for one picture, one uvghip_loop_plan (search -> deblocking -> SAO -> slice data), a stream of its own.
// begin() stages the source, enqueues everything and returns; finish() waits and copies out -- between the two the encoder goes on
// (with --owf N it begins the next frames: N encoders, N streams, N pictures' wavefronts on the device at once).
// The reference-side caller: csrc/shim/frame-hip.c, run by tests/test_gpu_dropin_frame.py.
#include "uvghip_common.h"
#include <new>
#include <cstring>
#include <cstdlib>

struct uvghip_frame_encoder {
  int device, bitdepth, sao_type, w, h, wc, hc, row_cap;
  uvghip_ctu_params_t P;
  size_t b, ysz, csz, psz;
  uint8_t *src, *rec, *out;                  // device: tight planes Y, U, V one after the other
  void *cu, *coeff, *models, *ws;
  uvghip_loop_plan_t *plan;
  hipStream_t st;
  uint8_t *host_src, *host_out;              // pinned staging of the source / the output picture
  int32_t *host_row_bytes;                   // pinned [hc]
  uint8_t *host_rows;                        // pinned: the rows' bytes one after the other (grown on demand)
  size_t host_rows_cap;
  const uint8_t *d_rows;
  const int32_t *d_row_bytes;
  bool busy;
};

namespace {

void release(uvghip_frame_encoder *e)
{
  if (e->plan) uvghip_loop_plan_destroy(e->plan);
  if (e->st) (void)hipStreamDestroy(e->st);
  void *dev[] = {e->src, e->rec, e->out, e->cu, e->coeff, e->models, e->ws};
  for (void *p : dev) if (p) (void)hipFree(p);
  void *host[] = {e->host_src, e->host_out, e->host_row_bytes, e->host_rows};
  for (void *p : host) if (p) (void)hipHostFree(p);
  delete e;
}

// rows of `bytes` bytes between a tight plane and a plane with a stride
void copy_rows(uint8_t *dst, size_t dst_pitch, const uint8_t *src, size_t src_pitch, size_t bytes, int rows)
{
  if (dst_pitch == bytes && src_pitch == bytes) { memcpy(dst, src, bytes * rows); return; }
  for (int y = 0; y < rows; ++y) memcpy(dst + y * dst_pitch, src + y * src_pitch, bytes);
}

int make_plan(uvghip_frame_encoder *e, const uvghip_ctu_params_t *p)
{
  if (e->plan) { uvghip_loop_plan_destroy(e->plan); e->plan = nullptr; }
  uvghip_loop_picture_t q;
  memset(&q, 0, sizeof q);
  q.search.src_y = e->src; q.search.src_u = e->src + e->ysz; q.search.src_v = e->src + e->ysz + e->csz;
  q.search.src_stride = e->w; q.search.src_stride_c = e->w / 2;
  q.search.rec_y = e->rec; q.search.rec_u = e->rec + e->ysz; q.search.rec_v = e->rec + e->ysz + e->csz;
  q.search.rec_stride = e->w; q.search.rec_stride_c = e->w / 2;
  q.search.cu = static_cast<uvghip_scu_t *>(e->cu); q.search.cu_stride = e->wc * 16;
  q.search.coeff = static_cast<int16_t *>(e->coeff); q.search.models = static_cast<uint32_t *>(e->models);
  q.out_y = e->out; q.out_u = e->out + e->ysz; q.out_v = e->out + e->ysz + e->csz; q.out_stride = e->w; q.out_stride_c = e->w / 2;
  if (int rc = uvghip_loop_plan_create(e->bitdepth, p, &q, 1, e->sao_type, e->ws, &e->plan)) return rc;
  int n_rows = 0;
  if (int rc = uvghip_loop_plan_slice_data(e->plan, &e->d_rows, &e->d_row_bytes, &e->row_cap, &n_rows)) return rc;
  if (n_rows != e->hc) return uvghip_set_error(hipErrorInvalidValue, "uvghip_frame_encoder: the plan's rows are not the picture's CTU rows");
  e->P = *p;
  return 0;
}

}  // namespace

extern "C" int uvghip_frame_encoder_create(int bitdepth, const uvghip_ctu_params_t *params, int sao_type, uvghip_frame_encoder_t **out)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!params || !out || params->pic_w <= 0 || params->pic_h <= 0 || (params->pic_w & 7) || (params->pic_h & 7))
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  uvghip_frame_encoder *e = new (std::nothrow) uvghip_frame_encoder();
  if (!e) return uvghip_set_error(hipErrorOutOfMemory, __func__);
  memset(static_cast<void *>(e), 0, sizeof *e);
  e->bitdepth = bitdepth; e->sao_type = sao_type; e->w = params->pic_w; e->h = params->pic_h;
  e->wc = (e->w + 63) / 64; e->hc = (e->h + 63) / 64;
  e->b = bitdepth == 8 ? 1 : 2; e->ysz = (size_t)e->w * e->h * e->b; e->csz = e->ysz / 4; e->psz = e->ysz + 2 * e->csz;
  const size_t ctus = (size_t)e->wc * e->hc, cu_bytes = (size_t)e->hc * 16 * e->wc * 16 * sizeof(uvghip_scu_t);
  hipError_t err = hipGetDevice(&e->device);
  auto dev = [&](void **p, size_t bytes, bool zero) {
    if (err == hipSuccess) err = hipMalloc(p, bytes);
    if (err == hipSuccess && zero) err = hipMemset(*p, 0, bytes);
  };
  dev(reinterpret_cast<void **>(&e->src), e->psz, false);
  dev(reinterpret_cast<void **>(&e->rec), e->psz, true);
  dev(reinterpret_cast<void **>(&e->out), e->psz, false);
  dev(&e->cu, cu_bytes, true);
  dev(&e->coeff, ctus * 6144 * sizeof(int16_t), false);
  dev(&e->models, ctus * 3 * UVGHIP_CTU_MODELS * sizeof(uint32_t), false);
  dev(&e->ws, uvghip_loop_workspace_bytes(bitdepth, 1, e->w, e->h), false);
  if (err == hipSuccess) err = hipHostMalloc(reinterpret_cast<void **>(&e->host_src), e->psz, hipHostMallocDefault);
  if (err == hipSuccess) err = hipHostMalloc(reinterpret_cast<void **>(&e->host_out), e->psz, hipHostMallocDefault);
  if (err == hipSuccess) err = hipHostMalloc(reinterpret_cast<void **>(&e->host_row_bytes), (size_t)e->hc * sizeof(int32_t), hipHostMallocDefault);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipDeviceSynchronize();            // the memsets above ran on the null stream, e->st does not wait for it
  if (err != hipSuccess) { release(e); return uvghip_set_error(err, __func__); }
  if (int rc = make_plan(e, params)) { release(e); return rc; }      // refuses what the loop plan refuses (configuration, sao_type 0, qp_c != qp)
  *out = e;
  return 0;
}

extern "C" int uvghip_frame_encoder_begin(uvghip_frame_encoder_t *e, const uvghip_ctu_params_t *params, const void *src_y, const void *src_u, const void *src_v,
                                          int src_stride, int src_stride_c)
{
  UVGHIP_REQUIRE_READY();
  if (!e || !params || !src_y || !src_u || !src_v || src_stride < e->w || src_stride_c < e->w / 2) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (e->busy) return uvghip_set_error(hipErrorNotReady, "uvghip_frame_encoder_begin: the previous picture has not been finished");
  if (params->pic_w != e->w || params->pic_h != e->h) return uvghip_set_error(hipErrorInvalidValue, "uvghip_frame_encoder_begin: the picture size is the encoder's for good");
  UVGHIP_TRY(hipSetDevice(e->device));
  if (memcmp(params, &e->P, sizeof e->P)) {         // another QP / lambda than the last picture's: the plan binds them
    if (int rc = make_plan(e, params)) return rc;
  }
  const size_t b = e->b;
  copy_rows(e->host_src, e->w * b, static_cast<const uint8_t *>(src_y), src_stride * b, e->w * b, e->h);
  copy_rows(e->host_src + e->ysz, e->w / 2 * b, static_cast<const uint8_t *>(src_u), src_stride_c * b, e->w / 2 * b, e->h / 2);
  copy_rows(e->host_src + e->ysz + e->csz, e->w / 2 * b, static_cast<const uint8_t *>(src_v), src_stride_c * b, e->w / 2 * b, e->h / 2);
  UVGHIP_TRY(hipMemcpyAsync(e->src, e->host_src, e->psz, hipMemcpyHostToDevice, e->st));
  if (int rc = uvghip_loop_plan_run(e->plan, e->st)) return rc;
  UVGHIP_TRY(hipMemcpyAsync(e->host_out, e->out, e->psz, hipMemcpyDeviceToHost, e->st));
  UVGHIP_TRY(hipMemcpyAsync(e->host_row_bytes, e->d_row_bytes, (size_t)e->hc * sizeof(int32_t), hipMemcpyDeviceToHost, e->st));
  e->busy = true;
  return 0;
}

extern "C" int uvghip_frame_encoder_finish(uvghip_frame_encoder_t *e, void *out_y, void *out_u, void *out_v, int out_stride, int out_stride_c,
                                           const uint8_t **rows, const int32_t **row_bytes, int *n_rows)
{
  UVGHIP_REQUIRE_READY();
  if (!e || !out_y || !out_u || !out_v || out_stride < e->w || out_stride_c < e->w / 2 || !rows || !row_bytes || !n_rows)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (!e->busy) return uvghip_set_error(hipErrorNotReady, "uvghip_frame_encoder_finish: no picture has been begun");
  UVGHIP_TRY(hipSetDevice(e->device));        // finish() may run on another thread than begin() (the encoder's bitstream job)
  e->busy = false;
  UVGHIP_TRY(hipStreamSynchronize(e->st));
  size_t total = 0;
  for (int r = 0; r < e->hc; ++r) {
    const int nb = e->host_row_bytes[r];
    if (nb <= 0 || nb > e->row_cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_frame_encoder_finish: a row overflowed its slot");
    total += nb;
  }
  if (total > e->host_rows_cap) {
    if (e->host_rows) { UVGHIP_TRY(hipHostFree(e->host_rows)); e->host_rows = nullptr; e->host_rows_cap = 0; }
    const size_t want = total + total / 2 + 4096;
    UVGHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->host_rows), want, hipHostMallocDefault));
    e->host_rows_cap = want;
  }
  size_t at = 0;
  for (int r = 0; r < e->hc; ++r) {
    const int nb = e->host_row_bytes[r];
    UVGHIP_TRY(hipMemcpyAsync(e->host_rows + at, e->d_rows + (size_t)r * e->row_cap, nb, hipMemcpyDeviceToHost, e->st));
    at += nb;
  }
  const size_t b = e->b;
  copy_rows(static_cast<uint8_t *>(out_y), out_stride * b, e->host_out, e->w * b, e->w * b, e->h);
  copy_rows(static_cast<uint8_t *>(out_u), out_stride_c * b, e->host_out + e->ysz, e->w / 2 * b, e->w / 2 * b, e->h / 2);
  copy_rows(static_cast<uint8_t *>(out_v), out_stride_c * b, e->host_out + e->ysz + e->csz, e->w / 2 * b, e->w / 2 * b, e->h / 2);
  UVGHIP_TRY(hipStreamSynchronize(e->st));
  *rows = e->host_rows; *row_bytes = e->host_row_bytes; *n_rows = e->hc;
  return 0;
}

extern "C" void uvghip_frame_encoder_destroy(uvghip_frame_encoder_t *e)
{
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->st) (void)hipStreamSynchronize(e->st);
  release(e);
}
