/* Frame-level half of the hip backend: an all-intra frame handed to the device's closed loop where the encoder would queue its
 * per-CTU jobs.
 *
 * This file is compiled INSIDE the uvg266 source tree like strategies-hip-state.c (copy it to src/strategies/hip/; INTEGRATION.md
 * section 10).  It sees encoder_state_t and does field extraction only: the frame's source planes and the frame-level QP / lambda in,
 * the picture after the in-loop filters into frame->rec and every WPP row's substream into the row's leaf state out -- the samples,
 * decisions and bins are libuvg266hip.so's (uvghip_frame_pool_*, include/uvg266_hip.h section 7a).
 *
 * Two call sites, both applied by tools/refcheck/patch_ref_hip.py:
 *   uvg_encode_one_frame (src/encoderstate.c:2051-2091): `if (uvg_hip_frame_enabled(state)) uvg_hip_frame_begin(state); else
 *     encoder_state_encode(state);` -- behind encoder_state_init_new_frame, in front of the creation of the bitstream job;
 *   uvg_encoder_state_worker_write_bitstream (src/encoder_state-bitstream.c:1609-1612): uvg_hip_frame_finish(state) in front of
 *     uvg_encoder_state_write_bitstream -- the job waits for the device where it would have waited for the rows' jobs;
 *   (and uvg266_close, src/uvg266.c:56: uvg_hip_frame_close(encoder->control) releases the pool with the encoder instance.)
 * Everything else of the frame is the encoder's own: parameter sets, slice header, entry points, the children's streams moved into
 * the main stream, the hash SEI over frame->rec.
 *
 * UVG266_HIP_FRAME=1 asks for it.  A configuration the closed loop does not cover is an ERROR then, not a silent fall-back to
 * the CPU search (the product path fails loudly): --preset medium / slow with -p 1 and --wpp is what is covered, with or without --tiles.
 */
#include "encoderstate.h"
#include "encoder.h"
#include "videoframe.h"
#include "image.h"
#include "bitstream.h"
#include "rate_control.h"
#include "uvg266_hip.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define HIP_FRAME_LEAVES 8192         /* WPP leaf states of a frame: every tile's CTU rows (2160p in 8 x 4 tiles: 272) */
#define HIP_FRAME_SLOTS 256           /* main encoder states = frames in flight (cfg.owf + 1): the pool's slots */

static struct {
  const encoder_state_t *state;
  int begun;
} hip_slots[HIP_FRAME_SLOTS];
static uvghip_frame_pool_t *hip_pool;
static const encoder_control_t *hip_pool_ctrl;      /* the encoder instance the pool was made for */
static pthread_mutex_t hip_slots_lock = PTHREAD_MUTEX_INITIALIZER;

static void hip_frame_die(const char *what)
{
  fprintf(stderr, "hip frame backend: %s (%s)\n", what, uvghip_last_error());
  abort();
}

static int hip_slot_of(const encoder_state_t *state, int create)
{
  int at = -1;
  pthread_mutex_lock(&hip_slots_lock);
  for (int i = 0; i < HIP_FRAME_SLOTS && at < 0; ++i) if (hip_slots[i].state == state) at = i;
  for (int i = 0; i < HIP_FRAME_SLOTS && at < 0 && create; ++i) {
    if (!hip_slots[i].state) { hip_slots[i].state = state; hip_slots[i].begun = 0; at = i; }
  }
  pthread_mutex_unlock(&hip_slots_lock);
  return at;
}

/* the configuration uvghip_loop_plan_create's kernels were built for: what --preset medium / slow -p 1 --wpp sets (cfg.c's preset table) */
static const char *hip_frame_unsupported(const encoder_state_t *state)
{
  const encoder_control_t *ctrl = state->encoder_control;
  const uvg_config *c = &ctrl->cfg;
  if (c->intra_period != 1) return "intra period != 1 (P / B pictures go through uvghip_loop_pb_run_inflight)";
  if (!c->wpp) return "--no-wpp";
  if (c->slices) return "slices";
  if (ctrl->chroma_format != UVG_CSP_420) return "chroma format";
  if (c->rdo > 1) return "rd >= 2";
  if (!c->rdoq_enable || c->rdoq_skip) return "rdoq off / rdoq-skip";
  if (c->signhide_enable) return "signhide";
  if (c->trskip_enable || c->chroma_trskip_enable) return "transform skip";
  if (c->mts || c->lfnst || c->jccr || c->cclm || c->dual_tree || c->mip || c->mrl || c->isp || c->ibc || c->dep_quant) return "a VVC tool the medium preset leaves off";
  if (c->alf_type || c->lmcs_enable) return "ALF / LMCS";
  if (!c->sao_type || !c->deblock_enable || c->deblock_beta || c->deblock_tc) return "SAO off, deblocking off or with offsets";
  if (c->cu_split_termination != UVG_CU_SPLIT_TERMINATION_ZERO) return "cu-split-termination";
  if (c->full_intra_search || c->intra_rdo_et || c->fast_residual_cost_limit || c->lossless || c->implicit_rdpcm) return "an intra search option";
  if (c->max_btt_depth[0] || c->ml_pu_depth_intra) return "multi-type tree / ml-pu-depth-intra";
  if (c->target_bitrate > 0 || c->vaq || c->set_qp_in_cu || c->erp_aqp || c->scaling_list) return "rate control / adaptive QP / scaling lists";
  if (state->tile->frame->source->roi.roi_array) return "roi";
  if (ctrl->in.width % 8 || ctrl->in.height % 8) return "picture size not a multiple of 8";
  return NULL;
}

int uvg_hip_frame_enabled(const encoder_state_t *state)
{
  const char *e = getenv("UVG266_HIP_FRAME");
  if (!e || !*e || !strcmp(e, "0")) return 0;
  const char *why = hip_frame_unsupported(state);
  if (why) {
    fprintf(stderr, "hip frame backend: UVG266_HIP_FRAME is set, but the closed loop does not cover this configuration: %s\n", why);
    abort();
  }
  return 1;
}

/* the WPP rows' leaf states in the order the bitstream writer visits them (encoder_state_write_bitstream_children) */
static int hip_collect_rows(encoder_state_t *state, encoder_state_t **rows, int cap, int n)
{
  if (state->is_leaf) {
    if (state->type != ENCODER_STATE_TYPE_WAVEFRONT_ROW || n >= cap) return -1;
    rows[n] = state;
    return n + 1;
  }
  for (int i = 0; state->children[i].encoder_control; ++i) {
    n = hip_collect_rows(&state->children[i], rows, cap, n);
    if (n < 0) return n;
  }
  return n;
}

void uvg_hip_frame_begin(encoder_state_t *state)
{
  const encoder_control_t *ctrl = state->encoder_control;
  encoder_state_t *rows[HIP_FRAME_LEAVES];
  const int n = hip_collect_rows(state, rows, HIP_FRAME_LEAVES, 0);
  if (n < state->tile->frame->height_in_lcu) { fprintf(stderr, "hip frame backend: %d WPP rows for %d CTU rows\n", n, state->tile->frame->height_in_lcu); abort(); }

  /* the frame-level parameters as every CTU of the frame would see them (no rate control, no ROI: checked above).  Derived on the MAIN
   * state: a tile state's frame has no source picture before encoder_state_encode makes its sub-image (src/encoderstate.c:1232-1262) */
  encoder_state_t *leaf = state;
  vector2d_t origin = {0, 0};
  uvg_set_lcu_lambda_and_qp(leaf, origin);                      /* src/rate_control.c:1097-1188 */
  uvghip_ctu_params_t p;
  memset(&p, 0, sizeof p);
  p.pic_w = ctrl->in.width; p.pic_h = ctrl->in.height;
  p.qp = leaf->qp; p.qp_c = ctrl->qp_map[0][leaf->qp];
  p.depth_min = ctrl->cfg.pu_depth_intra.min[0]; p.depth_max = ctrl->cfg.pu_depth_intra.max[0];
  p.wpp = ctrl->cfg.wpp; p.combine_intra_cus = ctrl->cfg.combine_intra_cus;
  p.rough_levels = ctrl->cfg.intra_rough_search_levels; p.rd = ctrl->cfg.rdo;
  p.lambda = leaf->lambda; p.lambda_sqrt = leaf->lambda_sqrt; p.c_lambda = leaf->c_lambda;
  p.chroma_weight_u = leaf->chroma_weights[1]; p.chroma_weight_v = leaf->chroma_weights[2];
  p.c_lambda_tu = uvg_calculate_chroma_lambda(leaf, false, 0);

  const int n_slots = ctrl->cfg.owf + 1;                        /* the encoder's main states (src/encoder_state-ctors_dtors.c) */
  const int at = hip_slot_of(state, 1);
  if (at < 0 || at >= n_slots) { fprintf(stderr, "hip frame backend: more main encoder states than --owf + 1 = %d\n", n_slots); abort(); }
  if (hip_pool && hip_pool_ctrl != ctrl) {
    fprintf(stderr, "hip frame backend: a second encoder instance in this process (one frame pool at a time: close the first encoder)\n");
    abort();
  }
  if (!hip_pool) {                                              /* (uvg_encode_one_frame runs on the encoder's own thread only) */
    /* frames collect in groups of half the frames in flight: two launches beside each other in the steady state (UVG266_HIP_FRAME_GROUP) */
    const char *e = getenv("UVG266_HIP_FRAME_GROUP");
    const int group = e && atoi(e) > 0 ? atoi(e) : (n_slots + 1) / 2;
    if (uvghip_init(0) != 0) hip_frame_die("uvghip_init");
    /* --tiles: the grid as the encoder derived it (encoder.c:445-478), in CTUs; every tile's WPP rows are leaf states of their own */
    const int cols = ctrl->cfg.tiles_width_count, trows = ctrl->cfg.tiles_height_count;
    if (uvghip_frame_pool_create_tiles(ctrl->bitdepth, &p, (int)ctrl->cfg.sao_type, n_slots, group, ctrl->tiles_col_width, cols, ctrl->tiles_row_height, trows, &hip_pool))
      hip_frame_die("uvghip_frame_pool_create_tiles");
  }
  hip_pool_ctrl = ctrl;
  const uvg_picture *src = state->tile->frame->source;
  if (uvghip_frame_pool_begin(hip_pool, at, &p, src->y, src->u, src->v, src->stride, src->stride / 2)) hip_frame_die("uvghip_frame_pool_begin");
  hip_slots[at].begun = 1;
}

/* uvg_bitstream_writebyte (src/bitstream.c:150-169) for n bytes at once: whole chunks instead of a call per byte (a 2160p frame has
 * megabytes of them), and the stream's zerocount as uvg_bitstream_put_byte (:215-226) would have left it */
static void hip_append(bitstream_t *s, const uint8_t *bytes, int n)
{
  uint8_t zeros = n >= 3 ? 0 : s->zerocount;       /* (the bytes carry their emulation prevention: never three zeros in a row) */
  for (int i = n >= 3 ? n - 3 : 0; i < n; ++i) zeros = bytes[i] == 0 ? zeros + 1 : 0;
  while (n > 0) {
    if (s->last == NULL || s->last->len == UVG_DATA_CHUNK_SIZE) {
      uvg_data_chunk *c = uvg_bitstream_alloc_chunk();
      if (!c) { fprintf(stderr, "hip frame backend: out of memory\n"); abort(); }
      if (!s->first) s->first = c;
      if (s->last) s->last->next = c;
      s->last = c;
    }
    const int room = UVG_DATA_CHUNK_SIZE - (int)s->last->len, k = n < room ? n : room;
    memcpy(s->last->data + s->last->len, bytes, k);
    s->last->len += k; s->len += k;
    bytes += k; n -= k;
  }
  s->zerocount = zeros;
}

void uvg_hip_frame_finish(encoder_state_t *state)
{
  const int at = hip_slot_of(state, 0);
  if (at < 0 || !hip_slots[at].begun) return;                   /* a frame the CPU encoded */
  hip_slots[at].begun = 0;
  uvg_picture *rec = state->tile->frame->rec;
  const uint8_t *bytes;
  const int32_t *row_bytes;
  int n_rows;
  if (uvghip_frame_pool_finish(hip_pool, at, rec->y, rec->u, rec->v, rec->stride, rec->stride / 2, &bytes, &row_bytes, &n_rows))
    hip_frame_die("uvghip_frame_pool_finish");
  encoder_state_t *rows[HIP_FRAME_LEAVES];
  const int n = hip_collect_rows(state, rows, HIP_FRAME_LEAVES, 0);
  if (n != n_rows) { fprintf(stderr, "hip frame backend: %d rows from the device for %d leaf states\n", n_rows, n); abort(); }
  for (int r = 0; r < n; ++r) {
    bitstream_t *s = &rows[r]->stream;
    /* the row's bytes as the row's coder leaves them (uvg_bitstream_put_byte's emulation prevention already applied) */
    hip_append(s, bytes, row_bytes[r]);
    bytes += row_bytes[r];
  }
}

/* Called by uvg266_close (src/uvg266.c:56-93) behind the stop of the thread queue: the pool and its device memory go with the encoder
 * instance that used them; the next instance of the process makes its own. */
void uvg_hip_frame_close(const encoder_control_t *ctrl)
{
  pthread_mutex_lock(&hip_slots_lock);
  if (!hip_pool || hip_pool_ctrl != ctrl) { pthread_mutex_unlock(&hip_slots_lock); return; }
  uvghip_frame_pool_destroy(hip_pool);
  hip_pool = NULL;
  hip_pool_ctrl = NULL;
  memset(hip_slots, 0, sizeof hip_slots);
  pthread_mutex_unlock(&hip_slots_lock);
}
