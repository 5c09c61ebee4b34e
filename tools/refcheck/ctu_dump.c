/*
 * Dev-time tool: run the REAL reference encoder (libuvg266.a built from the unmodified reference, see README.md) on a yuv file
 * and record, per CTU, what crosses the boundary of the closed-loop CTU search:
 *   - the CABAC models the search starts from (state->cabac.ctx when uvg_search_lcu is entered, src/search.c:2386),
 *   - lambda / qp of the CTU,
 *   - after uvg_search_lcu: the models the search ends with, the CTU's cu_info_t entries (frame->cu_array), its
 *     reconstruction before any in-loop filter (frame->rec, written by copy_lcu_to_cu_data, search.c:2331) and its
 *     coefficients (lcu_coeff_t),
 *   - after the CTU's uvg_encode_coding_tree (src/encoderstate.c:888): the models the real coder ends with.
 * The two functions are intercepted with the linker (-Wl,--wrap=...); nothing in the reference is modified.  The .266
 * bitstream the encoder produced is written next to the records.
 *
 *   ctu_dump <in.yuv> <W> <H> <frames> <out.bin> <out.266> [option value]...
 *
 * Not part of the product, the tests or the bench; tools/refcheck/ctu_dump.sh builds and runs it and
 * tools/refcheck/ctu_to_npz.py turns the records into tests/golden/ref_ctu_*.npz.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "uvg266.h"
#include "encoderstate.h"
#include "cabac.h"
#include "cu.h"
#include "videoframe.h"
#include "search.h"
#include "encode_coding_tree.h"

static FILE *g_out;

enum { NCTX = 257 };
typedef struct { uint16_t state0[NCTX], state1[NCTX]; uint8_t rate[NCTX]; } models_t;

static void put(models_t *s, int at, const cabac_ctx_t *src, int n)
{
  for (int i = 0; i < n; ++i) { s->state0[at + i] = src[i].state[0]; s->state1[at + i] = src[i].state[1]; s->rate[at + i] = src[i].rate; }
}
/* model index space of oracle/orc_search.c: the 244 residual models of orc_coeff_cost.c, then the CU-level ones */
static void snapshot(const cabac_data_t *cb, models_t *s)
{
  memset(s, 0, sizeof *s);
  put(s, 0, &cb->ctx.sig_coeff_group_model[0], 4);
  put(s, 4, cb->ctx.cu_sig_model_luma[0], 12); put(s, 16, cb->ctx.cu_sig_model_chroma[0], 8);
  put(s, 28, cb->ctx.cu_parity_flag_model_luma, 21); put(s, 49, cb->ctx.cu_parity_flag_model_chroma, 11);
  put(s, 70, cb->ctx.cu_gtx_flag_model_luma[1], 21); put(s, 91, cb->ctx.cu_gtx_flag_model_chroma[1], 11);
  put(s, 112, cb->ctx.cu_gtx_flag_model_luma[0], 21); put(s, 133, cb->ctx.cu_gtx_flag_model_chroma[0], 11);
  put(s, 154, cb->ctx.cu_ctx_last_x_luma, 20); put(s, 174, cb->ctx.cu_ctx_last_x_chroma, 3);
  put(s, 194, cb->ctx.cu_ctx_last_y_luma, 20); put(s, 214, cb->ctx.cu_ctx_last_y_chroma, 3);
  put(s, 234, cb->ctx.qt_cbf_model_luma, 4); put(s, 238, cb->ctx.qt_cbf_model_cb, 2); put(s, 240, cb->ctx.qt_cbf_model_cr, 3);
  put(s, 243, &cb->ctx.cu_qt_root_cbf_model, 1);
  put(s, 244, cb->ctx.split_flag_model, 9);
  put(s, 253, &cb->ctx.intra_luma_mpm_flag_model, 1);
  put(s, 254, cb->ctx.luma_planar_model, 2);
  put(s, 256, &cb->ctx.chroma_pred_model, 1);
}

/* the models of the inter syntax (P / B slices), index space of oracle/orc_search.c behind the 257 above:
 * skip flag [3], pred mode [2], merge flag, merge idx, inter dir [6], ref idx [2], mvd [2], mvp idx = 18 */
enum { NCTX_INTER = 18 };
typedef struct { uint16_t state0[NCTX_INTER], state1[NCTX_INTER]; uint8_t rate[NCTX_INTER]; } models_inter_t;
static void put2(models_inter_t *s, int at, const cabac_ctx_t *src, int n)
{
  for (int i = 0; i < n; ++i) { s->state0[at + i] = src[i].state[0]; s->state1[at + i] = src[i].state[1]; s->rate[at + i] = src[i].rate; }
}
static void snapshot_inter(const cabac_data_t *cb, models_inter_t *s)
{
  memset(s, 0, sizeof *s);
  put2(s, 0, cb->ctx.cu_skip_flag_model, 3);
  put2(s, 3, cb->ctx.cu_pred_mode_model, 2);
  put2(s, 5, &cb->ctx.cu_merge_flag_ext_model, 1);
  put2(s, 6, &cb->ctx.cu_merge_idx_ext_model, 1);
  put2(s, 7, cb->ctx.inter_dir, 6);
  put2(s, 13, cb->ctx.cu_ref_pic_model, 2);
  put2(s, 15, cb->ctx.cu_mvd_model, 2);
  put2(s, 17, &cb->ctx.mvp_idx_model, 1);
}

static void rec_begin(const char *name, int narr)
{
  uint32_t magic = 0x52454631, nl = (uint32_t)strlen(name), na = (uint32_t)narr;
  fwrite(&magic, 4, 1, g_out); fwrite(&nl, 4, 1, g_out); fwrite(name, 1, nl, g_out); fwrite(&na, 4, 1, g_out);
}
static void rec_arr(int code, const void *p, size_t n)
{
  static const int sz[] = {1, 2, 2, 4, 4, 8, 8};  /* u8 u16 i16 i32 u32 i64 f64 */
  uint32_t c = (uint32_t)code, nn = (uint32_t)n;
  fwrite(&c, 4, 1, g_out); fwrite(&nn, 4, 1, g_out); fwrite(p, (size_t)sz[code], n, g_out);
}
enum { A_U8 = 0, A_U16 = 1, A_I16 = 2, A_I32 = 3, A_U32 = 4, A_I64 = 5, A_F64 = 6 };
#define A_PX (UVG_BIT_DEPTH == 8 ? A_U8 : A_U16)

void __real_uvg_search_lcu(encoder_state_t *const state, const int x, const int y, const yuv_t *const hor_buf, const yuv_t *const ver_buf,
                           lcu_coeff_t *coeff);
void __wrap_uvg_search_lcu(encoder_state_t *const state, const int x, const int y, const yuv_t *const hor_buf, const yuv_t *const ver_buf,
                           lcu_coeff_t *coeff)
{
  models_t before, after;
  models_inter_t before2, after2;
  snapshot(&state->cabac, &before);
  snapshot_inter(&state->cabac, &before2);
  __real_uvg_search_lcu(state, x, y, hor_buf, ver_buf, coeff);
  snapshot(&state->search_cabac, &after);
  snapshot_inter(&state->search_cabac, &after2);
  const videoframe_t *frame = state->tile->frame;
  int32_t meta[8] = {(int32_t)state->frame->num, x, y, state->qp, frame->width, frame->height, state->frame->slicetype, state->frame->QP};
  double lam[6] = {state->lambda, state->lambda_sqrt, state->c_lambda, state->chroma_weights[1], state->chroma_weights[2], state->chroma_weights[3]};
  /* compact cu_info per 4x4 */
  uint8_t cu[256][12];
  uint32_t trees[256][2];
  memset(cu, 0, sizeof cu); memset(trees, 0, sizeof trees);
  static uvg_pixel ry[64 * 64], ru[32 * 32], rv[32 * 32];
  memset(ry, 0, sizeof ry); memset(ru, 0, sizeof ru); memset(rv, 0, sizeof rv);
  for (int yy = 0; yy < 64; yy += 4)
    for (int xx = 0; xx < 64; xx += 4) {
      if (x + xx >= frame->width || y + yy >= frame->height) continue;
      const cu_info_t *c = uvg_cu_array_at_const(frame->cu_array, x + xx, y + yy);
      uint8_t *o = cu[(yy >> 2) * 16 + (xx >> 2)];
      o[0] = c->type; o[1] = c->log2_width; o[2] = c->log2_height; o[3] = c->log2_chroma_width; o[4] = c->log2_chroma_height;
      o[5] = (uint8_t)c->cbf; o[6] = (uint8_t)c->intra.mode; o[7] = (uint8_t)c->intra.mode_chroma; o[8] = c->luma_deblocking;
      o[9] = c->chroma_deblocking; o[10] = c->qp; o[11] = (uint8_t)(c->tr_skip | (c->tr_idx << 3) | (c->joint_cb_cr << 6));
      trees[(yy >> 2) * 16 + (xx >> 2)][0] = c->split_tree;
      trees[(yy >> 2) * 16 + (xx >> 2)][1] = c->mode_type_tree;
    }
  for (int yy = 0; yy < 64 && y + yy < frame->height; ++yy)
    for (int xx = 0; xx < 64 && x + xx < frame->width; ++xx) ry[yy * 64 + xx] = frame->rec->y[(y + yy) * frame->rec->stride + x + xx];
  for (int yy = 0; yy < 32 && y / 2 + yy < frame->height / 2; ++yy)
    for (int xx = 0; xx < 32 && x / 2 + xx < frame->width / 2; ++xx) {
      ru[yy * 32 + xx] = frame->rec->u[(y / 2 + yy) * (frame->rec->stride / 2) + x / 2 + xx];
      rv[yy * 32 + xx] = frame->rec->v[(y / 2 + yy) * (frame->rec->stride / 2) + x / 2 + xx];
    }
  /* inter side information per 4x4 and the picture's reference lists (P / B pictures; zeros in an intra picture) */
  static int32_t inter[256][8];
  memset(inter, 0, sizeof inter);
  for (int yy = 0; yy < 64; yy += 4)
    for (int xx = 0; xx < 64; xx += 4) {
      if (x + xx >= frame->width || y + yy >= frame->height) continue;
      const cu_info_t *c = uvg_cu_array_at_const(frame->cu_array, x + xx, y + yy);
      if (c->type != CU_INTER) continue;
      int32_t *o = inter[(yy >> 2) * 16 + (xx >> 2)];
      o[0] = c->inter.mv[0][0]; o[1] = c->inter.mv[0][1]; o[2] = c->inter.mv[1][0]; o[3] = c->inter.mv[1][1];
      o[4] = c->inter.mv_ref[0]; o[5] = c->inter.mv_ref[1]; o[6] = c->inter.mv_dir;
      o[7] = c->skipped | c->merged << 1 | c->merge_idx << 2 | c->inter.imv << 5 | c->inter.mv_cand0 << 8 | c->inter.mv_cand1 << 11 | c->root_cbf << 14;
    }
  int32_t refs[1 + 16 + 2 + 32 + 1];
  memset(refs, 0, sizeof refs);
  refs[0] = (int32_t)state->frame->ref->used_size;
  for (unsigned i = 0; i < state->frame->ref->used_size && i < 16; ++i) refs[1 + i] = state->frame->ref->pocs[i];
  refs[17] = state->frame->ref_LX_size[0]; refs[18] = state->frame->ref_LX_size[1];
  for (int l = 0; l < 2; ++l) for (int i = 0; i < 16; ++i) refs[19 + 16 * l + i] = state->frame->ref_LX[l][i];
  refs[51] = state->frame->poc;
  rec_begin("search", 15);
  rec_arr(A_I32, meta, 8); rec_arr(A_F64, lam, 6);
  rec_arr(A_U8, &before, sizeof before); rec_arr(A_U8, &after, sizeof after);
  rec_arr(A_U8, cu, sizeof cu); rec_arr(A_U32, trees, 512);
  rec_arr(A_PX, ry, 64 * 64); rec_arr(A_PX, ru, 32 * 32); rec_arr(A_PX, rv, 32 * 32);
  rec_arr(A_I16, coeff->y, 64 * 64);
  {
    static int16_t uv[2 * 32 * 32];
    memcpy(uv, coeff->u, 2 * 32 * 32); memcpy(uv + 32 * 32, coeff->v, 2 * 32 * 32);
    rec_arr(A_I16, uv, 2 * 32 * 32);
  }
  rec_arr(A_I32, inter, 256 * 8);
  rec_arr(A_I32, refs, 52);
  rec_arr(A_U8, &before2, sizeof before2); rec_arr(A_U8, &after2, sizeof after2);
}

/* payload bytes the arithmetic coder hands to the bitstream (uvg_bitstream_put_byte, called from uvg_cabac_write): counted
 * here rather than with uvg_bitstream_tell, which also sees the emulation-prevention bytes the bitstream layer inserts */
static int64_t g_cabac_bytes;
static uint8_t g_tree_bytes[1 << 16];      /* the bytes of the coding tree being recorded */
static int g_tree_n = -1;
void __real_uvg_bitstream_put_byte(bitstream_t *const stream, const uint32_t data);
void __wrap_uvg_bitstream_put_byte(bitstream_t *const stream, const uint32_t data)
{
  ++g_cabac_bytes;
  if (g_tree_n >= 0 && g_tree_n < (int)sizeof g_tree_bytes) g_tree_bytes[g_tree_n++] = (uint8_t)data;
  __real_uvg_bitstream_put_byte(stream, data);
}

/* the end of a WPP row's substream (src/encoderstate.c:921-938): end_of_sub_stream_one_bit, uvg_cabac_finish, a one bit and
 * the alignment -- after that the row's stream holds the whole substream (emulation prevention included). */
static int g_row_end_pending, g_last_coded_y;
void __real_uvg_cabac_finish(cabac_data_t *const data);
void __wrap_uvg_cabac_finish(cabac_data_t *const data)
{
  g_row_end_pending = 1;
  __real_uvg_cabac_finish(data);
}
void __real_uvg_bitstream_align_zero(bitstream_t *const stream);
void __wrap_uvg_bitstream_align_zero(bitstream_t *const stream)
{
  __real_uvg_bitstream_align_zero(stream);
  if (!g_row_end_pending) return;
  g_row_end_pending = 0;
  static uint8_t buf[1 << 22];
  size_t n = 0;
  for (const uvg_data_chunk *c = stream->first; c; c = c->next) { if (n + c->len > sizeof buf) break; memcpy(buf + n, c->data, c->len); n += c->len; }
  int32_t meta[2] = {g_last_coded_y / 64, (int32_t)stream->len};
  rec_begin("row", 2);
  rec_arr(A_I32, meta, 2);
  rec_arr(A_U8, buf, n);
}

void __real_uvg_encode_coding_tree(encoder_state_t *const state, lcu_coeff_t *coeff, enum uvg_tree_type tree_type, const cu_loc_t *const cu_loc,
                                   const cu_loc_t *const chroma_loc, split_tree_t split_tree, bool has_chroma);
void __wrap_uvg_encode_coding_tree(encoder_state_t *const state, lcu_coeff_t *coeff, enum uvg_tree_type tree_type, const cu_loc_t *const cu_loc,
                                   const cu_loc_t *const chroma_loc, split_tree_t split_tree, bool has_chroma)
{
  models_t before, after;
  snapshot(&state->cabac, &before);
  /* bits the arithmetic coder has consumed so far (every renormalisation shift and every bypass bin takes one from bits_left,
   * uvg_cabac_write gives eight back per byte it moves on) and its range: the hand-over test counts the same bins */
  int64_t coder[4];
  coder[0] = 8 * g_cabac_bytes + 8 * (int64_t)state->cabac.num_buffered_bytes + 23 - state->cabac.bits_left;
  coder[1] = state->cabac.range;
  int64_t full[10];        /* the arithmetic coder's whole state before / after: low, range, bits_left, num_buffered_bytes, buffered_byte */
  full[0] = state->cabac.low; full[1] = state->cabac.range; full[2] = state->cabac.bits_left; full[3] = state->cabac.num_buffered_bytes; full[4] = state->cabac.buffered_byte;
  g_tree_n = 0;
  g_last_coded_y = cu_loc->y;
  __real_uvg_encode_coding_tree(state, coeff, tree_type, cu_loc, chroma_loc, split_tree, has_chroma);
  const int n_bytes = g_tree_n;
  g_tree_n = -1;
  full[5] = state->cabac.low; full[6] = state->cabac.range; full[7] = state->cabac.bits_left; full[8] = state->cabac.num_buffered_bytes; full[9] = state->cabac.buffered_byte;
  coder[2] = 8 * g_cabac_bytes + 8 * (int64_t)state->cabac.num_buffered_bytes + 23 - state->cabac.bits_left;
  coder[3] = state->cabac.range;
  snapshot(&state->cabac, &after);
  models_inter_t after2;
  snapshot_inter(&state->cabac, &after2);
  int32_t meta[3] = {(int32_t)state->frame->num, cu_loc->x, cu_loc->y};
  uint16_t m_sao[6];
  m_sao[0] = state->cabac.ctx.sao_merge_flag_model.state[0]; m_sao[1] = state->cabac.ctx.sao_merge_flag_model.state[1]; m_sao[2] = state->cabac.ctx.sao_merge_flag_model.rate;
  m_sao[3] = state->cabac.ctx.sao_type_idx_model.state[0]; m_sao[4] = state->cabac.ctx.sao_type_idx_model.state[1]; m_sao[5] = state->cabac.ctx.sao_type_idx_model.rate;
  rec_begin("coded", 8);
  rec_arr(A_I32, meta, 3); rec_arr(A_U8, &before, sizeof before); rec_arr(A_U8, &after, sizeof after);
  rec_arr(A_U16, m_sao, 6);        /* the two SAO models after this CTU's SAO syntax (encode_sao precedes the coding tree) */
  rec_arr(A_I64, coder, 4);
  rec_arr(A_I64, full, 10);
  rec_arr(A_U8, g_tree_bytes, (size_t)n_bytes);
  rec_arr(A_U8, &after2, sizeof after2);
}

/* uvg_search_cu_inter (src/search_inter.c:2329): every call with what it decided -- the costs it returned and the CU's entry of the
 * lcu_t afterwards (CTU_DUMP_CU_INTER=1).  The sequence of calls is the trace of search_cu's walk over a P / B picture. */
#include "search_inter.h"
static int g_cu_inter = 0;
void __real_uvg_search_cu_inter(encoder_state_t *const state, const cu_loc_t *const cu_loc, lcu_t *lcu, double *inter_cost, double *inter_bitcost);
void __wrap_uvg_search_cu_inter(encoder_state_t *const state, const cu_loc_t *const cu_loc, lcu_t *lcu, double *inter_cost, double *inter_bitcost)
{
  __real_uvg_search_cu_inter(state, cu_loc, lcu, inter_cost, inter_bitcost);
  if (!g_cu_inter) return;
  const cu_info_t *c = LCU_GET_CU_AT_PX(lcu, SUB_SCU(cu_loc->x), SUB_SCU(cu_loc->y));
  int32_t v[20] = {(int32_t)state->frame->num, cu_loc->x, cu_loc->y, cu_loc->width, cu_loc->height, c->type, c->skipped, c->merged, c->merge_idx,
                   c->inter.mv_dir, c->inter.mv[0][0], c->inter.mv[0][1], c->inter.mv[1][0], c->inter.mv[1][1], c->inter.mv_ref[0], c->inter.mv_ref[1],
                   c->inter.mv_cand0, c->inter.mv_cand1, (int32_t)c->cbf, c->root_cbf};
  double d[2] = {*inter_cost, *inter_bitcost};
  rec_begin("cuinter", 2);
  rec_arr(A_I32, v, 20); rec_arr(A_F64, d, 2);
}

/* uvg_sao_search_lcu (src/sao.c:670, called at src/encoderstate.c:849 right after the CTU's own uvg_filter_deblock_lcu): the CTU's
 * block of frame->rec as the decision sees it (deblocked by the CTU's own edges only), the two SAO context models the bit
 * estimates read (state->search_cabac) with that structure's flags, and the decision. */
#include "sao.h"
static void sao_models(const cabac_data_t *cb, uint16_t *o)
{
  o[0] = cb->ctx.sao_merge_flag_model.state[0]; o[1] = cb->ctx.sao_merge_flag_model.state[1]; o[2] = cb->ctx.sao_merge_flag_model.rate;
  o[3] = cb->ctx.sao_type_idx_model.state[0]; o[4] = cb->ctx.sao_type_idx_model.state[1]; o[5] = cb->ctx.sao_type_idx_model.rate;
}
static void sao_pack(const sao_info_t *s, int32_t *o)
{
  o[0] = s->type; o[1] = s->eo_class; o[2] = s->ddistortion; o[3] = s->merge_left_flag; o[4] = s->merge_up_flag;
  o[5] = s->band_position[0]; o[6] = s->band_position[1];
  for (int i = 0; i < 10; ++i) o[7 + i] = s->offsets[i];
}
void __real_uvg_sao_search_lcu(const encoder_state_t *const state, int lcu_x, int lcu_y);
void __wrap_uvg_sao_search_lcu(const encoder_state_t *const state, int lcu_x, int lcu_y)
{
  const videoframe_t *frame = state->tile->frame;
  const int x = lcu_x * 64, y = lcu_y * 64;
  static uvg_pixel ry[64 * 64], ru[32 * 32], rv[32 * 32];
  memset(ry, 0, sizeof ry); memset(ru, 0, sizeof ru); memset(rv, 0, sizeof rv);
  for (int yy = 0; yy < 64 && y + yy < frame->height; ++yy)
    for (int xx = 0; xx < 64 && x + xx < frame->width; ++xx) ry[yy * 64 + xx] = frame->rec->y[(y + yy) * frame->rec->stride + x + xx];
  for (int yy = 0; yy < 32 && y / 2 + yy < frame->height / 2; ++yy)
    for (int xx = 0; xx < 32 && x / 2 + xx < frame->width / 2; ++xx) {
      ru[yy * 32 + xx] = frame->rec->u[(y / 2 + yy) * (frame->rec->stride / 2) + x / 2 + xx];
      rv[yy * 32 + xx] = frame->rec->v[(y / 2 + yy) * (frame->rec->stride / 2) + x / 2 + xx];
    }
  uint16_t m_search[6], m_coder[6], m_after[6];
  sao_models(&state->search_cabac, m_search); sao_models(&state->cabac, m_coder);
  int32_t meta[6] = {(int32_t)state->frame->num, lcu_x, lcu_y, state->search_cabac.update, state->search_cabac.only_count, state->encoder_control->cfg.sao_type};
  double lam[1] = {state->lambda};
  __real_uvg_sao_search_lcu(state, lcu_x, lcu_y);
  sao_models(&state->search_cabac, m_after);
  int32_t luma[17], chroma[17];
  sao_pack(&frame->sao_luma[lcu_y * frame->width_in_lcu + lcu_x], luma);
  sao_pack(&frame->sao_chroma[lcu_y * frame->width_in_lcu + lcu_x], chroma);
  rec_begin("sao", 10);
  rec_arr(A_I32, meta, 6); rec_arr(A_F64, lam, 1);
  rec_arr(A_U16, m_search, 6); rec_arr(A_U16, m_coder, 6); rec_arr(A_U16, m_after, 6);
  rec_arr(A_I32, luma, 17); rec_arr(A_I32, chroma, 17);
  rec_arr(A_PX, ry, 64 * 64); rec_arr(A_PX, ru, 32 * 32); rec_arr(A_PX, rv, 32 * 32);
}

/* uvg_inter_get_merge_cand (src/inter.c:1989-2192) as the inter search calls it: every g_merge_every-th call is recorded with all it
 * reads -- the lcu_t's side information as it stands at that moment (work-tree state: 17 x 17 + 1 entries), the picture's reference
 * lists, the collocated picture's motion at the 8x8 grid (with the POCs its vectors point to), the row's HMVP table -- and what it
 * returned.  (inter_clear_cu_unused modifies neighbours it looks at: the table is taken BEFORE the call.) */
#include "inter.h"
static int g_merge_every = 0, g_merge_calls = 0;
static void pack_cu(const cu_info_t *c, int32_t *o)
{
  o[0] = c->type; o[1] = c->inter.mv[0][0]; o[2] = c->inter.mv[0][1]; o[3] = c->inter.mv[1][0]; o[4] = c->inter.mv[1][1];
  o[5] = c->inter.mv_ref[0]; o[6] = c->inter.mv_ref[1]; o[7] = c->inter.mv_dir;
}
static void cand_context(const encoder_state_t *const state, const cu_loc_t *const cu_loc, const lcu_t *lcu, int32_t ctx[64], int32_t **col_out, size_t *col_n,
                         int32_t hm[1 + MAX_NUM_HMVP_CANDS * 8])
{
  const videoframe_t *frame = state->tile->frame;
  memset(ctx, 0, 64 * sizeof(int32_t));
  ctx[0] = (int32_t)state->frame->num; ctx[1] = cu_loc->x; ctx[2] = cu_loc->y; ctx[3] = cu_loc->width; ctx[4] = cu_loc->height;
  ctx[5] = state->frame->poc; ctx[6] = state->frame->slicetype; ctx[7] = frame->width; ctx[8] = frame->height;
  ctx[9] = state->encoder_control->cfg.tmvp_enable; ctx[10] = state->encoder_control->cfg.max_merge;
  ctx[11] = state->encoder_control->cfg.log2_parallel_merge_level; ctx[12] = state->encoder_control->cfg.wpp;
  ctx[13] = (int32_t)state->frame->ref->used_size;
  for (unsigned i = 0; i < state->frame->ref->used_size && i < 16; ++i) ctx[14 + i] = state->frame->ref->pocs[i];
  ctx[30] = state->frame->ref_LX_size[0]; ctx[31] = state->frame->ref_LX_size[1];
  for (int l = 0; l < 2; ++l) for (int i = 0; i < 8; ++i) ctx[32 + 8 * l + i] = state->frame->ref_LX[l][i];
  const cu_info_t *cur = LCU_GET_CU_AT_PX(lcu, SUB_SCU(cu_loc->x), SUB_SCU(cu_loc->y));
  ctx[49] = (int32_t)cur->split_tree;
  const int gw = (frame->width + 7) / 8, gh = (frame->height + 7) / 8;
  int32_t *col = calloc((size_t)gw * gh * 8, sizeof(int32_t));
  if (state->frame->ref->used_size && state->frame->ref_LX_size[0] > 0) {
    const int cr = state->frame->ref_LX[0][0];
    const cu_array_t *ca = state->frame->ref->cu_arrays[cr];
    for (int gy = 0; gy < gh; ++gy)
      for (int gx = 0; gx < gw; ++gx) {
        const cu_info_t *c = &ca->data[(gx * 8) / SCU_WIDTH + ((gy * 8) / SCU_WIDTH) * (ca->width / SCU_WIDTH)];
        int32_t *o = col + ((size_t)gy * gw + gx) * 8;
        o[0] = c->type; o[1] = c->inter.mv[0][0]; o[2] = c->inter.mv[0][1]; o[3] = c->inter.mv[1][0]; o[4] = c->inter.mv[1][1]; o[5] = c->inter.mv_dir;
        for (int l = 0; l < 2; ++l)
          o[6 + l] = (c->type == CU_INTER && (c->inter.mv_dir & (1 << l)))
                         ? state->frame->ref->images[cr]->ref_pocs[state->frame->ref->ref_LXs[cr][l][c->inter.mv_ref[l]]] : -1;
      }
  }
  *col_out = col; *col_n = (size_t)gw * gh * 8;
  const uint32_t row = (uint32_t)cu_loc->y >> LOG2_LCU_WIDTH;
  memset(hm, 0, (1 + MAX_NUM_HMVP_CANDS * 8) * sizeof(int32_t));
  hm[0] = frame->hmvp_size[row];
  for (int i = 0; i < MAX_NUM_HMVP_CANDS; ++i) pack_cu(&frame->hmvp_lut[row * MAX_NUM_HMVP_CANDS + i], hm + 1 + 8 * i);
}

/* uvg_inter_get_mv_cand (src/inter.c:1711-1737; the two AMVP predictors of a reference list): recorded like the merge calls */
static int g_amvp_calls = 0;
void __real_uvg_inter_get_mv_cand(const encoder_state_t *const state, mv_t mv_cand[2][2], const cu_info_t *const cur_cu, lcu_t *lcu, int8_t reflist,
                                  const cu_loc_t *const cu_loc);
void __wrap_uvg_inter_get_mv_cand(const encoder_state_t *const state, mv_t mv_cand[2][2], const cu_info_t *const cur_cu, lcu_t *lcu, int8_t reflist,
                                  const cu_loc_t *const cu_loc)
{
  const int take = g_merge_every > 0 && cur_cu->type != CU_IBC && (g_amvp_calls++ % g_merge_every) == 0;
  static int32_t tab[LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1][8];
  int32_t ctx[64], hm[1 + MAX_NUM_HMVP_CANDS * 8], *col = NULL;
  size_t col_n = 0;
  if (take) {
    for (int i = 0; i < LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1; ++i) pack_cu(&lcu->cu[i], tab[i]);
    cand_context(state, cu_loc, lcu, ctx, &col, &col_n, hm);
    ctx[50] = reflist; ctx[51] = cur_cu->inter.mv_ref[0]; ctx[52] = cur_cu->inter.mv_ref[1];
  }
  __real_uvg_inter_get_mv_cand(state, mv_cand, cur_cu, lcu, reflist, cu_loc);
  if (!take) return;
  int32_t out[4] = {mv_cand[0][0], mv_cand[0][1], mv_cand[1][0], mv_cand[1][1]};
  rec_begin("amvp", 5);
  rec_arr(A_I32, ctx, 64);
  rec_arr(A_I32, tab, (LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1) * 8);
  rec_arr(A_I32, col, col_n);
  rec_arr(A_I32, hm, 1 + MAX_NUM_HMVP_CANDS * 8);
  rec_arr(A_I32, out, 4);
  free(col);
}

uint8_t __real_uvg_inter_get_merge_cand(const encoder_state_t *const state, const cu_loc_t *const cu_loc, inter_merge_cand_t mv_cand[MRG_MAX_NUM_CANDS], lcu_t *lcu);
uint8_t __wrap_uvg_inter_get_merge_cand(const encoder_state_t *const state, const cu_loc_t *const cu_loc, inter_merge_cand_t mv_cand[MRG_MAX_NUM_CANDS], lcu_t *lcu)
{
  const int take = g_merge_every > 0 && (g_merge_calls++ % g_merge_every) == 0;
  static int32_t tab[LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1][8];
  if (take) for (int i = 0; i < LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1; ++i) pack_cu(&lcu->cu[i], tab[i]);
  const uint8_t n = __real_uvg_inter_get_merge_cand(state, cu_loc, mv_cand, lcu);
  if (!take) return n;
  const videoframe_t *frame = state->tile->frame;
  int32_t ctx[64];
  memset(ctx, 0, sizeof ctx);
  ctx[0] = (int32_t)state->frame->num; ctx[1] = cu_loc->x; ctx[2] = cu_loc->y; ctx[3] = cu_loc->width; ctx[4] = cu_loc->height;
  ctx[5] = state->frame->poc; ctx[6] = state->frame->slicetype; ctx[7] = frame->width; ctx[8] = frame->height;
  ctx[9] = state->encoder_control->cfg.tmvp_enable; ctx[10] = state->encoder_control->cfg.max_merge;
  ctx[11] = state->encoder_control->cfg.log2_parallel_merge_level; ctx[12] = state->encoder_control->cfg.wpp;
  ctx[13] = (int32_t)state->frame->ref->used_size;
  for (unsigned i = 0; i < state->frame->ref->used_size && i < 16; ++i) ctx[14 + i] = state->frame->ref->pocs[i];
  ctx[30] = state->frame->ref_LX_size[0]; ctx[31] = state->frame->ref_LX_size[1];
  for (int l = 0; l < 2; ++l) for (int i = 0; i < 8; ++i) ctx[32 + 8 * l + i] = state->frame->ref_LX[l][i];
  ctx[48] = n;
  /* the collocated picture (L0[0]) at the 8x8 grid: type, vectors, direction, the POCs the vectors point to */
  const int gw = (frame->width + 7) / 8, gh = (frame->height + 7) / 8;
  int32_t *col = calloc((size_t)gw * gh * 8, sizeof(int32_t));
  if (state->frame->ref->used_size && state->frame->ref_LX_size[0] > 0) {
    const int cr = state->frame->ref_LX[0][0];
    const cu_array_t *ca = state->frame->ref->cu_arrays[cr];
    for (int gy = 0; gy < gh; ++gy)
      for (int gx = 0; gx < gw; ++gx) {
        const cu_info_t *c = &ca->data[(gx * 8) / SCU_WIDTH + ((gy * 8) / SCU_WIDTH) * (ca->width / SCU_WIDTH)];
        int32_t *o = col + ((size_t)gy * gw + gx) * 8;
        o[0] = c->type; o[1] = c->inter.mv[0][0]; o[2] = c->inter.mv[0][1]; o[3] = c->inter.mv[1][0]; o[4] = c->inter.mv[1][1]; o[5] = c->inter.mv_dir;
        for (int l = 0; l < 2; ++l)
          o[6 + l] = (c->type == CU_INTER && (c->inter.mv_dir & (1 << l)))
                         ? state->frame->ref->images[cr]->ref_pocs[state->frame->ref->ref_LXs[cr][l][c->inter.mv_ref[l]]] : -1;
      }
  }
  const uint32_t row = (uint32_t)cu_loc->y >> LOG2_LCU_WIDTH;
  int32_t hm[1 + MAX_NUM_HMVP_CANDS * 8];
  memset(hm, 0, sizeof hm);
  hm[0] = frame->hmvp_size[row];
  for (int i = 0; i < MAX_NUM_HMVP_CANDS; ++i) pack_cu(&frame->hmvp_lut[row * MAX_NUM_HMVP_CANDS + i], hm + 1 + 8 * i);
  int32_t out[MRG_MAX_NUM_CANDS][7];
  memset(out, 0, sizeof out);
  for (int i = 0; i < n && i < MRG_MAX_NUM_CANDS; ++i) {
    out[i][0] = mv_cand[i].dir; out[i][1] = mv_cand[i].ref[0]; out[i][2] = mv_cand[i].ref[1];
    out[i][3] = mv_cand[i].mv[0][0]; out[i][4] = mv_cand[i].mv[0][1]; out[i][5] = mv_cand[i].mv[1][0]; out[i][6] = mv_cand[i].mv[1][1];
  }
  const cu_info_t *cur = LCU_GET_CU_AT_PX(lcu, SUB_SCU(cu_loc->x), SUB_SCU(cu_loc->y));
  ctx[49] = (int32_t)cur->split_tree;
  rec_begin("merge", 5);
  rec_arr(A_I32, ctx, 64);
  rec_arr(A_I32, tab, (LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1) * 8);
  rec_arr(A_I32, col, (size_t)gw * gh * 8);
  rec_arr(A_I32, hm, 1 + MAX_NUM_HMVP_CANDS * 8);
  rec_arr(A_I32, out, MRG_MAX_NUM_CANDS * 7);
  free(col);
  return n;
}

/* ---- ALF: the whole per-picture process (src/alf.c:5193, called from encoderstate.c:1045 once every CTU of the picture is through SAO).
 * Recorded around it: the picture it gets and the picture it leaves, and every decision the reconstruction half (alf_reconstruct
 * :5032, apply_cc_alf_filter :1726) and the syntax need: the slice's flags and APS ids, the APSs' coded coefficients, the CTU flags,
 * filter set indices, chroma alternatives and CC-ALF controls. */
#include "alf.h"
static void planes_of(const videoframe_t *frame, uvg_pixel *y, uvg_pixel *u, uvg_pixel *v)
{
  const int W = frame->width, H = frame->height, S = frame->rec->stride;
  for (int r = 0; r < H; ++r) memcpy(y + (size_t)r * W, frame->rec->y + (size_t)r * S, sizeof(uvg_pixel) * W);
  for (int r = 0; r < H / 2; ++r) {
    memcpy(u + (size_t)r * (W / 2), frame->rec->u + (size_t)r * (S / 2), sizeof(uvg_pixel) * (W / 2));
    memcpy(v + (size_t)r * (W / 2), frame->rec->v + (size_t)r * (S / 2), sizeof(uvg_pixel) * (W / 2));
  }
}
/* the classification is freed before the process returns (alf.c:3382): taken when the luma filter first runs, through its strategy pointer */
#include "strategies/strategies-alf.h"
static alf_filter_7x7_blk_func *g_filter7_real;
static uint8_t *g_cls;
static int g_cls_taken;
/* The frame's statistics as alf_derive_stats_for_filtering (alf.c:4227) left them in the per-CTU covariances -- freed, like the classification,
 * before the process returns (alf_covariance_destroy, :5446), and untouched by the derivation in between: summed over the CTUs, per luma class and
 * per chroma plane, in the layout of uvghip_alf_cov_reduce -- the ee triangle k <= l ((k * 13 - k (k - 1) / 2 + l - k) * 16 + b0 * 4 + b1),
 * y[k][b] at 1456 + 4 k + b, pix_acc (an integer kept in a double) at 1508.  Taken with the classification (a picture no CTU of which is
 * filtered has neither). */
enum { SUMW = 1509 };
static int64_t *g_cov_luma, *g_cov_chroma;
static void take_covariances(const videoframe_t *frame)
{
  const alf_info_t *ai = frame->alf_info;
  const int n = frame->width_in_lcu * frame->height_in_lcu;
  for (int pass = 0; pass < 3; ++pass) {
    const alf_covariance *base = pass == 0 ? ai->alf_covariance_y : (pass == 1 ? ai->alf_covariance_u : ai->alf_covariance_v);
    const int ncls = pass == 0 ? MAX_NUM_ALF_CLASSES : 1, nco = pass == 0 ? MAX_NUM_ALF_LUMA_COEFF : MAX_NUM_ALF_CHROMA_COEFF;
    if (!base) continue;
    for (int ctu = 0; ctu < n; ++ctu)
      for (int c = 0; c < ncls; ++c) {
        const alf_covariance *cv = &base[ctu * ncls + c];
        int64_t *o = pass == 0 ? g_cov_luma + (size_t)c * SUMW : g_cov_chroma + (size_t)(pass - 1) * SUMW;
        for (int k = 0; k < nco; ++k)
          for (int l = k; l < nco; ++l)
            for (int b0 = 0; b0 < 4; ++b0) for (int b1 = 0; b1 < 4; ++b1)
              o[(k * 13 - k * (k - 1) / 2 + l - k) * 16 + b0 * 4 + b1] += cv->ee[k][l][b0][b1];
        for (int k = 0; k < nco; ++k) for (int b = 0; b < 4; ++b) o[1456 + 4 * k + b] += cv->y[k][b];
        o[1508] += (int64_t)cv->pix_acc;
      }
  }
}
static void filter7_hook(encoder_state_t *const state, const uvg_pixel *src_pixels, uvg_pixel *dst_pixels, const int src_stride, const int dst_stride,
                         const short *filter_set, const int16_t *fClipSet, clp_rng clp_rng, const int width, const int height, int x_pos, int y_pos,
                         int blk_dst_x, int blk_dst_y, int vb_pos, const int vb_ctu_height)
{
  if (!g_cls_taken) {
    const videoframe_t *frame = state->tile->frame;
    const int cw = (frame->width + 3) / 4, chh = (frame->height + 3) / 4;
    alf_classifier **cl = frame->alf_info->classifier;
    for (int by = 0; by < chh; ++by) for (int bx = 0; bx < cw; ++bx) g_cls[by * cw + bx] = (uint8_t)(cl[by * 4][bx * 4].class_idx | cl[by * 4][bx * 4].transpose_idx << 5);
    take_covariances(frame);
    g_cls_taken = 1;
  }
  g_filter7_real(state, src_pixels, dst_pixels, src_stride, dst_stride, filter_set, fClipSet, clp_rng, width, height, x_pos, y_pos, blk_dst_x, blk_dst_y, vb_pos, vb_ctu_height);
}
void __real_uvg_alf_enc_process(encoder_state_t *const state);
void __wrap_uvg_alf_enc_process(encoder_state_t *const state)
{
  videoframe_t *frame = state->tile->frame;
  const int W = frame->width, H = frame->height, n = frame->width_in_lcu * frame->height_in_lcu;
  uvg_pixel *pre[3], *post[3];
  for (int c = 0; c < 3; ++c) { pre[c] = malloc(sizeof(uvg_pixel) * (size_t)W * H); post[c] = malloc(sizeof(uvg_pixel) * (size_t)W * H); }
  planes_of(frame, pre[0], pre[1], pre[2]);
  g_cls = calloc((size_t)((W + 3) / 4) * ((H + 3) / 4), 1); g_cls_taken = 0;
  g_cov_luma = calloc((size_t)MAX_NUM_ALF_CLASSES * SUMW, sizeof(int64_t)); g_cov_chroma = calloc((size_t)2 * SUMW, sizeof(int64_t));
  g_filter7_real = uvg_alf_filter_7x7_blk; uvg_alf_filter_7x7_blk = filter7_hook;
  __real_uvg_alf_enc_process(state);
  uvg_alf_filter_7x7_blk = g_filter7_real;
  planes_of(frame, post[0], post[1], post[2]);
  const encoder_state_config_alf_t *sa = state->slice->alf;
  const alf_info_t *ai = frame->alf_info;
  int32_t meta[32] = {0};
  meta[0] = (int32_t)state->frame->num; meta[1] = W; meta[2] = H; meta[3] = state->encoder_control->cfg.alf_type;
  meta[4] = sa->tile_group_alf_enabled_flag[0]; meta[5] = sa->tile_group_alf_enabled_flag[1]; meta[6] = sa->tile_group_alf_enabled_flag[2];
  meta[7] = sa->tile_group_num_aps; meta[8] = sa->tile_group_chroma_aps_id;
  for (int i = 0; i < ALF_CTB_MAX_NUM_APS; ++i) meta[9 + i] = i < sa->tile_group_num_aps ? sa->tile_group_luma_aps_id[i] : -1;
  meta[17] = sa->cc_filter_param->cc_alf_filter_enabled[0]; meta[18] = sa->cc_filter_param->cc_alf_filter_enabled[1];
  meta[19] = sa->cc_filter_param->cc_alf_filter_count[0]; meta[20] = sa->cc_filter_param->cc_alf_filter_count[1];
  meta[21] = sa->tile_group_cc_alf_cb_enabled_flag; meta[22] = sa->tile_group_cc_alf_cr_enabled_flag;
  meta[23] = sa->tile_group_cc_alf_cb_aps_id; meta[24] = sa->tile_group_cc_alf_cr_aps_id;
  meta[25] = (int32_t)state->frame->poc; meta[26] = state->frame->slicetype; meta[27] = state->frame->QP;
  meta[29] = g_cls_taken;              /* the classification and the covariance sums below were taken (a CTU was filtered) */
  meta[28] = state->encoder_control->cfg.input_bitdepth;      /* the classification's activity shift is input_bitdepth + 4 (alf.c:5185): the INPUT's depth, 8 unless --input-bitdepth says otherwise */
  uint8_t *flags = calloc((size_t)n, 7);                       /* enable Y / Cb / Cr, alternative Cb / Cr, CC-ALF control Cb / Cr */
  int16_t *set_idx = calloc((size_t)n, sizeof(int16_t));
  for (int k = 0; k < n; ++k) {
    for (int c = 0; c < 3; ++c) flags[c * n + k] = ai->ctu_enable_flag[c][k];
    flags[3 * n + k] = ai->ctu_alternative[1][k]; flags[4 * n + k] = ai->ctu_alternative[2][k];
    flags[5 * n + k] = ai->cc_alf_filter_control[0][k]; flags[6 * n + k] = ai->cc_alf_filter_control[1][k];
    set_idx[k] = ai->alf_ctb_filter_index[k];
  }
  /* the APSs the slice refers to, as coded (before alf_reconstruct_coeff expands them per class) */
  enum { LN = MAX_NUM_ALF_CLASSES * MAX_NUM_ALF_LUMA_COEFF };
  int16_t *luma = calloc((size_t)ALF_CTB_MAX_NUM_APS * (2 * LN + MAX_NUM_ALF_CLASSES + 2), sizeof(int16_t));
  for (int i = 0; i < sa->tile_group_num_aps; ++i) {
    const alf_aps *a = &sa->apss[sa->tile_group_luma_aps_id[i]];
    int16_t *o = luma + (size_t)i * (2 * LN + MAX_NUM_ALF_CLASSES + 2);
    for (int k = 0; k < LN; ++k) { o[k] = a->luma_coeff[k]; o[LN + k] = a->luma_clipp[k]; }
    for (int k = 0; k < MAX_NUM_ALF_CLASSES; ++k) o[2 * LN + k] = a->filter_coeff_delta_idx[k];
    o[2 * LN + MAX_NUM_ALF_CLASSES] = (int16_t)a->num_luma_filters; o[2 * LN + MAX_NUM_ALF_CLASSES + 1] = a->non_linear_flag[0];
  }
  int16_t chroma[2 * MAX_NUM_ALF_ALTERNATIVES_CHROMA * MAX_NUM_ALF_CHROMA_COEFF + 2] = {0};
  if (sa->tile_group_chroma_aps_id >= 0 && sa->tile_group_chroma_aps_id < ALF_CTB_MAX_NUM_APS) {
    const alf_aps *a = &sa->apss[sa->tile_group_chroma_aps_id];
    for (int t = 0; t < MAX_NUM_ALF_ALTERNATIVES_CHROMA; ++t)
      for (int k = 0; k < MAX_NUM_ALF_CHROMA_COEFF; ++k) {
        chroma[t * MAX_NUM_ALF_CHROMA_COEFF + k] = a->chroma_coeff[t][k];
        chroma[(MAX_NUM_ALF_ALTERNATIVES_CHROMA + t) * MAX_NUM_ALF_CHROMA_COEFF + k] = a->chroma_clipp[t][k];
      }
    chroma[2 * MAX_NUM_ALF_ALTERNATIVES_CHROMA * MAX_NUM_ALF_CHROMA_COEFF] = (int16_t)a->num_alternatives_chroma;
    chroma[2 * MAX_NUM_ALF_ALTERNATIVES_CHROMA * MAX_NUM_ALF_CHROMA_COEFF + 1] = a->non_linear_flag[1];
  }
  int16_t cc[2 * MAX_NUM_CC_ALF_FILTERS * MAX_NUM_CC_ALF_CHROMA_COEFF];
  for (int c = 0; c < 2; ++c) for (int f = 0; f < MAX_NUM_CC_ALF_FILTERS; ++f) for (int k = 0; k < MAX_NUM_CC_ALF_CHROMA_COEFF; ++k)
    cc[(c * MAX_NUM_CC_ALF_FILTERS + f) * MAX_NUM_CC_ALF_CHROMA_COEFF + k] = sa->cc_filter_param->cc_alf_coeff[c][f][k];
  /* the two tables of the standard the fixed filter sets are made of (alf.h:46-133) */
  static int32_t fixed[64 * MAX_NUM_ALF_LUMA_COEFF + ALF_NUM_FIXED_FILTER_SETS * MAX_NUM_ALF_CLASSES];
  for (int i = 0; i < 64; ++i) for (int k = 0; k < MAX_NUM_ALF_LUMA_COEFF; ++k) fixed[i * MAX_NUM_ALF_LUMA_COEFF + k] = g_fixed_filter_set_coeff[i][k];
  for (int i = 0; i < ALF_NUM_FIXED_FILTER_SETS; ++i) for (int k = 0; k < MAX_NUM_ALF_CLASSES; ++k)
    fixed[64 * MAX_NUM_ALF_LUMA_COEFF + i * MAX_NUM_ALF_CLASSES + k] = g_class_to_filter_mapping[i][k];
  /* the classification the luma filter worked from, one byte per 4x4 block (all zero when no CTU was filtered) */
  const int cw = (W + 3) / 4, chh = (H + 3) / 4;
  uint8_t *cls = g_cls;
  rec_begin("alf", 16);
  rec_arr(A_I32, meta, 32);
  for (int c = 0; c < 3; ++c) rec_arr(A_PX, pre[c], c ? (size_t)(W / 2) * (H / 2) : (size_t)W * H);
  for (int c = 0; c < 3; ++c) rec_arr(A_PX, post[c], c ? (size_t)(W / 2) * (H / 2) : (size_t)W * H);
  rec_arr(A_U8, flags, (size_t)n * 7);
  rec_arr(A_I16, set_idx, (size_t)n);
  rec_arr(A_I16, luma, (size_t)ALF_CTB_MAX_NUM_APS * (2 * LN + MAX_NUM_ALF_CLASSES + 2));
  rec_arr(A_I16, chroma, sizeof chroma / sizeof chroma[0]);
  rec_arr(A_I16, cc, sizeof cc / sizeof cc[0]);
  rec_arr(A_I32, fixed, sizeof fixed / sizeof fixed[0]);
  rec_arr(A_U8, cls, (size_t)cw * chh);
  rec_arr(A_I64, g_cov_luma, (size_t)MAX_NUM_ALF_CLASSES * SUMW);
  rec_arr(A_I64, g_cov_chroma, (size_t)2 * SUMW);
  free(g_cov_luma); free(g_cov_chroma);
  free(cls);
  for (int c = 0; c < 3; ++c) { free(pre[c]); free(post[c]); }
  free(flags); free(set_idx); free(luma);
}

/* ---- the APS NAL units a picture is preceded by (alf.c:1610 -> encode_alf_aps :1575): every parameter set the map marks as changed,
 * with all the fields encode_alf_aps_flags (:1452-1545) writes. */
void __real_uvg_encode_alf_adaptive_parameter_set(encoder_state_t *const state);
void __wrap_uvg_encode_alf_adaptive_parameter_set(encoder_state_t *const state)
{
  const encoder_control_t *const encoder = state->encoder_control;
  const encoder_state_config_alf_t *sa = state->slice->alf;
  if (encoder->cfg.alf_type && (sa->tile_group_alf_enabled_flag[COMPONENT_Y] || sa->tile_group_cc_alf_cb_enabled_flag || sa->tile_group_cc_alf_cr_enabled_flag)) {
    const param_set_map *map = state->tile->frame->alf_param_set_map;
    for (int id = 0; id < ALF_CTB_MAX_NUM_APS; ++id) {
      if (!map[id + T_ALF_APS].b_changed) continue;
      const alf_aps *a = &map[id + T_ALF_APS].parameter_set;
      enum { LN = MAX_NUM_ALF_CLASSES * MAX_NUM_ALF_LUMA_COEFF };
      int32_t meta[16] = {(int32_t)state->frame->num, id, a->aps_id, (int32_t)a->aps_type, a->new_filter_flag[0], a->new_filter_flag[1], a->non_linear_flag[0], a->non_linear_flag[1],
                          a->num_luma_filters, a->num_alternatives_chroma, a->cc_alf_aps_param.new_cc_alf_filter[0], a->cc_alf_aps_param.new_cc_alf_filter[1],
                          a->cc_alf_aps_param.cc_alf_filter_count[0], a->cc_alf_aps_param.cc_alf_filter_count[1], encoder->cfg.alf_type, 0};
      int16_t luma[2 * LN + MAX_NUM_ALF_CLASSES], chroma[2 * MAX_NUM_ALF_ALTERNATIVES_CHROMA * MAX_NUM_ALF_CHROMA_COEFF], cc[2 * MAX_NUM_CC_ALF_FILTERS * MAX_NUM_CC_ALF_CHROMA_COEFF];
      for (int k = 0; k < LN; ++k) { luma[k] = a->luma_coeff[k]; luma[LN + k] = a->luma_clipp[k]; }
      for (int k = 0; k < MAX_NUM_ALF_CLASSES; ++k) luma[2 * LN + k] = a->filter_coeff_delta_idx[k];
      for (int t = 0; t < MAX_NUM_ALF_ALTERNATIVES_CHROMA; ++t) for (int k = 0; k < MAX_NUM_ALF_CHROMA_COEFF; ++k) {
        chroma[t * MAX_NUM_ALF_CHROMA_COEFF + k] = a->chroma_coeff[t][k]; chroma[(MAX_NUM_ALF_ALTERNATIVES_CHROMA + t) * MAX_NUM_ALF_CHROMA_COEFF + k] = a->chroma_clipp[t][k];
      }
      for (int c = 0; c < 2; ++c) for (int f = 0; f < MAX_NUM_CC_ALF_FILTERS; ++f) for (int k = 0; k < MAX_NUM_CC_ALF_CHROMA_COEFF; ++k)
        cc[(c * MAX_NUM_CC_ALF_FILTERS + f) * MAX_NUM_CC_ALF_CHROMA_COEFF + k] = a->cc_alf_aps_param.cc_alf_coeff[c][f][k];
      rec_begin("aps", 4);
      rec_arr(A_I32, meta, 16); rec_arr(A_I16, luma, 2 * LN + MAX_NUM_ALF_CLASSES); rec_arr(A_I16, chroma, sizeof chroma / sizeof chroma[0]); rec_arr(A_I16, cc, sizeof cc / sizeof cc[0]);
    }
  }
  __real_uvg_encode_alf_adaptive_parameter_set(state);
}

int main(int argc, char **argv)
{
  if (getenv("CTU_DUMP_MERGE_EVERY")) g_merge_every = atoi(getenv("CTU_DUMP_MERGE_EVERY"));
  if (getenv("CTU_DUMP_CU_INTER")) g_cu_inter = atoi(getenv("CTU_DUMP_CU_INTER"));
  if (argc < 7) { fprintf(stderr, "usage: %s in.yuv W H frames out.bin out.266 [opt val]...\n", argv[0]); return 2; }
  const int W = atoi(argv[2]), H = atoi(argv[3]), nframes = atoi(argv[4]);
  FILE *in = fopen(argv[1], "rb");
  g_out = fopen(argv[5], "wb");
  FILE *bs = fopen(argv[6], "wb");
  if (!in || !g_out || !bs) { perror("open"); return 2; }
  const uvg_api *api = uvg_api_get(UVG_BIT_DEPTH);
  uvg_config *cfg = api->config_alloc();
  api->config_init(cfg);
  char res[64];
  snprintf(res, sizeof res, "%dx%d", W, H);
  int ok = api->config_parse(cfg, "input-res", res);
  ok &= api->config_parse(cfg, "threads", "0");
  ok &= api->config_parse(cfg, "owf", "0");
  ok &= api->config_parse(cfg, "cpuid", "0");
  for (int i = 7; i + 1 < argc; i += 2) {
    if (!api->config_parse(cfg, argv[i], argv[i + 1])) { fprintf(stderr, "bad option %s %s\n", argv[i], argv[i + 1]); return 2; }
  }
  if (!ok) { fprintf(stderr, "config failed\n"); return 2; }
  uvg_encoder *enc = api->encoder_open(cfg);
  if (!enc) { fprintf(stderr, "encoder_open failed\n"); return 2; }
  int fed = 0, done = 0;
  for (;;) {
    uvg_picture *pic = NULL;
    if (fed < nframes) {
      pic = api->picture_alloc(W, H);
      for (int p = 0; p < 3; ++p) {
        const int w = p ? W / 2 : W, h = p ? H / 2 : H;
        uvg_pixel *dst = p == 0 ? pic->y : (p == 1 ? pic->u : pic->v);
        const int stride = p ? pic->stride / 2 : pic->stride;
        for (int y = 0; y < h; ++y)
          if (fread(dst + (size_t)y * stride, sizeof(uvg_pixel), (size_t)w, in) != (size_t)w) { fprintf(stderr, "short read\n"); return 2; }
      }
      pic->pts = fed;
      ++fed;
    }
    uvg_data_chunk *chunks = NULL;
    uint32_t len = 0;
    uvg_picture *rec = NULL, *src = NULL;
    uvg_frame_info info;
    if (!api->encoder_encode(enc, pic, &chunks, &len, &rec, &src, &info)) { fprintf(stderr, "encode failed\n"); return 2; }
    api->picture_free(pic);
    if (chunks) {
      for (uvg_data_chunk *c = chunks; c; c = c->next) fwrite(c->data, 1, c->len, bs);
      api->chunk_free(chunks);
      ++done;
    }
    if (rec) {
      /* the picture after all in-loop filters (deblocking + SAO; ALF and LMCS are off by default, src/cfg.c:62-67) */
      int32_t meta[3] = {(int32_t)rec->pts, W, H};
      rec_begin("final", 4);
      rec_arr(A_I32, meta, 3);
      for (int p = 0; p < 3; ++p) {
        const int w = p ? W / 2 : W, h = p ? H / 2 : H;
        const uvg_pixel *srcp = p == 0 ? rec->y : (p == 1 ? rec->u : rec->v);
        const int stride = p ? rec->stride / 2 : rec->stride;
        uvg_pixel *tmp = malloc(sizeof(uvg_pixel) * (size_t)w * h);
        for (int y = 0; y < h; ++y) memcpy(tmp + (size_t)y * w, srcp + (size_t)y * stride, sizeof(uvg_pixel) * (size_t)w);
        rec_arr(A_PX, tmp, (size_t)w * h);
        free(tmp);
      }
    }
    api->picture_free(rec); api->picture_free(src);
    if (!pic && !chunks) break;
    if (done >= nframes) break;
  }
  api->encoder_close(enc);
  api->config_destroy(cfg);
  fclose(g_out); fclose(bs); fclose(in);
  fprintf(stderr, "ctu_dump: %d frames\n", done);
  return 0;
}
