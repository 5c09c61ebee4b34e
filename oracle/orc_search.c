/*
 * oracle/orc_search.c -- restatement of the closed-loop intra CTU search of an all-intra picture (the caller of the hot path):
 *   uvg_search_lcu / init_lcu_t / copy_lcu_to_cu_data      src/search.c:2230-2479
 *   search_cu (split / no-split RD decision, work tree)    src/search.c:1299-2221   (+ helpers :73-400, :1075-1286)
 *   cu_rd_cost_tr_split_accurate, uvg_cu_rd_cost_chroma    src/search.c:625-986
 *   uvg_search_cu_intra / search_intra_rough / count_bits  src/search_intra.c:949-1228, 1771-1988   (rd = 0: no RDO refinement)
 *   uvg_intra_build_reference(_inner/_any), recon leaf     src/intra.c:756-1370, 1537-1745
 *   uvg_intra_get_dir_luma_predictor                       src/intra.c:88-188
 *   uvg_quantize_lcu_residual / quantize_tr_residual       src/transform.c:1283-1603
 *   uvg_quantize_residual (RDOQ branch)                    src/strategies/generic/quant-generic.c:460-612
 *   uvg_mock_encode_coding_unit, uvg_write_split_flag, uvg_encode_intra_luma_coding_unit, encode_chroma_intra_cu,
 *   uvg_encode_coding_tree / encode_transform_coeff (model adaptation of the real coder)   src/encode_coding_tree.c
 *   uvg_get_possible_splits, uvg_get_split_locs, uvg_count_available_edge_cus, uvg_derive_mode_type_cond   src/cu.c:323-537
 *   WPP model hand-over between CTU rows                   src/encoderstate.c:863-976
 * Configuration subset (the reference's --preset medium, -p 1, 4:2:0): I slices, pu-depth-intra = min..max, quad-tree splits only
 * (mtt depth 0), rd = 0, rdoq = 1, no sign hiding, no transform skip / MTS / LFNST / ISP / MRL / MIP / CCLM / JCCR / dual tree,
 * combine-intra-cus = 1, cu-split-termination = zero, WPP on.  Everything else is refused by the caller.
 * Costs are double precision in the reference's order of operations.
 * Pinned by tests/golden/ref_ctu_*.npz (decisions, reconstruction, levels and models of every CTU of reference-run pictures,
 * tools/refcheck/ctu_dump.c).  TEST INFRASTRUCTURE ONLY (see orc_common.h).
 */
#include "orc_common.h"
#include "orc_ctx_init.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* ---- other oracle files ---- */
#define REF_LEN 400
void ORC_FN(intra_filter_refs)(const orc_px *top, const orc_px *left, int w, int h, orc_px *ftop, orc_px *fleft);
int ORC_FN(intra_predict)(int mode, int is_chroma, int width, int height, const orc_px *top, const orc_px *left, const orc_px *ftop,
                          const orc_px *fleft, orc_px *dst);
unsigned ORC_FN(satd_nxn)(const orc_px *, const orc_px *, int);
unsigned ORC_FN(sad_nxn)(const orc_px *, const orc_px *, int);
unsigned ORC_FN(pixels_calc_ssd)(const orc_px *ref, const orc_px *rec, int ref_stride, int rec_stride, int w, int h);
void ORC_FN(dct_nxn)(int bitdepth, int n, const int16_t *in, int16_t *out);
void ORC_FN(idct_nxn)(int bitdepth, int n, const int16_t *in, int16_t *out);
void ORC_FN(dequant)(const int16_t *q_coef, int16_t *coef, int width, int height, int bitdepth, int qp_scaled, int transform_skip);
int ORC_FN(rdoq)(const int16_t *coef, int16_t *dest_coeff, int width, int height, int color, int block_type, int cbf_u, int lfnst_idx, int mts_idx,
                 int qp_scaled, double lambda, const void *ctx_snapshot);

enum { NMODELS = 257, NRES = 244, NINTER = 18, NM = NMODELS + NINTER };
enum { M_CBF_LUMA = 234, M_CBF_CB = 238, M_CBF_CR = 240, M_ROOT_CBF = 243, M_SPLIT = 244, M_MPM = 253, M_PLANAR = 254, M_CHROMA_PRED = 256,
       /* the inter syntax (P / B slices), order of snapshot_inter in tools/refcheck/ctu_dump.c */
       M_SKIP = 257, M_PRED_MODE = 260, M_MERGE_FLAG = 262, M_MERGE_IDX = 263, M_INTER_DIR = 264, M_REF_PIC = 270, M_MVD = 272, M_MVP_IDX = 274 };

/* the exported layouts: the 257 models of an intra picture's search (rounds 1-3) and, beside them, the 18 of the inter syntax */
typedef struct orc_models_ext { uint16_t state0[NMODELS], state1[NMODELS]; uint8_t rate[NMODELS]; } orc_models_ext;
typedef struct orc_models_inter { uint16_t state0[NINTER], state1[NINTER]; uint8_t rate[NINTER]; } orc_models_inter;
/* internal: one index space */
typedef struct orc_models { uint16_t state0[NM], state1[NM]; uint8_t rate[NM]; } orc_models;
typedef struct orc_cabac_models { uint16_t state0[NRES], state1[NRES]; uint8_t rate[NRES]; } orc_cabac_models;
double ORC_FN(coeff_cost)(const int16_t *coeff, int width, int height, int color, const orc_cabac_models *models_in, uint32_t *flags_out,
                          orc_cabac_models *models_out);

typedef struct orc_rdoq_ctx {
  uint8_t sig_group[2][2], sig[2][12], par[2][21], gt1[2][21], gt2[2][21], last_x[2][20], last_y[2][20];
  uint8_t cbf_luma[4], cbf_cb[2], cbf_cr[3], root_cbf;
} orc_rdoq_ctx;

/* what the reference reads from encoder_state_t / encoder_control_t on this path */
typedef struct orc_search_params {
  int32_t pic_w, pic_h;
  int32_t qp;                 /* state->qp */
  int32_t qp_c;               /* encoder->qp_map[0][qp] (transform.c:158) */
  int32_t depth_min, depth_max;   /* pu-depth-intra */
  int32_t wpp, combine_intra_cus, rough_levels;   /* cfg.wpp, cfg.combine_intra_cus, cfg.intra_rough_search_levels */
  int32_t rd;                 /* cfg.rdo: 0 or 1 (they differ in the P / B CU's skip of the intra search, search.c:1413-1419) */
  double lambda, lambda_sqrt, c_lambda, cw_u, cw_v;   /* state->lambda ..., state->chroma_weights[1..2] */
} orc_search_params;

/* cu_info_t reduced to what this path reads or writes (cu.h:134-198) */
typedef struct s_cu {
  uint8_t type, log2_w, log2_h, log2_cw, log2_ch, cbf;
  int8_t mode, mode_chroma;
  uint8_t luma_deblocking, chroma_deblocking, qp, pad;
  uint32_t split_tree, mode_type_tree;
  /* P / B slices (cu_info_t: skipped, merged, merge_idx, root_cbf; inter.mv_dir, mv_ref, mv_cand0/1, mv) */
  uint8_t skipped, merged, merge_idx, root_cbf, mv_dir, mv_cand0, mv_cand1, pad2;
  uint8_t mv_ref[2], pad3[2];
  int32_t mv[2][2];
} s_cu;
enum { CU_NOTSET = 0, CU_INTRA = 1, CU_INTER = 2 };

/* what the reference reads of encoder_state_t / the picture's reference lists on the inter path (all of it host-side bookkeeping of
 * the encoder: the GOP structure, the reference picture sets and the per-picture QP / lambda stay with the caller) */
typedef struct orc_inter_frame {
  int32_t slice_type;             /* state->frame->slicetype: 0 B, 1 P, 2 I */
  int32_t poc;
  int32_t n_refs;                 /* state->frame->ref->used_size */
  int32_t ref_pocs[16];           /* state->frame->ref->pocs */
  int32_t l_size[2];              /* state->frame->ref_LX_size */
  int32_t l[2][16];               /* state->frame->ref_LX: indices into the reference array */
  int32_t tmvp, max_merge, merge_level, bipred, fme_level, early_skip, depth_inter_min, depth_inter_max;
  int32_t ref_cu_stride, frame_qp;    /* frame_qp: state->frame->QP, what the slice's context models are initialised with */
  const orc_px *ref_y[16], *ref_u[16], *ref_v[16];      /* the reference pictures after the in-loop filters, pic_w x pic_h, tightly packed */
  const int32_t *ref_cu[16];      /* per reference picture and 4x4 (stride ref_cu_stride): type, mv[2][2], mv_dir, the POC the L0 / L1 vector points to */
  /* cfg.owf != 0 (with cfg.wpp): pictures are coded while their reference pictures are still being coded, so no vector may reach beyond what is
   * final there -- fracmv_within_tile, search_inter.c:94-149.  owf_margin: the samples the in-loop filters still change above the frontier
   * (SAO_DELAY_PX 10 with cfg.sao_type, else DEBLOCK_DELAY_PX 8 with deblocking, else 0; global.h:240-252) */
  int32_t owf, owf_margin;
} orc_inter_frame;
enum { NO_SPLIT = 0, QT_SPLIT = 1 };
enum { MODE_TYPE_ALL = 0, MODE_TYPE_INTER = 1, MODE_TYPE_INTRA = 2 };
enum { EDGE_VER = 1, EDGE_HOR = 2 };     /* filter.h edge_dir */

#define LCU 64
#define LCU_C 32
#define TCW 17
typedef struct s_lcu {          /* lcu_t (cu.h:357-395) */
  orc_px top_y[97], top_u[49], top_v[49], left_y[97], left_u[49], left_v[49];
  orc_px ref_y[LCU * LCU], ref_u[LCU_C * LCU_C], ref_v[LCU_C * LCU_C];
  orc_px rec_y[LCU * LCU], rec_u[LCU_C * LCU_C], rec_v[LCU_C * LCU_C];
  int16_t coeff_y[LCU * LCU], coeff_u[LCU_C * LCU_C], coeff_v[LCU_C * LCU_C];
  s_cu cu[TCW * TCW + 1];
} s_lcu;
#define CU_AT(l, x, y) (&(l)->cu[TCW + 1 + ((x) >> 2) + ((y) >> 2) * TCW])

typedef struct s_loc { int x, y, lx, ly, w, h, cw, ch; } s_loc;      /* cu_loc_t */
typedef struct s_tree { uint32_t split_tree, mode_type_tree; int depth, mtt_depth, implicit_mtt_depth, part_index; } s_tree;   /* split_tree_t */

typedef struct s_cabac { orc_models m; int update; } s_cabac;       /* cabac_data_t: models + the update flag (only_count is 1 in the search) */

typedef struct s_state {
  const orc_search_params *p;
  s_cabac search;               /* state->search_cabac */
  orc_rdoq_ctx rdoq;            /* CTX_STATE of state->cabac at the CTU's start: uvg_rdoq prices with THAT (rdo.c:1462) */
  double c_lambda;              /* state->c_lambda (temporarily replaced in uvg_quantize_lcu_residual, transform.c:1575) */
  /* P / B pictures */
  const orc_inter_frame *fr;    /* NULL: an intra picture */
  const orc_px *src_y;          /* frame->source->y */
  const int32_t *col;           /* the collocated picture (L0[0]) on the 8x8 grid, layout of orc_inter_cand.c */
  int32_t *hmvp;                /* the CTU row's history table: [size, 5 x 8 ints] */
  const s_cu *cua;              /* the picture's cu array (the real coder's predictors) */
  int cu_stride;
} s_state;

static void loc_ctor(s_loc *l, int x, int y, int w, int h)
{
  l->x = x; l->y = y; l->lx = x % LCU; l->ly = y % LCU; l->w = w; l->h = h; l->cw = w >> 1; l->ch = h >> 1;
}

/* ------------------------------------------------------------------------------------------------------------------ CABAC -- */
static float g_fbits[512];
static int g_fbits_ready = 0;
static void fbits_init(void)
{
  if (g_fbits_ready) return;
  for (int i = 0; i < 512; ++i) {
    const double p1 = (2 * (i >> 1) + 1) / 512.0;
    g_fbits[i] = (float)(floor(-log2((i & 1) ? p1 : 1.0 - p1) * 32768.0 + 0.5) / 32768.0);
  }
  g_fbits_ready = 1;
}
static inline int ctx_state(const orc_models *m, int c) { return (m->state0[c] + m->state1[c]) >> 8; }
static inline double ctx_fbits(const orc_models *m, int c, int bin) { return g_fbits[(ctx_state(m, c) << 1) ^ bin]; }   /* CTX_ENTROPY_FBITS */
static void ctx_update(orc_models *m, int c, int bin)           /* CTX_UPDATE, cabac.h:182-193 */
{
  const int rate0 = m->rate[c] >> 4, rate1 = m->rate[c] & 15;
  const unsigned mask0 = (~(~0u << 10)) << 5, mask1 = (~(~0u << 14)) << 1;
  m->state0[c] = (uint16_t)(m->state0[c] - ((m->state0[c] >> rate0) & mask0));
  m->state1[c] = (uint16_t)(m->state1[c] - ((m->state1[c] >> rate1) & mask1));
  if (bin) {
    m->state0[c] = (uint16_t)(m->state0[c] + ((0x7fffu >> rate0) & mask0));
    m->state1[c] = (uint16_t)(m->state1[c] + ((0x7fffu >> rate1) & mask1));
  }
}
/* CABAC_FBITS_UPDATE with only_count = 1 */
extern orc_cabac_sim ORC_FN(cabac_sim);
void ORC_FN(cabac_sim_bin)(int state, int bin);
void ORC_FN(cabac_sim_ep)(uint32_t bin);
void ORC_FN(cabac_sim_eps)(uint32_t bin_values, int num_bins);
void ORC_FN(cabac_sim_start)(void);
void ORC_FN(cabac_sim_row_end)(void);
static void fbits_update(s_cabac *cb, int c, int bin, double *bits)
{
  *bits += ctx_fbits(&cb->m, c, bin);
  if (ORC_FN(cabac_sim).on) { ORC_FN(cabac_sim).regular_fbits += ctx_fbits(&cb->m, c, bin); ORC_FN(cabac_sim_bin)(ctx_state(&cb->m, c), bin); }
  if (cb->update) ctx_update(&cb->m, c, bin);
}

static void models_init(orc_models *m, int qp, int slice)      /* uvg_init_contexts / uvg_ctx_init, context.c:471-500 */
{
  memset(m, 0, sizeof *m);
  for (int i = 0; i < NM; ++i) {
    const int v = i < NMODELS ? k_ctx_init[slice][i] : k_ctx_init_inter[slice][i - NMODELS];
    if (v == 255) continue;
    const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
    int s = ((slope * (qp - 16)) >> 1) + offset;
    s = s < 1 ? 1 : (s > 127 ? 127 : s);
    const int p1 = s << 8;
    m->state0[i] = (uint16_t)(p1 & ((~(~0u << 10)) << 5));
    m->state1[i] = (uint16_t)(p1 & ((~(~0u << 14)) << 1));
    m->rate[i] = i < NMODELS ? k_ctx_init[3][i] : k_ctx_init_inter[3][i - NMODELS];
  }
}
static void models_to_ext(const orc_models *m, orc_models_ext *e, orc_models_inter *x)
{
  memcpy(e->state0, m->state0, sizeof e->state0); memcpy(e->state1, m->state1, sizeof e->state1); memcpy(e->rate, m->rate, sizeof e->rate);
  if (x) { memcpy(x->state0, m->state0 + NMODELS, sizeof x->state0); memcpy(x->state1, m->state1 + NMODELS, sizeof x->state1); memcpy(x->rate, m->rate + NMODELS, sizeof x->rate); }
}
static void models_from_ext(orc_models *m, const orc_models_ext *e)
{
  memset(m, 0, sizeof *m);
  memcpy(m->state0, e->state0, sizeof e->state0); memcpy(m->state1, e->state1, sizeof e->state1); memcpy(m->rate, e->rate, sizeof e->rate);
}

static void rdoq_ctx_from(const orc_models *m, orc_rdoq_ctx *c)
{
  memset(c, 0, sizeof *c);
#define ST(i) ((uint8_t)ctx_state(m, (i)))
  for (int i = 0; i < 2; ++i) { c->sig_group[0][i] = ST(i); c->sig_group[1][i] = ST(2 + i); }
  for (int i = 0; i < 12; ++i) c->sig[0][i] = ST(4 + i);
  for (int i = 0; i < 8; ++i) c->sig[1][i] = ST(16 + i);
  for (int i = 0; i < 21; ++i) { c->par[0][i] = ST(28 + i); c->gt1[0][i] = ST(70 + i); c->gt2[0][i] = ST(112 + i); }
  for (int i = 0; i < 11; ++i) { c->par[1][i] = ST(49 + i); c->gt1[1][i] = ST(91 + i); c->gt2[1][i] = ST(133 + i); }
  for (int i = 0; i < 20; ++i) { c->last_x[0][i] = ST(154 + i); c->last_y[0][i] = ST(194 + i); }
  for (int i = 0; i < 3; ++i) { c->last_x[1][i] = ST(174 + i); c->last_y[1][i] = ST(214 + i); }
  for (int i = 0; i < 4; ++i) c->cbf_luma[i] = ST(M_CBF_LUMA + i);
  for (int i = 0; i < 2; ++i) c->cbf_cb[i] = ST(M_CBF_CB + i);
  for (int i = 0; i < 3; ++i) c->cbf_cr[i] = ST(M_CBF_CR + i);
  c->root_cbf = ST(243);
#undef ST
}

/* uvg_get_coeff_cost (rdo.c:393-454, not the fast estimate) on a block gathered from the CTU coefficient array (COEFF_ORDER_CU).
 * The coder runs on a copy of the models; the copy is kept only when cabac->update is set (rdo.c:354). */
static double coeff_cost_cu(s_cabac *cb, const int16_t *plane, int lcu_stride, int lx, int ly, int w, int h, int color)
{
  int16_t blk[32 * 32];
  int any = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) { blk[y * w + x] = plane[(ly + y) * lcu_stride + lx + x]; any |= blk[y * w + x] != 0; }
  if (!any) return 0;
  orc_cabac_models in, out;
  memcpy(in.state0, cb->m.state0, sizeof in.state0); memcpy(in.state1, cb->m.state1, sizeof in.state1); memcpy(in.rate, cb->m.rate, sizeof in.rate);
  uint32_t flags;
  const double bits = ORC_FN(coeff_cost)(blk, w, h, color, &in, &flags, &out);
  if (cb->update) { memcpy(cb->m.state0, out.state0, sizeof out.state0); memcpy(cb->m.state1, out.state1, sizeof out.state1); }
  return bits;
}

/* --------------------------------------------------------------------------------------------------------- split geometry -- */
/* uvg_get_possible_splits (cu.c:412-514) with max_btt_depth = 0, min_qt_size = 4: only NO_SPLIT / QT_SPLIT can be true */
static int possible_splits(const s_state *st, const s_loc *loc, s_tree tree, int can[6])
{
  const int right_ok = st->p->pic_w >= loc->x + loc->w, bottom_ok = st->p->pic_h >= loc->y + loc->h;
  const int implicit = !(right_ok && bottom_ok);      /* uvg_get_implicit_split with max_mtt_depth 0: QT_SPLIT */
  for (int i = 0; i < 6; ++i) can[i] = 1;
  const int last_split = (int)((tree.split_tree >> ((tree.depth - 1 > 0 ? tree.depth - 1 : 0) * 3)) & 7);
  if (tree.depth != 0 && last_split != QT_SPLIT) can[QT_SPLIT] = 0;
  if (loc->w <= 4) can[QT_SPLIT] = 0;
  if (implicit) {
    can[NO_SPLIT] = can[4] = can[5] = 0;
    can[2] = can[3] = 0;                              /* implicitSplit == QT_SPLIT */
    if (!can[2] && !can[3] && !can[QT_SPLIT]) can[QT_SPLIT] = 1;
    return 1;
  }
  can[2] = can[3] = can[4] = can[5] = 0;              /* can_btt = mtt_depth < max_btd = 0 is false */
  return 0;
}

/* uvg_write_split_flag (encode_coding_tree.c:1240-1363): only split_cu_flag exists when the multi-type splits are off */
static void write_split_flag(const s_state *st, s_cabac *cb, const s_cu *left, const s_cu *above, const s_loc *loc, s_tree tree, int *is_implicit_out,
                             double *bits_out)
{
  double bits = 0;
  int can[6];
  const int is_implicit = possible_splits(st, loc, tree, can);
  const int allow_split = can[1] || can[2] || can[3] || can[4] || can[5];
  const int split_flag = (int)((tree.split_tree >> (tree.depth * 3)) & 7);
  *is_implicit_out = is_implicit;
  if (can[NO_SPLIT] && allow_split) {
    int split_model = 0;
    if (left && (1 << left->log2_h) < loc->h) split_model++;
    if (above && (1 << above->log2_w) < loc->w) split_model++;
    unsigned split_num = 0;
    if (can[QT_SPLIT]) split_num += 2;
    if (split_num > 0) split_num--;
    split_model += 3 * (int)(split_num >> 1);
    fbits_update(cb, M_SPLIT + split_model, split_flag != NO_SPLIT, &bits);
  }
  /* qt_split_flag / mtt flags: need a binary or ternary split to be possible (:1300-1360) -- never here */
  if (bits_out) *bits_out += bits;
}

/* uvg_derive_mode_type_cond (cu.c:388-410), I slice, single tree, 4:2:0, quad split */
static int derive_mode_type_cond(const s_loc *loc, int mode_type)
{
  if (mode_type != MODE_TYPE_ALL) return 0;              /* MODE_TYPE_INHERIT */
  if (loc->w * loc->h == 64) return 1;                   /* MODE_TYPE_INFER */
  return 0;
}

/* ------------------------------------------------------------------------------------------------------ intra prediction -- */
/* uvg_count_available_edge_cus (cu.c:516-537) */
static int count_available_edge_cus(const s_loc *loc, const s_lcu *lcu, int left)
{
  if ((left && loc->x == 0) || (!left && loc->y == 0)) return 0;
  if (left && loc->lx == 0) return (LCU - loc->ly) / 4;
  if (!left && loc->ly == 0) return loc->w / 2;
  int amount = left ? loc->h & ~3 : loc->w & ~3;
  if (left) {
    const s_cu *cu = CU_AT(lcu, loc->lx, loc->ly);
    if (loc->ly == 0 && loc->lx == 32 && cu->log2_h == 6 && cu->log2_w == 6) return 8;
    while (loc->ly + amount < LCU && CU_AT(lcu, loc->lx - 4, loc->ly + amount)->type != CU_NOTSET) amount += 4;
    return (amount / 4) > (loc->h / 4) ? amount / 4 : loc->h / 4;
  }
  while (loc->lx + amount < LCU && CU_AT(lcu, loc->lx + amount, loc->ly - 4)->type != CU_NOTSET) amount += 4;
  return (amount / 4) > (loc->w / 4) ? amount / 4 : loc->w / 4;
}

/* uvg_intra_build_reference (intra.c:1344): _inner when the block touches neither picture edge, _any otherwise.
 * multi_ref_idx = 0, no ISP, no dual tree.  top[]/left[]: index 0 = corner. */
static void build_reference(const s_state *st, const s_loc *loc, int color, const s_lcu *lcu, orc_px *top, orc_px *left)
{
  const int is_chroma = color != 0;
  const int width = is_chroma ? loc->cw : loc->w, height = is_chroma ? loc->ch : loc->h;
  const int lw = LCU >> is_chroma;
  const int px_x = loc->lx >> is_chroma, px_y = loc->ly >> is_chroma;
  const orc_px *left_ref = color == 0 ? &lcu->left_y[1] : (color == 1 ? &lcu->left_u[1] : &lcu->left_v[1]);
  const orc_px *top_ref = color == 0 ? &lcu->top_y[1] : (color == 1 ? &lcu->top_u[1] : &lcu->top_v[1]);
  const orc_px *rec = color == 0 ? lcu->rec_y : (color == 1 ? lcu->rec_u : lcu->rec_v);
  const orc_px *top_border = px_y ? &rec[px_x + (px_y - 1) * lw] : &top_ref[px_x];
  const orc_px *left_border;
  int left_stride;
  if (px_x) { left_border = &rec[px_x - 1 + px_y * lw]; left_stride = lw; }
  else { left_border = &left_ref[px_y]; left_stride = 1; }
  for (int i = 0; i < REF_LEN; ++i) top[i] = left[i] = 0;
  const int log2_ratio = orc_log2i(width) - orc_log2i(height);

  if (loc->x > 0 && loc->y > 0) {                                   /* uvg_intra_build_reference_inner, intra.c:1065-1341 */
    if (px_x == 0) left[0] = top[0] = left_border[-1 * left_stride];
    else left[0] = top[0] = top_border[-1];
    int avail_left = count_available_edge_cus(loc, lcu, 1) * (is_chroma ? 2 : 4);
    if (avail_left > 2 * height) avail_left = 2 * height;           /* MIN(.., cu_height + pu height) */
    if (avail_left > ((st->p->pic_h - loc->y) >> is_chroma)) avail_left = (st->p->pic_h - loc->y) >> is_chroma;
    int i = 0;
    if (px_y % 4 != 0 || avail_left % 4 != 0) {
      do { left[i + 1] = left_border[i * left_stride]; i += 1; } while (i < avail_left);
    } else {
      do {
        left[i + 1] = left_border[(i + 0) * left_stride]; left[i + 2] = left_border[(i + 1) * left_stride];
        left[i + 3] = left_border[(i + 2) * left_stride]; left[i + 4] = left_border[(i + 3) * left_stride];
        i += 4;
      } while (i < avail_left);
    }
    orc_px nearest = left[i];
    int s = -log2_ratio > 0 ? -log2_ratio : 0;
    int total = height * 2 + ((height << s) + 2);
    if (total > 358 - 2) total = 358 - 2;                           /* INTRA_REF_LENGTH - 2 */
    for (; i < total; i += 4) { left[i + 1] = nearest; left[i + 2] = nearest; left[i + 3] = nearest; left[i + 4] = nearest; }

    int avail_top = count_available_edge_cus(loc, lcu, 0) * (is_chroma ? 2 : 4);
    if (avail_top > 2 * width) avail_top = 2 * width;
    if (avail_top > ((st->p->pic_w - loc->x) >> is_chroma)) avail_top = (st->p->pic_w - loc->x) >> is_chroma;
    if (st->p->wpp && px_y == 0 && avail_top > lw - px_x) avail_top = lw - px_x;
    i = 0;
    do { top[i + 1] = top_border[i]; i += 1; } while (i < avail_top);
    nearest = top[i];
    total = width * 2 + ((width << s) + 2);
    if (total > 358 - 2) total = 358 - 2;
    for (; i < total; i += 4) { top[i + 1] = nearest; top[i + 2] = nearest; top[i + 3] = nearest; top[i + 4] = nearest; }
    return;
  }

  /* uvg_intra_build_reference_any, intra.c:756-1063 */
  const orc_px dc_val = (orc_px)(1 << (ORC_BIT_DEPTH - 1));
  int s = -log2_ratio > 0 ? -log2_ratio : 0;
  int ext = (height << s) + 2;
  if (loc->x > 0) {
    int avail_left = count_available_edge_cus(loc, lcu, 1) * (is_chroma ? 2 : 4);
    if (avail_left > 2 * height) avail_left = 2 * height;
    if (avail_left > ((st->p->pic_h - loc->y) >> is_chroma)) avail_left = (st->p->pic_h - loc->y) >> is_chroma;
    for (int i = 0; i < avail_left; ++i) left[i + 1] = left_border[i * left_stride];
    const orc_px nearest = left_border[(avail_left - 1) * left_stride];
    int total = height * 2 + ext;
    if (total > 358) total = 358;
    for (int i = avail_left; i < total; ++i) left[i + 1] = nearest;
  } else {
    const orc_px nearest = loc->y > 0 ? top_border[0] : dc_val;
    int total = height * 2 + ext;
    if (total > 358) total = 358;
    for (int i = 0; i < total; ++i) left[i + 1] = nearest;
  }
  left[0] = top[0] = left[1];                                       /* x == 0 or y == 0: "copy reference clockwise" (:1030-1034) */
  s = log2_ratio > 0 ? log2_ratio : 0;
  ext = (width << s) + 2;
  if (loc->y > 0) {
    int avail_top = count_available_edge_cus(loc, lcu, 0) * (is_chroma ? 2 : 4);
    if (avail_top > 2 * width) avail_top = 2 * width;
    if (avail_top > ((st->p->pic_w - loc->x) >> is_chroma)) avail_top = (st->p->pic_w - loc->x) >> is_chroma;
    for (int i = 0; i < avail_top; ++i) top[i + 1] = top_border[i];
    const orc_px nearest = top_border[avail_top - 1];
    int total = width * 2 + ext;
    if (total > 358) total = 358;
    for (int i = avail_top; i < total; ++i) top[i + 1] = nearest;
  } else {
    const orc_px nearest = loc->x > 0 ? left_border[0] : dc_val;
    int total = width * 2 + ext;
    if (total > 358) total = 358;
    for (int i = 0; i < total; ++i) top[i + 1] = nearest;
  }
}

/* uvg_intra_get_dir_luma_predictor (intra.c:88-188), MIP off */
static int get_dir_luma_predictor(int y, int8_t *preds, const s_cu *left_pu, const s_cu *above_pu)
{
  int n = 0;
  int left_dir = 0, above_dir = 0;
  if (left_pu && left_pu->type == CU_INTRA) left_dir = left_pu->mode;
  if (above_pu && above_pu->type == CU_INTRA && y % LCU != 0) above_dir = above_pu->mode;
  const int offset = 61, mod = 64;
  preds[0] = 0; preds[1] = 1; preds[2] = 50; preds[3] = 18; preds[4] = 46; preds[5] = 54;
  if (left_dir == above_dir) {
    n = 1;
    if (left_dir > 1) {
      preds[0] = 0; preds[1] = (int8_t)left_dir;
      preds[2] = (int8_t)(((left_dir + offset) % mod) + 2); preds[3] = (int8_t)(((left_dir - 1) % mod) + 2);
      preds[4] = (int8_t)(((left_dir + offset - 1) % mod) + 2); preds[5] = (int8_t)((left_dir % mod) + 2);
    }
  } else {
    n = 2;
    if (left_dir > 1 && above_dir > 1) {
      preds[0] = 0; preds[1] = (int8_t)left_dir; preds[2] = (int8_t)above_dir;
      const int mx = preds[1] > preds[2] ? 1 : 2, mn = preds[1] > preds[2] ? 2 : 1;
      const int diff = preds[mx] - preds[mn];
      if (diff == 1) {
        preds[3] = (int8_t)(((preds[mn] + offset) % mod) + 2); preds[4] = (int8_t)(((preds[mx] - 1) % mod) + 2);
        preds[5] = (int8_t)(((preds[mn] + offset - 1) % mod) + 2);
      } else if (diff >= 62) {
        preds[3] = (int8_t)(((preds[mn] - 1) % mod) + 2); preds[4] = (int8_t)(((preds[mx] + offset) % mod) + 2);
        preds[5] = (int8_t)((preds[mn] % mod) + 2);
      } else if (diff == 2) {
        preds[3] = (int8_t)(((preds[mn] - 1) % mod) + 2); preds[4] = (int8_t)(((preds[mn] + offset) % mod) + 2);
        preds[5] = (int8_t)(((preds[mx] - 1) % mod) + 2);
      } else {
        preds[3] = (int8_t)(((preds[mn] + offset) % mod) + 2); preds[4] = (int8_t)(((preds[mn] - 1) % mod) + 2);
        preds[5] = (int8_t)(((preds[mx] + offset) % mod) + 2);
      }
    } else if (left_dir + above_dir >= 2) {
      preds[0] = 0;
      preds[1] = (int8_t)(left_dir < above_dir ? above_dir : left_dir);
      preds[2] = (int8_t)(((preds[1] + offset) % mod) + 2); preds[3] = (int8_t)(((preds[1] - 1) % mod) + 2);
      preds[4] = (int8_t)(((preds[1] + offset - 1) % mod) + 2); preds[5] = (int8_t)((preds[1] % mod) + 2);
    }
  }
  return n;
}

/* the MPM neighbours as uvg_search_cu_intra (search_intra.c:1792-1803) and uvg_encode_intra_luma_coding_unit (:1114-1150) pick them */
static void mpm_neighbours(const s_loc *loc, const s_lcu *lcu, const s_cu **left, const s_cu **above)
{
  *left = *above = NULL;
  if (loc->x > 0) *left = CU_AT(lcu, loc->lx - 1, (loc->y + loc->h - 1) % LCU);
  if (loc->y % LCU > 0 && loc->y > 0) *above = CU_AT(lcu, (loc->x + loc->w - 1) % LCU, loc->ly - 1);
}

/* uvg_encode_intra_luma_coding_unit (encode_coding_tree.c:992-1238) in count mode: MIP / MRL / ISP off */
static void encode_intra_luma(s_cabac *cb, int mode, const s_loc *loc, const s_cu *left_pu, const s_cu *above_pu, double *bits_out)
{
  int8_t preds[6];
  double bits = 0;
  get_dir_luma_predictor(loc->y, preds, left_pu, above_pu);
  int mpm = -1;
  for (int i = 0; i < 6; ++i) if (preds[i] == mode) { mpm = i; break; }
  fbits_update(cb, M_MPM, mpm != -1, &bits);
  if (mpm != -1) {
    fbits_update(cb, M_PLANAR + 1, mpm > 0, &bits);
    if (mpm > 0) { bits += 1; ORC_FN(cabac_sim_ep)(mpm > 1); }          /* mpm_idx: one bypass bin each (encode_coding_tree.c:1174-1189) */
    if (mpm > 1) { bits += 1; ORC_FN(cabac_sim_ep)(mpm > 2); }
    if (mpm > 2) { bits += 1; ORC_FN(cabac_sim_ep)(mpm > 3); }
    if (mpm > 3) { bits += 1; ORC_FN(cabac_sim_ep)(mpm > 4); }
  } else {
    /* sort the candidates, remove them from the mode's index, truncated binary code of 61 symbols (cabac.c:203-229) */
    int8_t sorted[6];
    memcpy(sorted, preds, 6);
    for (int i = 0; i < 6; ++i)
      for (int j = i + 1; j < 6; ++j)
        if ((uint8_t)sorted[j] < (uint8_t)sorted[i]) { int8_t t = sorted[i]; sorted[i] = sorted[j]; sorted[j] = t; }
    int tmp = mode;
    for (int i = 5; i >= 0; --i) if (tmp > sorted[i]) tmp--;
    if (bits_out) *bits_out += (tmp < 3) ? 5 : 6;       /* uvg_cabac_encode_trunc_bin adds to bits_out directly */
    if (tmp < 3) ORC_FN(cabac_sim_eps)((uint32_t)tmp, 5);        /* 61 symbols: 2^5 = 32, 29 above: the first 3 take 5 bits (cabac.c:203-229) */
    else ORC_FN(cabac_sim_eps)((uint32_t)tmp + 3, 6);
  }
  if (bits_out) *bits_out += bits;
}

/* encode_chroma_intra_cu (encode_coding_tree.c:902-990), CCLM off */
static void encode_chroma_intra(s_cabac *cb, int chroma_mode, int luma_dir, double *bits_out)
{
  double bits = 0;
  const int derived = chroma_mode == luma_dir;
  fbits_update(cb, M_CHROMA_PRED, derived ? 0 : 1, &bits);
  if (!derived) {
    bits += 2;
    int modes[4] = {0, 50, 18, 1}, idx = 0;              /* :909-915, 943-947: the entry equal to the luma mode stands for 66 */
    for (int i = 0; i < 4; ++i) if (modes[i] == luma_dir) modes[i] = 66;
    while (idx < 4 && modes[idx] != chroma_mode) ++idx;
    ORC_FN(cabac_sim_eps)((uint32_t)idx, 2);
  }
  if (bits_out) *bits_out += bits;
}

/* intra_predict_regular + DC/planar/angular/PDPC (intra.c:660-753) through orc_intra.c */
/* fw x fh: the size the smoothing filter covers -- the CU's, not the transform block's (intra.c:715-725: a 32x32 block of a 64x64 CU
   has its entry 2N smoothed too, against the padding behind it) */
static void predict_in_cu(int mode, int color, int w, int h, int fw, int fh, const orc_px *top, const orc_px *left, orc_px *dst)
{
  orc_px ftop[REF_LEN], fleft[REF_LEN];
  ORC_FN(intra_filter_refs)(top, left, fw, fh, ftop, fleft);
  ORC_FN(intra_predict)(mode, color != 0, w, h, top, left, ftop, fleft, dst);
}
static void predict(int mode, int color, int w, int h, const orc_px *top, const orc_px *left, orc_px *dst)
{
  orc_px ftop[REF_LEN], fleft[REF_LEN];
  ORC_FN(intra_filter_refs)(top, left, w, h, ftop, fleft);
  ORC_FN(intra_predict)(mode, color != 0, w, h, top, left, ftop, fleft, dst);
}

/* count_bits (search_intra.c:949-984) */
static double count_bits(const int8_t *preds, double planar, double not_planar, double mpm_bit, double not_mpm_bit, int mode)
{
  int i = 0, smaller = 0;
  double bits;
  for (; i < 6; i++) {
    if (preds[i] == mode) break;
    if (mode > preds[i]) smaller += 1;
  }
  if (i == 0) bits = planar + mpm_bit;
  else if (i < 6) bits = not_planar + mpm_bit + (i < 4 ? i : 4);
  else bits = not_mpm_bit + 5 + (mode - smaller > 2);
  return bits;       /* + not_mrl + not_mip + not_isp_flag, all 0 */
}

static double cost_of(const orc_px *pred, const orc_px *orig, int n)   /* get_cost_dual, one block (search_intra.c:133-192) */
{
  const unsigned satd = ORC_FN(satd_nxn)(pred, orig, n), sad = ORC_FN(sad_nxn)(pred, orig, n);
  return (double)(satd < sad * 2 ? satd : sad * 2);
}

/* search_intra_rough (search_intra.c:986-1229) for mode_list_size = 3; returns the best mode */
static int search_intra_rough(const s_state *st, const s_loc *loc, const s_lcu *lcu, const orc_px *top, const orc_px *left, const int8_t *intra_preds,
                              double *best_cost_out)
{
  const int n = loc->w;
  orc_px orig[32 * 32], pred[32 * 32];
  for (int y = 0; y < n; ++y) memcpy(&orig[y * n], &lcu->ref_y[(loc->ly + y) * LCU + loc->lx], (size_t)n * sizeof(orc_px));
  const orc_models *m = &st->search.m;
  const double mpm_bit = ctx_fbits(m, M_MPM, 1), not_mpm_bit = ctx_fbits(m, M_MPM, 0);
  const double planar = ctx_fbits(m, M_PLANAR + 1, 0), not_planar = ctx_fbits(m, M_PLANAR + 1, 1);
  const double lsq = st->p->lambda_sqrt;
  double costs[67];
  int checked[67] = {0};
  struct { int mode; double cost; } best[6], tmp_best[6];
  const int list = 3;
  int offset = 1 << st->p->rough_levels;
#define EVAL(md) (predict((md), 0, n, n, top, left, pred), cost_of(pred, orig, n) + count_bits(intra_preds, planar, not_planar, mpm_bit, not_mpm_bit, (md)) * lsq)
  costs[0] = EVAL(0);
  costs[1] = EVAL(1);
  checked[0] = checked[1] = 1;
  double min_cost, max_cost;
  if (costs[0] < costs[1]) { min_cost = costs[0]; max_cost = costs[1]; best[0].mode = 0; best[0].cost = costs[0]; best[1].mode = 1; best[1].cost = costs[1]; }
  else { min_cost = costs[1]; max_cost = costs[0]; best[1].mode = 0; best[1].cost = costs[0]; best[0].mode = 1; best[0].cost = costs[1]; }
  for (int i = 2; i < 6; ++i) { best[i].cost = 1.7e308; best[i].mode = 0; }
  best[2].cost = best[3].cost = best[4].cost = best[5].cost = 1.7976931348623157e308;
  for (int mode = 2 + offset / 2; mode <= 66; mode += 2 * offset) {
    for (int i = 0; i < 2; ++i) {
      const int mi = mode + i * offset;
      if (mi > 66) continue;
      costs[mi] = EVAL(mi);
      checked[mi] = 1;
      if (costs[mi] < min_cost) min_cost = costs[mi];
      if (costs[mi] > max_cost) max_cost = costs[mi];
      for (int j = 0; j < list; j++) {
        if (costs[mi] < best[j].cost) {
          for (int k = list - 1; k > j; k--) best[k] = best[k - 1];
          best[j].cost = costs[mi]; best[j].mode = mi;
          break;
        }
      }
    }
  }
  offset >>= 1;
  if (min_cost != max_cost) {
    for (; offset > 0; offset >>= 1) {
      memcpy(tmp_best, best, sizeof tmp_best);
      int to_check[12], n_check = 0;
      for (int i = 0; i < list; i++) {
        const int center = best[i].mode;
        if (offset != 0 && (center < 3 || center > 65)) continue;
        const int test[2] = {center - offset, center + offset};
        for (int j = 0; j < 2; j++)
          if (test[j] >= 2 && test[j] <= 66 && !checked[test[j]]) { to_check[n_check++] = test[j]; checked[test[j]] = 1; }
      }
      /* (the reference pads the list with DC to a multiple of two and ignores those results) */
      for (int i = 0; i < n_check; ++i) {
        const int mode = to_check[i];
        costs[mode] = EVAL(mode);
        for (int j = 0; j < list; j++) {
          if (costs[mode] < best[j].cost) {
            for (int k = list - 1; k > j; k--) best[k] = best[k - 1];
            best[j].cost = costs[mode]; best[j].mode = mode;
            break;
          }
        }
      }
    }
  }
#undef EVAL
  *best_cost_out = best[0].cost;
  return best[0].mode;
}

/* --------------------------------------------------------------------------------------------------------- reconstruction -- */
static int scaled_qp(const s_state *st, int color)
{
  return (color == 0 ? st->p->qp : st->p->qp_c) + 6 * (ORC_BIT_DEPTH - 8);      /* uvg_get_scaled_qp, transform.c:150-165 */
}

/* quantize_tr_residual -> uvg_quantize_residual (transform.c:1283-1480, quant-generic.c:460-612): the prediction is in rec */
static void quantize_tr_residual(s_state *st, int color, const s_loc *loc, s_cu *cur_pu, s_lcu *lcu, int early_skip)
{
  const int shift = color == 0 ? 0 : 1;
  const int lx = loc->lx >> shift, ly = loc->ly >> shift;
  const int w = color == 0 ? loc->w : loc->cw, h = color == 0 ? loc->h : loc->ch;
  const int lw = LCU >> shift, off = lx + ly * lw;
  orc_px *pred = (color == 0 ? lcu->rec_y : color == 1 ? lcu->rec_u : lcu->rec_v) + off;
  const orc_px *ref = (color == 0 ? lcu->ref_y : color == 1 ? lcu->ref_u : lcu->ref_v) + off;
  int16_t *dst_coeff = (color == 0 ? lcu->coeff_y : color == 1 ? lcu->coeff_u : lcu->coeff_v) + off;
  int16_t residual[32 * 32], coeff[32 * 32], q[32 * 32];
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) residual[y * w + x] = (int16_t)((int)ref[y * lw + x] - (int)pred[y * lw + x]);
  ORC_FN(dct_nxn)(ORC_BIT_DEPTH, w, residual, coeff);
  memset(q, 0, sizeof q);
  const double lambda = color == 0 ? st->p->lambda : st->c_lambda;
  ORC_FN(rdoq)(coeff, q, w, h, color, cur_pu->type, (cur_pu->cbf >> 1) & 1, 0, 0, scaled_qp(st, color), lambda, &st->rdoq);
  int has_coeffs = 0;
  for (int i = 0; i < w * h; ++i) if (q[i]) { has_coeffs = 1; break; }
  if (has_coeffs && !early_skip) {
    ORC_FN(dequant)(q, coeff, w, h, ORC_BIT_DEPTH, scaled_qp(st, color), 0);
    ORC_FN(idct_nxn)(ORC_BIT_DEPTH, w, coeff, residual);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int16_t val = (int16_t)(residual[y * w + x] + pred[y * lw + x]);
        pred[y * lw + x] = orc_clip_px(val);
      }
  }
  cur_pu->cbf &= (uint8_t)~(1 << color);
  if (has_coeffs) {
    for (int y = 0; y < h; ++y) memcpy(&dst_coeff[y * lw], &q[y * w], (size_t)w * sizeof(int16_t));
    cur_pu->cbf |= (uint8_t)(1 << color);
  } else {
    for (int y = 0; y < h; ++y) memset(&dst_coeff[y * lw], 0, (size_t)w * sizeof(int16_t));
  }
}

/* intra_recon_tb_leaf (intra.c:1537-1614) */
static void recon_tb_leaf(s_state *st, const s_loc *loc, s_lcu *lcu, int color, int mode, const s_cu *cu)
{
  orc_px top[REF_LEN], left[REF_LEN], pred[32 * 32];
  const int shift = color == 0 ? 0 : 1;
  const int w = color == 0 ? loc->w : loc->cw, h = color == 0 ? loc->h : loc->ch;
  const int lw = LCU >> shift;
  build_reference(st, loc, color, lcu, top, left);
  if (color == 0) predict_in_cu(mode, color, w, h, 1 << cu->log2_w, 1 << cu->log2_h, top, left, pred);
  else predict(mode, color, w, h, top, left, pred);
  orc_px *block = (color == 0 ? lcu->rec_y : color == 1 ? lcu->rec_u : lcu->rec_v) + (loc->lx >> shift) + (loc->ly >> shift) * lw;
  for (int y = 0; y < h; ++y) memcpy(&block[y * lw], &pred[y * w], (size_t)w * sizeof(orc_px));
}

/* uvg_intra_recon_cu (intra.c:1632-1745) + uvg_quantize_lcu_residual (transform.c:1487-1603).  cur_cu == NULL: the CTU's entry. */
static void intra_recon_cu(s_state *st, int mode, int mode_chroma, const s_loc *loc, s_cu *cur_cu, s_lcu *lcu, int recon_luma, int recon_chroma)
{
  if (cur_cu == NULL) cur_cu = CU_AT(lcu, loc->lx, loc->ly);
  if (recon_luma) cur_cu->cbf &= (uint8_t)~1;
  if (recon_chroma) cur_cu->cbf &= (uint8_t)~6;
  if (loc->w > 32 || loc->h > 32) {
    const int hw = loc->w >> 1, hh = loc->h >> 1;
    for (int i = 0; i < 4; ++i) {
      s_loc sl;
      loc_ctor(&sl, loc->x + (i & 1) * hw, loc->y + (i >> 1) * hh, hw, hh);
      intra_recon_cu(st, mode, mode_chroma, &sl, NULL, lcu, recon_luma, recon_chroma);
    }
    return;
  }
  if (recon_luma) recon_tb_leaf(st, loc, lcu, 0, mode, cur_cu);
  if (recon_chroma) { recon_tb_leaf(st, loc, lcu, 1, mode_chroma, cur_cu); recon_tb_leaf(st, loc, lcu, 2, mode_chroma, cur_cu); }
  /* uvg_quantize_lcu_residual */
  if (recon_luma) cur_cu->cbf &= (uint8_t)~1;
  if (recon_chroma) cur_cu->cbf &= (uint8_t)~6;
  if (recon_luma) quantize_tr_residual(st, 0, loc, cur_cu, lcu, 0);
  const double c_lambda = st->c_lambda;
  {   /* uvg_calculate_chroma_lambda (rate_control.c:1216-1233), no dep-quant, no JCCR */
    double lambda = st->p->lambda;
    const double w = pow(2.0, (st->p->qp - st->p->qp_c) / 3.0);
    lambda /= w;
    lambda *= 1.0;
    st->c_lambda = lambda;
  }
  if (recon_chroma) {
    /* handled_elsewhere (transform.c:1301): a chroma block of a luma CU narrower than 8 that is not on the 8x8 grid */
    quantize_tr_residual(st, 1, loc, cur_cu, lcu, 0);
    quantize_tr_residual(st, 2, loc, cur_cu, lcu, 0);
  }
  st->c_lambda = c_lambda;
}

/* uvg_quantize_lcu_residual (transform.c:1487-1603) as the inter path calls it: the prediction is in rec; cur_pu == NULL: the entry of the
 * lcu_t.  Blocks wider than 32 are four transform units, each with the flags of its own entry; the first one's root_cbf collects them. */
static void quantize_lcu_residual(s_state *st, int luma, int chroma, const s_loc *loc, s_cu *cur_pu, s_lcu *lcu, int early_skip)
{
  if (cur_pu == NULL) cur_pu = CU_AT(lcu, loc->lx, loc->ly);
  if (luma) cur_pu->cbf &= (uint8_t)~1;
  if (chroma) cur_pu->cbf &= (uint8_t)~6;
  if (loc->w > 32 || loc->h > 32) {
    const int hw = loc->w >> 1, hh = loc->h >> 1;
    uint8_t child_cbfs[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      s_loc sl;
      loc_ctor(&sl, loc->x + (i & 1) * hw, loc->y + (i >> 1) * hh, hw, hh);
      quantize_lcu_residual(st, luma, chroma, &sl, NULL, lcu, early_skip);
      if (i != 0) child_cbfs[i - 1] = CU_AT(lcu, sl.lx, sl.ly)->cbf;
    }
    cur_pu->root_cbf = (cur_pu->cbf & 7) || (child_cbfs[0] & 7) || (child_cbfs[1] & 7) || (child_cbfs[2] & 7);
    return;
  }
  if (luma) quantize_tr_residual(st, 0, loc, cur_pu, lcu, early_skip);
  const double c_lambda = st->c_lambda;
  st->c_lambda = st->p->lambda / pow(2.0, (st->p->qp - st->p->qp_c) / 3.0) * 1.0;     /* uvg_calculate_chroma_lambda, no JCCR */
  if (chroma) {
    quantize_tr_residual(st, 1, loc, cur_pu, lcu, early_skip);
    quantize_tr_residual(st, 2, loc, cur_pu, lcu, early_skip);
  }
  st->c_lambda = c_lambda;
}

#define MAX_DOUBLE 1.7976931348623157e308
#include "orc_search_inter.inc"

/* ------------------------------------------------------------------------------------------------------------- RD costs -- */
/* uvg_cu_rd_cost_chroma (search.c:625-722) */
static double cu_rd_cost_chroma(s_state *st, const s_cu *pred_cu, s_lcu *lcu, const s_loc *loc)
{
  const int px = loc->lx / 2, py = loc->ly / 2;
  double tr_tree_bits = 0, coeff_bits = 0;
  const int u_is_set = (pred_cu->cbf >> 1) & 1, v_is_set = (pred_cu->cbf >> 2) & 1;
  s_cabac *cb = &st->search;
  fbits_update(cb, M_CBF_CB + 0, u_is_set, &tr_tree_bits);
  fbits_update(cb, M_CBF_CR + u_is_set, v_is_set, &tr_tree_bits);
  const int index = py * LCU_C + px;
  const int ssd_u = (int)ORC_FN(pixels_calc_ssd)(&lcu->ref_u[index], &lcu->rec_u[index], LCU_C, LCU_C, loc->cw, loc->ch);
  const int ssd_v = (int)ORC_FN(pixels_calc_ssd)(&lcu->ref_v[index], &lcu->rec_v[index], LCU_C, LCU_C, loc->cw, loc->ch);
  const int ssd = ssd_u + ssd_v;
  coeff_bits += coeff_cost_cu(cb, lcu->coeff_u, LCU_C, px, py, loc->cw, loc->ch, 1);
  coeff_bits += coeff_cost_cu(cb, lcu->coeff_v, LCU_C, px, py, loc->cw, loc->ch, 2);
  const double bits = tr_tree_bits + coeff_bits;
  return (double)ssd * 1.0 + bits * st->c_lambda;
}

/* cu_rd_cost_tr_split_accurate (search.c:724-986) */
static double cu_rd_cost_tr_split_accurate(s_state *st, const s_cu *pred_cu, s_lcu *lcu, const s_loc *loc, const s_loc *chroma_loc, int has_chroma)
{
  s_cu *tr_cu = CU_AT(lcu, loc->lx, loc->ly);
  double coeff_bits = 0, luma_bits = 0, chroma_bits = 0;
  const int cb_flag_u = (tr_cu->cbf >> 1) & 1, cb_flag_v = (tr_cu->cbf >> 2) & 1;
  const int skip_residual_coding = pred_cu->skipped || (pred_cu->type != CU_INTRA && pred_cu->cbf == 0);
  s_cabac *cb = &st->search;
  if (pred_cu->type != CU_INTRA && !pred_cu->merged) fbits_update(cb, M_ROOT_CBF, (tr_cu->cbf & 7) != 0, &luma_bits);      /* at every level of the recursion (:744-751) */
  if (loc->w > 32 || loc->h > 32) {
    double sum = 0;
    const int hw = loc->w >> 1, hh = loc->h >> 1;
    for (int i = 0; i < 4; ++i) {
      s_loc sl, scl;
      loc_ctor(&sl, loc->x + (i & 1) * hw, loc->y + (i >> 1) * hh, hw, hh);
      if (chroma_loc) loc_ctor(&scl, chroma_loc->x + (i & 1) * hw, chroma_loc->y + (i >> 1) * hh, hw, hh);
      sum += cu_rd_cost_tr_split_accurate(st, pred_cu, lcu, &sl, chroma_loc ? &scl : NULL, has_chroma);
    }
    return sum + luma_bits * st->p->lambda;
  }
  if (!skip_residual_coding && has_chroma) {
    fbits_update(cb, M_CBF_CB + 0, cb_flag_u, &chroma_bits);
    fbits_update(cb, M_CBF_CR + cb_flag_u, cb_flag_v, &chroma_bits);
  }
  const int cb_flag_y = tr_cu->cbf & 1;
  if ((pred_cu->type == CU_INTRA || cb_flag_u || cb_flag_v) && !skip_residual_coding) fbits_update(cb, M_CBF_LUMA + 0, cb_flag_y, &luma_bits);
  const int index = loc->lx + LCU * loc->ly;
  const unsigned luma_ssd = ORC_FN(pixels_calc_ssd)(&lcu->ref_y[index], &lcu->rec_y[index], LCU, LCU, loc->w, loc->h);
  if (cb_flag_y) coeff_bits += coeff_cost_cu(cb, lcu->coeff_y, LCU, loc->lx, loc->ly, loc->w, loc->h, 0);
  unsigned chroma_ssd = 0;
  if (has_chroma) {
    const int px = chroma_loc->lx >> 1, py = chroma_loc->ly >> 1;
    const int cidx = py * LCU_C + px;
    const unsigned ssd_u = (unsigned)(ORC_FN(pixels_calc_ssd)(&lcu->ref_u[cidx], &lcu->rec_u[cidx], LCU_C, LCU_C, chroma_loc->cw, chroma_loc->ch) * st->p->cw_u);
    const unsigned ssd_v = (unsigned)(ORC_FN(pixels_calc_ssd)(&lcu->ref_v[cidx], &lcu->rec_v[cidx], LCU_C, LCU_C, chroma_loc->cw, chroma_loc->ch) * st->p->cw_v);
    chroma_ssd = ssd_u + ssd_v;
    chroma_bits += coeff_cost_cu(cb, lcu->coeff_u, LCU_C, px, py, chroma_loc->cw, chroma_loc->ch, 1);
    chroma_bits += coeff_cost_cu(cb, lcu->coeff_v, LCU_C, px, py, chroma_loc->cw, chroma_loc->ch, 2);
  }
  const double bits = luma_bits + coeff_bits;
  return luma_ssd * 1.0 + chroma_ssd * 1.0 + (bits + chroma_bits) * st->p->lambda;
}

/* uvg_mock_encode_coding_unit (encode_coding_tree.c:1730-1862), I slice */
static double mock_encode_coding_unit(s_state *st, s_cabac *cb, const s_loc *loc, const s_loc *chroma_loc, s_lcu *lcu, const s_cu *cur_cu, s_tree tree)
{
  double bits = 0;
  const s_cu *left_cu = NULL, *above_cu = NULL;
  if (loc->x) left_cu = CU_AT(lcu, loc->lx - 1, loc->ly);
  if (loc->y) above_cu = CU_AT(lcu, loc->lx, loc->ly - 1);
  if (cur_cu->log2_h + cur_cu->log2_w > 4) {
    int is_implicit;
    s_tree t = tree;                                    /* the CU itself is not split at this depth: the flag's value is NO_SPLIT */
    write_split_flag(st, cb, left_cu, above_cu, loc, t, &is_implicit, &bits);
  }
  const int non_i = st->fr && st->fr->slice_type != 2;
  if (non_i && (loc->w != 4 || loc->h != 4)) {           /* skip flag (:1788-1823) */
    int ctx_skip = 0;
    if (left_cu && left_cu->skipped) ctx_skip++;
    if (above_cu && above_cu->skipped) ctx_skip++;
    fbits_update(cb, M_SKIP + ctx_skip, cur_cu->skipped, &bits);
    if (cur_cu->skipped) {
      const int num_cand = st->fr->max_merge;
      if (num_cand > 1)
        for (int ui = 0; ui < num_cand - 1; ui++) {
          const int symbol = ui != cur_cu->merge_idx;
          if (ui == 0) fbits_update(cb, M_MERGE_IDX, symbol, &bits);
          else { ORC_FN(cabac_sim_ep)((uint32_t)symbol); bits += 1; }
          if (symbol == 0) break;
        }
      return bits;
    }
  }
  if (non_i && (loc->w != 4 || loc->h != 4)) {           /* prediction mode (:1824-1834) */
    const int ctx_predmode = (left_cu && left_cu->type == CU_INTRA) || (above_cu && above_cu->type == CU_INTRA);
    fbits_update(cb, M_PRED_MODE + ctx_predmode, cur_cu->type == CU_INTRA, &bits);
  }
  if (cur_cu->type == CU_INTER) {
    encode_inter_prediction_unit(st, cb, cur_cu, lcu, &bits, loc);       /* (amvr off: no imv flag) */
    return bits;
  }
  const s_cu *left_pu, *above_pu;
  mpm_neighbours(loc, lcu, &left_pu, &above_pu);
  encode_intra_luma(cb, cur_cu->mode, loc, left_pu, above_pu, &bits);
  if (chroma_loc) {
    /* uvg_get_co_located_luma_mode (intra.c:1423-1454; UVG_CHROMA_T for a separate tree: the CENTRE of the chroma area): this CU
     * whenever it carries chroma */
    const int luma_dir = cur_cu->mode;
    encode_chroma_intra(cb, cur_cu->mode_chroma, luma_dir, &bits);
  }
  return bits;
}

/* ----------------------------------------------------------------------------------------------------- work tree helpers -- */
static void lcu_fill_cu_info(s_lcu *lcu, int lx, int ly, int w, int h, const s_cu *cu)      /* search.c:314-353 */
{
  for (int y = ly; y < ly + h; y += 4)
    for (int x = lx; x < lx + w; x += 4) {
      s_cu *to = CU_AT(lcu, x, y);
      to->type = cu->type; to->qp = cu->qp; to->split_tree = cu->split_tree; to->mode_type_tree = cu->mode_type_tree;
      to->log2_h = cu->log2_h; to->log2_w = cu->log2_w; to->log2_ch = cu->log2_ch; to->log2_cw = cu->log2_cw;
      if (cu->type == CU_INTRA) { to->mode = cu->mode; to->mode_chroma = cu->mode_chroma; }
      else {
        to->skipped = cu->skipped; to->merged = cu->merged; to->merge_idx = cu->merge_idx;
        to->mv_dir = cu->mv_dir; to->mv_cand0 = cu->mv_cand0; to->mv_cand1 = cu->mv_cand1; to->mv_ref[0] = cu->mv_ref[0]; to->mv_ref[1] = cu->mv_ref[1];
        memcpy(to->mv, cu->mv, sizeof to->mv);
      }
    }
}
static void lcu_fill_cbf(s_lcu *lcu, int lx, int ly, int w, int h)                           /* search.c:402-420, single tree */
{
  for (int y = 0; y < h; y += 4)
    for (int x = 0; x < w; x += 4) {
      const s_cu *from = CU_AT(lcu, lx + (x & ~31), ly + (y & ~31));
      s_cu *to = CU_AT(lcu, lx + x, ly + y);
      if (from != to) to->cbf = (uint8_t)((to->cbf & ~7) | (from->cbf & 7));
    }
}
static void lcu_fill_chroma_cu_info(s_lcu *lcu, const s_loc *loc)                           /* search.c:355-378 */
{
  const s_cu *br = CU_AT(lcu, loc->lx + loc->w - 1, loc->ly + loc->h - 1);
  if (br->type != CU_INTRA) return;
  for (int y = loc->ly; y < loc->ly + loc->h; y += 4)
    for (int x = loc->lx; x < loc->lx + loc->w; x += 4) {
      s_cu *cu = CU_AT(lcu, x, y);
      cu->mode_chroma = br->mode_chroma; cu->log2_ch = br->log2_ch; cu->log2_cw = br->log2_cw; cu->type = br->type;
    }
}
static void lcu_fill_chroma_cbfs(s_lcu *lcu, const s_loc *chroma_loc)                       /* search.c:380-400 */
{
  const int offset = ~31;
  for (int y = 0; y < chroma_loc->h; y += 4)
    for (int x = 0; x < chroma_loc->w; x += 4) {
      const s_cu *from = CU_AT(lcu, chroma_loc->lx + (x & offset), chroma_loc->ly + (y & offset));
      s_cu *to = CU_AT(lcu, chroma_loc->lx + x, chroma_loc->ly + y);
      if (from != to) to->cbf = (uint8_t)((to->cbf & ~6) | (from->cbf & 6));
    }
}

static void blit_px(const orc_px *src, orc_px *dst, int w, int h, int ss, int ds)
{
  for (int y = 0; y < h; ++y) memcpy(&dst[y * ds], &src[y * ss], (size_t)w * sizeof(orc_px));
}

/* initialize_partial_work_tree (search.c:86-222), single tree.  The parts of `to` the reference leaves uninitialised (malloc) are
 * zero here; nothing reads them. */
static void initialize_partial_work_tree(const s_state *st, const s_lcu *from, s_lcu *to, const s_loc *loc, const s_loc *chroma_loc)
{
  const int y_limit = LCU < st->p->pic_h - loc->y / 64 * 64 ? LCU : st->p->pic_h - loc->y / 64 * 64;
  const int x_limit = LCU < st->p->pic_w - loc->x / 64 * 64 ? LCU : st->p->pic_w - loc->x / 64 * 64;
  if (loc->lx == 0) {
    memcpy(to->left_y, from->left_y, sizeof to->left_y); memcpy(to->left_u, from->left_u, sizeof to->left_u); memcpy(to->left_v, from->left_v, sizeof to->left_v);
    to->cu[TCW * TCW] = from->cu[TCW * TCW];
  } else {
    blit_px(from->rec_y, to->rec_y, loc->lx, LCU, LCU, LCU);
    blit_px(from->rec_u, to->rec_u, chroma_loc->lx / 2, LCU_C, LCU_C, LCU_C);
    blit_px(from->rec_v, to->rec_v, chroma_loc->lx / 2, LCU_C, LCU_C, LCU_C);
  }
  if (loc->ly == 0) {
    memcpy(to->top_y, from->top_y, sizeof to->top_y); memcpy(to->top_u, from->top_u, sizeof to->top_u); memcpy(to->top_v, from->top_v, sizeof to->top_v);
    to->cu[TCW * TCW] = from->cu[TCW * TCW];
  } else {
    blit_px(&from->rec_y[loc->lx], &to->rec_y[loc->lx], LCU - loc->lx, loc->ly, LCU, LCU);
    blit_px(&from->rec_u[chroma_loc->lx / 2], &to->rec_u[chroma_loc->lx / 2], LCU_C - chroma_loc->lx / 2, chroma_loc->ly / 2, LCU_C, LCU_C);
    blit_px(&from->rec_v[chroma_loc->lx / 2], &to->rec_v[chroma_loc->lx / 2], LCU_C - chroma_loc->lx / 2, chroma_loc->ly / 2, LCU_C, LCU_C);
  }
  {
    const int off = loc->lx + loc->ly * LCU;
    blit_px(&from->ref_y[off], &to->ref_y[off], loc->w, loc->h, LCU, LCU);
    const int coff = chroma_loc->lx / 2 + chroma_loc->ly / 2 * LCU_C;
    blit_px(&from->ref_u[coff], &to->ref_u[coff], chroma_loc->cw, chroma_loc->ch, LCU_C, LCU_C);
    blit_px(&from->ref_v[coff], &to->ref_v[coff], chroma_loc->cw, chroma_loc->ch, LCU_C, LCU_C);
  }
  const int y_start = loc->ly - 4, x_start = loc->lx - 4;
  for (int y = y_start; y < y_limit; y += 4) *CU_AT(to, x_start, y) = *CU_AT(from, x_start, y);
  for (int x = x_start; x < x_limit; x += 4) *CU_AT(to, x, y_start) = *CU_AT(from, x, y_start);
  for (int y = loc->ly; y < y_limit; y += 4)
    for (int x = loc->lx; x < x_limit; x += 4) memset(CU_AT(to, x, y), 0, sizeof(s_cu));
  /* (chroma_loc has the CU's own origin on this path: the branch of search.c:150-158 / 174-208 for a detached chroma area is never taken) */
  if (x_limit != LCU) for (int y = y_start; y < y_limit; y += 4) memset(CU_AT(to, x_limit, y), 0, sizeof(s_cu));
  if (y_limit != LCU) for (int x = x_start; x < x_limit; x += 4) memset(CU_AT(to, x, y_limit), 0, sizeof(s_cu));
}

/* work_tree_copy_up (search.c:278-311), single tree, no JCCR */
static void work_tree_copy_up(const s_lcu *from, s_lcu *to, const s_loc *loc, const s_loc *chroma_loc)
{
  for (int y = loc->ly; y < loc->ly + loc->h; y += 4)
    for (int x = loc->lx; x < loc->lx + loc->w; x += 4) *CU_AT(to, x, y) = *CU_AT(from, x, y);
  const int li = loc->lx + loc->ly * LCU, ci = (loc->lx / 2) + (loc->ly / 2) * LCU_C;
  blit_px(&from->rec_y[li], &to->rec_y[li], loc->w, loc->h, LCU, LCU);
  blit_px(&from->rec_u[ci], &to->rec_u[ci], loc->cw, loc->ch, LCU_C, LCU_C);
  blit_px(&from->rec_v[ci], &to->rec_v[ci], loc->cw, loc->ch, LCU_C, LCU_C);
  for (int y = 0; y < loc->h; ++y) memcpy(&to->coeff_y[li + y * LCU], &from->coeff_y[li + y * LCU], (size_t)loc->w * 2);
  for (int y = 0; y < loc->ch; ++y) {
    memcpy(&to->coeff_u[ci + y * LCU_C], &from->coeff_u[ci + y * LCU_C], (size_t)loc->cw * 2);
    memcpy(&to->coeff_v[ci + y * LCU_C], &from->coeff_v[ci + y * LCU_C], (size_t)loc->cw * 2);
  }
  if (chroma_loc) {
    const int cci = (chroma_loc->lx / 2) + (chroma_loc->ly / 2) * LCU_C;
    blit_px(&from->rec_u[cci], &to->rec_u[cci], chroma_loc->cw, chroma_loc->ch, LCU_C, LCU_C);
    blit_px(&from->rec_v[cci], &to->rec_v[cci], chroma_loc->cw, chroma_loc->ch, LCU_C, LCU_C);
    for (int y = 0; y < chroma_loc->ch; ++y) {
      memcpy(&to->coeff_u[cci + y * LCU_C], &from->coeff_u[cci + y * LCU_C], (size_t)chroma_loc->cw * 2);
      memcpy(&to->coeff_v[cci + y * LCU_C], &from->coeff_v[cci + y * LCU_C], (size_t)chroma_loc->cw * 2);
    }
    for (int y = chroma_loc->ly; y < chroma_loc->ly + chroma_loc->h; y += 4)
      for (int x = chroma_loc->lx; x < chroma_loc->lx + chroma_loc->w; x += 4) {
        s_cu *t = CU_AT(to, x, y);
        const s_cu *f = CU_AT(from, x, y);
        t->mode_chroma = f->mode_chroma; t->chroma_deblocking = f->chroma_deblocking; t->log2_cw = f->log2_cw; t->log2_ch = f->log2_ch;
        t->cbf = (uint8_t)((t->cbf & ~6) | (f->cbf & 6));
      }
  }
}

/* mark_deblocking (search.c:1075-1174), single tree, not skipped */
static void mark_deblocking(const s_loc *loc, const s_loc *chroma_loc, s_lcu *lcu, int has_chroma, int is_separate_tree, int is_skip)
{
  if (loc->x) {
    for (int x = loc->lx; x < loc->lx + loc->w; x += 32) {
      for (int y = loc->ly; y < loc->ly + loc->h; y += 4) {
        CU_AT(lcu, x, y)->luma_deblocking |= EDGE_VER;
        if (!is_separate_tree) CU_AT(lcu, x, y)->chroma_deblocking |= EDGE_VER;
      }
      if (is_skip) break;
    }
  } else if (loc->w == 64 && !is_skip) {
    for (int y = loc->ly; y < loc->ly + loc->h; y += 4) {
      CU_AT(lcu, 32, y)->luma_deblocking |= EDGE_VER;
      if (!is_separate_tree) CU_AT(lcu, 32, y)->chroma_deblocking |= EDGE_VER;
    }
  }
  if (loc->y) {
    for (int y = loc->ly; y < loc->ly + loc->h; y += 32) {
      for (int x = loc->lx; x < loc->lx + loc->w; x += 4) {
        CU_AT(lcu, x, y)->luma_deblocking |= EDGE_HOR;
        if (!is_separate_tree) CU_AT(lcu, x, y)->chroma_deblocking |= EDGE_HOR;
      }
      if (is_skip) break;
    }
  } else if (loc->h == 64 && !is_skip) {
    for (int x = loc->lx; x < loc->lx + loc->w; x += 4) {
      CU_AT(lcu, x, 32)->luma_deblocking |= EDGE_HOR;
      if (!is_separate_tree) CU_AT(lcu, x, 32)->chroma_deblocking |= EDGE_HOR;
    }
  }
  if (is_separate_tree && has_chroma) {
    if (chroma_loc->x) {
      for (int x = chroma_loc->lx; x < chroma_loc->lx + chroma_loc->w; x += 32)
        for (int y = chroma_loc->ly; y < chroma_loc->ly + chroma_loc->h; y += 4) CU_AT(lcu, x, y)->chroma_deblocking |= EDGE_VER;
    }
    if (chroma_loc->y) {
      for (int y = chroma_loc->ly; y < chroma_loc->ly + chroma_loc->h; y += 32)
        for (int x = chroma_loc->lx; x < chroma_loc->lx + chroma_loc->w; x += 4) CU_AT(lcu, x, y)->chroma_deblocking |= EDGE_HOR;
    }
  }
}

/* ------------------------------------------------------------------------------------------------------------- search_cu -- */

static double search_cu(s_state *st, const s_loc *loc, const s_loc *chroma_loc, s_lcu *lcu, s_tree tree, int has_chroma)
{
  const orc_search_params *p = st->p;
  const int depth = tree.depth;
  const int x = loc->x, y = loc->y, cu_width = loc->w, cu_height = loc->h;
  const int is_separate_tree = chroma_loc == NULL || loc->h != chroma_loc->h || loc->w != chroma_loc->w;
  double cost = MAX_DOUBLE;
  double inter_bitcost = 2147483647;
  s_cabac pre_search_cabac = st->search;
  const int x_local = x % LCU, y_local = y % LCU;
  const int non_i = st->fr && st->fr->slice_type != 2;
  int32_t hmvp_lut[41];                    /* the row's history table at this CU's start (search.c:1330-1337) */
  if (non_i) memcpy(hmvp_lut, st->hmvp, sizeof hmvp_lut);
  if (x >= p->pic_w || y >= p->pic_h) return 0;

  s_cu *cur_cu = CU_AT(lcu, x_local, y_local);
  memset(cur_cu, 0, sizeof *cur_cu);
  cur_cu->type = CU_NOTSET;
  cur_cu->qp = (uint8_t)p->qp;
  cur_cu->split_tree = tree.split_tree;
  cur_cu->log2_w = (uint8_t)orc_log2i(cu_width);
  cur_cu->log2_h = (uint8_t)orc_log2i(cu_height);
  if (chroma_loc) { cur_cu->log2_ch = (uint8_t)orc_log2i(chroma_loc->ch); cur_cu->log2_cw = (uint8_t)orc_log2i(chroma_loc->cw); }
  const int mode_type_parent = (int)((tree.mode_type_tree >> ((depth - 1 > 0 ? depth - 1 : 0) * 2)) & 3);
  cur_cu->mode_type_tree = tree.mode_type_tree | (uint32_t)mode_type_parent << (depth * 2);

  s_cu pred_cu;
  memset(&pred_cu, 0, sizeof pred_cu);
  const int completely_inside = x + cu_width <= p->pic_w && y + cu_height <= p->pic_h;
  if (completely_inside) {
    /* check_can_use_inter (search.c:1212-1255) */
    int can_use_inter = non_i;
    if (can_use_inter) {
      const int dmin = st->fr->depth_inter_min, dmax = st->fr->depth_inter_max, min_wi = LCU >> dmax;
      if (depth > 6 || !((depth >= dmin && depth <= dmax) || (x & ~(min_wi - 1)) + min_wi > p->pic_w || (y & ~(min_wi - 1)) + min_wi > p->pic_h)) can_use_inter = 0;
      if (cu_width == 4 && cu_height == 4) can_use_inter = 0;
      if (loc->ch * loc->cw < 16) can_use_inter = 0;
      if (mode_type_parent == MODE_TYPE_INTRA) can_use_inter = 0;
    }
    if (can_use_inter) {
      double mode_cost, mode_bitcost;
      search_cu_inter(st, loc, lcu, &mode_cost, &mode_bitcost);
      if (mode_cost < cost) { cost = mode_cost; inter_bitcost = mode_bitcost; cur_cu->type = CU_INTER; }
    }
    /* no intra search when -- rd = 0 only -- the inter cost per sample is below INTRA_THRESHOLD = 8, or after an early skip (search.c:1413-1419) */
    const int skip_intra = (p->rd == 0 && cur_cu->type != CU_NOTSET && cost / (cu_width * cu_width) < 8) || (st->fr && st->fr->early_skip && cur_cu->skipped);
    /* check_can_use_intra (search.c:1257-1287) */
    const int min_w = LCU >> p->depth_max;
    int can_use_intra = 1;
    if (!((depth >= p->depth_min && depth <= p->depth_max) || (x & ~(min_w - 1)) + min_w > p->pic_w || (y & ~(min_w - 1)) + min_w > p->pic_h)) can_use_intra = 0;
    if (mode_type_parent == MODE_TYPE_INTER) can_use_intra = 0;
    if (can_use_intra && !skip_intra) {
      double intra_cost0 = 0;
      pred_cu = *cur_cu;
      /* uvg_search_cu_intra (search_intra.c:1771-1988) */
      {
        const s_cu *left_cu, *above_cu;
        int8_t cand[6];
        mpm_neighbours(loc, lcu, &left_cu, &above_cu);
        if (!(loc->x >= 4)) left_cu = NULL;
        if (!(loc->y >= 4 && y_local > 0)) above_cu = NULL;
        get_dir_luma_predictor(loc->y, cand, left_cu, above_cu);
        orc_px top[REF_LEN], left[REF_LEN];
        build_reference(st, loc, 0, lcu, top, left);
        pred_cu.type = CU_INTRA;
        double rough_cost = 0;
        const int best = search_intra_rough(st, loc, lcu, top, left, cand, &rough_cost);
        pred_cu.mode = (int8_t)best;
        pred_cu.mode_chroma = (int8_t)best;
        intra_cost0 = rough_cost;
      }
      double intra_cost = intra_cost0;      /* intra_search.cost: the rough cost of the best mode */
      if (intra_cost < cost) {
        int intra_mode = pred_cu.mode;
        if (has_chroma) {
          if (is_separate_tree) {
            /* uvg_get_co_located_luma_mode: the centre of the chroma area is in the last 4x4 CU, this one */
            intra_mode = pred_cu.mode;
            pred_cu.type = CU_INTRA;
          }
          pred_cu.mode_chroma = (int8_t)intra_mode;
          intra_recon_cu(st, pred_cu.mode, pred_cu.mode_chroma, chroma_loc, &pred_cu, lcu, 0, 1);
          intra_cost += cu_rd_cost_chroma(st, &pred_cu, lcu, chroma_loc);
        } else {
          pred_cu.mode_chroma = (int8_t)intra_mode;
        }
      }
      if (intra_cost < cost) {
        cost = intra_cost;
        *cur_cu = pred_cu;
        cur_cu->type = CU_INTRA;
        cur_cu->skipped = 0;
        cur_cu->merged = 0;
      }
    }
    if (cur_cu->type == CU_INTRA) {
      int recon_chroma = 1;
      const int recon_luma = 1;
      if (is_separate_tree || !has_chroma || loc->ch % 4 == 2) recon_chroma = 0;
      lcu_fill_cu_info(lcu, x_local, y_local, cu_width, cu_height, cur_cu);
      intra_recon_cu(st, cur_cu->mode, cur_cu->mode_chroma, loc, NULL, lcu, recon_luma, recon_chroma);
      if (!recon_chroma) {
        pred_cu.mode_chroma = cur_cu->mode_chroma;
        lcu_fill_chroma_cu_info(lcu, chroma_loc);
        intra_recon_cu(st, pred_cu.mode, pred_cu.mode_chroma, chroma_loc, NULL, lcu, 0, 1);
        lcu_fill_chroma_cbfs(lcu, chroma_loc);
      }
      lcu_fill_cu_info(lcu, x_local, y_local, cu_width, cu_height, cur_cu);
    } else if (cur_cu->type == CU_INTER) {
      if (!cur_cu->skipped) {
        if (!cur_cu->merged) {          /* uvg_round_precision(INTERNAL_MV_PREC, 2): to quarter samples and back */
          for (int l = 0; l < 2; ++l)
            if (cur_cu->mv_dir & (1 << l)) { cur_cu->mv[l][0] = (int32_t)((uint32_t)to_quarter(cur_cu->mv[l][0]) << 2); cur_cu->mv[l][1] = (int32_t)((uint32_t)to_quarter(cur_cu->mv[l][1]) << 2); }
        }
        inter_pred_pu(st, lcu, 1, 1, loc);
        quantize_lcu_residual(st, 1, 1, loc, NULL, lcu, 0);
        const int cbf = (cur_cu->cbf & 7) != 0 || cur_cu->root_cbf;
        if (cur_cu->merged && !cbf) {
          cur_cu->merged = 0;
          cur_cu->skipped = 1;
          const int skip_ctx = get_skip_context(x, y, lcu, NULL);
          inter_bitcost = ctx_fbits(&st->search.m, M_SKIP + skip_ctx, 1);
          inter_bitcost += ctx_fbits(&st->search.m, M_MERGE_IDX, cur_cu->merge_idx != 0);
          inter_bitcost += cur_cu->merge_idx;
        }
      }
      lcu_fill_cu_info(lcu, x_local, y_local, cu_width, cu_height, cur_cu);
      lcu_fill_cbf(lcu, x_local, y_local, cu_width, cu_height);
    }
  }
  (void)inter_bitcost;

  if (cur_cu->type == CU_INTRA || cur_cu->type == CU_INTER) {
    double bits = 0;
    s_cabac *cb = &st->search;
    cb->update = 1;
    bits += mock_encode_coding_unit(st, cb, loc, is_separate_tree && !has_chroma ? NULL : chroma_loc, lcu, cur_cu, tree);
    cost = bits * p->lambda;
    cost += cu_rd_cost_tr_split_accurate(st, cur_cu, lcu, loc, chroma_loc, has_chroma);
    cb->update = 0;
    mark_deblocking(loc, chroma_loc, lcu, has_chroma, is_separate_tree, cur_cu->skipped);
  }

  int can_split_cu = cur_cu->type == CU_NOTSET || depth < p->depth_max;
  int can_split[6];
  int is_implicit = possible_splits(st, loc, tree, can_split);
  {
    const int minimum_split_amount = p->depth_min - depth;
    if (minimum_split_amount > 0 && !is_implicit && can_split[1]) can_split[2] = can_split[3] = can_split[4] = can_split[5] = 0;
  }
  can_split_cu &= can_split[1] || can_split[2] || can_split[3] || can_split[4] || can_split[5];
  const int cbf = (cur_cu->cbf & 7) != 0;
  (void)cbf;        /* (cu-split-termination = zero is compiled to "always try", search.c:1824: "|| true") */

  if (can_split_cu) {
    s_lcu *split_lcu = (s_lcu *)calloc(1, sizeof(s_lcu));
    double best_split_cost = MAX_DOUBLE;
    s_cabac post_search_cabac = st->search, best_split_cabac = st->search;
    int have_split = 0;
    int32_t best_split_hmvp[41];
    if (non_i) memcpy(best_split_hmvp, st->hmvp, sizeof best_split_hmvp);
    if (can_split[QT_SPLIT]) {
      double split_cost = 0.0, split_bits = 0;
      const int cond = derive_mode_type_cond(loc, mode_type_parent);
      const int mode_type = cond == 1 ? MODE_TYPE_INTRA : mode_type_parent;
      int pruned = 0;
      st->search = pre_search_cabac;
      s_tree new_split = {tree.split_tree | (uint32_t)QT_SPLIT << (depth * 3), tree.mode_type_tree | (uint32_t)mode_type << (depth * 2), depth + 1,
                          tree.mtt_depth, tree.implicit_mtt_depth, 0};
      if (cur_cu->log2_h + cur_cu->log2_w > 4) {
        st->search.update = 1;
        const s_cu *left_cu = NULL, *above_cu = NULL;
        if (x) left_cu = CU_AT(lcu, x_local - 1, y_local);
        if (y) above_cu = CU_AT(lcu, x_local, y_local - 1);
        s_tree count_tree = tree;
        count_tree.split_tree = tree.split_tree | (uint32_t)QT_SPLIT << (depth * 3);
        count_tree.mode_type_tree = tree.mode_type_tree | (uint32_t)mode_type << (depth * 2);
        write_split_flag(st, &st->search, left_cu, above_cu, loc, count_tree, &is_implicit, &split_bits);
      }
      const double factor = p->qp > 30 ? 1.1 : 1.075;
      if (split_bits * p->lambda + cost / factor > cost) {
        pruned = 1;
      } else {
        st->search.update = 0;
        split_cost += split_bits * p->lambda;
        s_loc nl[4];
        const int hw = cu_width >> 1, hh = cu_height >> 1;
        loc_ctor(&nl[0], x, y, hw, hh); loc_ctor(&nl[1], x + hw, y, hw, hh); loc_ctor(&nl[2], x, y + hh, hw, hh); loc_ctor(&nl[3], x + hw, y + hh, hw, hh);
        int separate_chroma = hh == 4;
        separate_chroma |= !has_chroma;
        separate_chroma &= mode_type != MODE_TYPE_INTER;
        if (non_i) memcpy(st->hmvp, hmvp_lut, sizeof hmvp_lut);         /* (search.c:1972-1975) */
        initialize_partial_work_tree(st, lcu, split_lcu, loc, separate_chroma ? chroma_loc : loc);
        for (int split = 0; split < 4; ++split) {
          new_split.part_index = split;
          split_cost += search_cu(st, &nl[split], separate_chroma ? chroma_loc : &nl[split], split_lcu, new_split,
                                  !separate_chroma || (split == 3 && has_chroma));
          if (split_cost > cost || split_cost > best_split_cost) break;
        }
        have_split = 1;
        if (split_cost < best_split_cost) {
          best_split_cost = split_cost; best_split_cabac = st->search;
          if (non_i) memcpy(best_split_hmvp, st->hmvp, sizeof best_split_hmvp);
        }
      }
      (void)pruned;
    }

    /* combine_intra_cus (search.c:2082-2143): at a depth without a search, try the mode of the top-left CU of the next depth */
    if (cur_cu->type == CU_NOTSET && depth < 4 && x + cu_width <= p->pic_w && y + cu_width <= p->pic_h && p->combine_intra_cus && have_split) {
      const s_cu *cu_d1 = CU_AT(split_lcu, x_local, y_local);
      if (cu_d1->type == CU_INTRA && (cu_d1->log2_h + 1 == cur_cu->log2_h || cu_d1->log2_w + 1 == cur_cu->log2_w)) {
        const s_cabac temp_cabac = st->search;
        st->search = pre_search_cabac;
        cost = 0;
        double bits = 0;
        int is_impl = 0;
        write_split_flag(st, &st->search, x > 0 ? CU_AT(lcu, x_local - 1, y_local) : NULL, y > 0 ? CU_AT(lcu, x_local, y_local - 1) : NULL, loc, tree, &is_impl, &bits);
        cur_cu->mode = cu_d1->mode; cur_cu->mode_chroma = cu_d1->mode_chroma;
        cur_cu->type = CU_INTRA;
        lcu_fill_cu_info(lcu, x_local, y_local, cu_width, cu_height, cur_cu);
        intra_recon_cu(st, cur_cu->mode, cur_cu->mode_chroma, loc, NULL, lcu, 1, 1);
        /* calc_mode_bits (search.c:988-1003): uvg_luma_mode_bits on a copy of the models + uvg_chroma_mode_bits */
        double mode_bits = 0;
        {
          s_cabac copy = st->search;
          const s_cu *left_pu, *above_pu;
          mpm_neighbours(loc, lcu, &left_pu, &above_pu);
          encode_intra_luma(&copy, cur_cu->mode, loc, left_pu, above_pu, &mode_bits);
          if (cur_cu->mode_chroma == cur_cu->mode) mode_bits += ctx_fbits(&st->search.m, M_CHROMA_PRED, 0);
          else mode_bits += 2.0 + ctx_fbits(&st->search.m, M_CHROMA_PRED, 1);
        }
        mode_bits += bits;
        cost += mode_bits * p->lambda;
        cost += cu_rd_cost_tr_split_accurate(st, cur_cu, lcu, loc, chroma_loc, has_chroma);
        mark_deblocking(loc, chroma_loc, lcu, has_chroma, is_separate_tree, cur_cu->skipped);
        post_search_cabac = st->search;
        st->search = temp_cabac;
      }
    }

    if (best_split_cost < cost) {
      cost = best_split_cost;
      st->search = best_split_cabac;
      work_tree_copy_up(split_lcu, lcu, loc, is_separate_tree && !has_chroma ? NULL : chroma_loc);
      if (non_i) memcpy(st->hmvp, best_split_hmvp, sizeof best_split_hmvp);
    } else if (depth > 0) {
      st->search = post_search_cabac;
      if (non_i) { memcpy(st->hmvp, hmvp_lut, sizeof hmvp_lut); hmvp_add(st->hmvp, cur_cu); }
    }
    free(split_lcu);
  } else if (cur_cu->log2_h + cur_cu->log2_w > 4) {
    if (non_i) { memcpy(st->hmvp, hmvp_lut, sizeof hmvp_lut); hmvp_add(st->hmvp, cur_cu); }
  }
  return cost;
}

/* --------------------------------------------------------------------------------- the real coder's model adaptation -- */
/* uvg_encode_coding_tree (encode_coding_tree.c:1365-1727) on the picture's cu array: only which models see which bins matters */
typedef struct s_frame {
  const orc_search_params *p;
  s_cu *cua;            /* frame->cu_array: one entry per 4x4, stride */
  int cu_stride;
} s_frame;
static const s_cu *cua_at(const s_frame *f, int x, int y) { return &f->cua[(y >> 2) * f->cu_stride + (x >> 2)]; }

static double g_tree_bits;        /* the bit estimate of everything encode_coding_tree codes (count mode: bypass bins = this - the regular share) */
static void code_coeffs(s_cabac *cb, const int16_t *plane, int stride, int lx, int ly, int w, int h, int color)
{
  g_tree_bits += coeff_cost_cu(cb, plane, stride, lx, ly, w, h, color);
}

static void encode_transform_coeff(const s_frame *f, s_cabac *cb, const s_loc *loc, int only_chroma, const int16_t *cy, const int16_t *cu, const int16_t *cv,
                                   int *luma_cbf_ctx, const s_loc *chroma_loc)
{
  const s_cu *cur_tu = cua_at(f, loc->x, loc->y);
  if (loc->w > 32 || loc->h > 32) {
    const int hw = loc->w >> 1, hh = loc->h >> 1;
    for (int i = 0; i < 4; ++i) {
      s_loc sl;
      loc_ctor(&sl, loc->x + (i & 1) * hw, loc->y + (i >> 1) * hh, hw, hh);
      encode_transform_coeff(f, cb, &sl, only_chroma, cy, cu, cv, luma_cbf_ctx, chroma_loc ? &sl : NULL);
    }
    return;
  }
  const int cb_flag_y = cur_tu->cbf & 1, cb_flag_u = (cur_tu->cbf >> 1) & 1, cb_flag_v = (cur_tu->cbf >> 2) & 1;
  if (chroma_loc || only_chroma) {
    fbits_update(cb, M_CBF_CB + 0, cb_flag_u, &g_tree_bits);
    fbits_update(cb, M_CBF_CR + (cb_flag_u ? 1 : 0), cb_flag_v, &g_tree_bits);
  }
  const int pu_is_tu = cur_tu->log2_w <= 5 && cur_tu->log2_h <= 5;
  if ((cur_tu->type == CU_INTRA || !pu_is_tu || cb_flag_u || cb_flag_v) && !only_chroma) {
    fbits_update(cb, M_CBF_LUMA + *luma_cbf_ctx, cb_flag_y, &g_tree_bits);
    if (pu_is_tu) *luma_cbf_ctx = 2 + cb_flag_y;
  }
  if (cb_flag_y | cb_flag_u | cb_flag_v) {
    /* encode_transform_unit (:530-626) */
    if (cb_flag_y && !only_chroma) code_coeffs(cb, cy, LCU, loc->lx, loc->ly, loc->w, loc->h, 0);
    if (cur_tu->log2_h + cur_tu->log2_w < 6 && !only_chroma) return;
    if ((cb_flag_u || cb_flag_v) && chroma_loc) {
      const int px = (chroma_loc->x >> 1) % LCU_C, py = (chroma_loc->y >> 1) % LCU_C;
      if (cb_flag_u) code_coeffs(cb, cu, LCU_C, px, py, chroma_loc->cw, chroma_loc->ch, 1);
      if (cb_flag_v) code_coeffs(cb, cv, LCU_C, px, py, chroma_loc->cw, chroma_loc->ch, 2);
    }
  }
}

static void encode_coding_tree(const s_frame *f, s_state *st, s_cabac *cb, const int16_t *cy, const int16_t *cu, const int16_t *cv, const s_loc *loc,
                               const s_loc *chroma_loc, s_tree tree, int has_chroma)
{
  const orc_search_params *p = f->p;
  const int x = loc->x, y = loc->y;
  if (x >= p->pic_w || y >= p->pic_h) return;
  const s_cu *cur_cu = cua_at(f, x, y);
  const s_cu *left_cu = x > 0 ? cua_at(f, x - 1, y) : NULL, *above_cu = y > 0 ? cua_at(f, x, y - 1) : NULL;
  const int depth = tree.depth;
  const int mode_type_curr = (int)((cur_cu->mode_type_tree >> (depth * 2)) & 3);
  if (loc->w + loc->h > 8) {
    tree.split_tree = cur_cu->split_tree;
    tree.mode_type_tree = cur_cu->mode_type_tree;
    int is_implicit;
    write_split_flag(st, cb, left_cu, above_cu, loc, tree, &is_implicit, &g_tree_bits);
    const int split_flag = (int)((tree.split_tree >> (depth * 3)) & 7);
    if (split_flag != NO_SPLIT) {
      s_tree nt = {cur_cu->split_tree, cur_cu->mode_type_tree, depth + 1, tree.mtt_depth, tree.implicit_mtt_depth, 0};
      s_loc nl[4];
      const int hw = loc->w >> 1, hh = loc->h >> 1;
      loc_ctor(&nl[0], x, y, hw, hh); loc_ctor(&nl[1], x + hw, y, hw, hh); loc_ctor(&nl[2], x, y + hh, hw, hh); loc_ctor(&nl[3], x + hw, y + hh, hw, hh);
      int separate_chroma = hh == 4;
      separate_chroma |= !has_chroma;
      separate_chroma &= mode_type_curr != MODE_TYPE_INTER;
      for (int s = 0; s < 4; ++s) {
        nt.part_index = s;
        encode_coding_tree(f, st, cb, cy, cu, cv, &nl[s], separate_chroma ? chroma_loc : &nl[s], nt, !separate_chroma || (s == 3 && has_chroma));
      }
      return;
    }
  }
  const int non_i = st->fr && st->fr->slice_type != 2;
  if (non_i) {                  /* skip flag, prediction mode, the inter prediction unit (encode_coding_tree.c:1470-1640) */
    int ctx_skip = 0;
    if (left_cu && left_cu->skipped) ctx_skip++;
    if (above_cu && above_cu->skipped) ctx_skip++;
    if ((loc->w != 4 || loc->h != 4) && mode_type_curr != MODE_TYPE_INTRA) fbits_update(cb, M_SKIP + ctx_skip, cur_cu->skipped, &g_tree_bits);
    if (cur_cu->skipped) {
      hmvp_add(st->hmvp, cur_cu);
      const int num_cand = st->fr->max_merge;
      if (num_cand > 1)
        for (int ui = 0; ui < num_cand - 1; ui++) {
          const int symbol = ui != cur_cu->merge_idx;
          if (ui == 0) fbits_update(cb, M_MERGE_IDX, symbol, &g_tree_bits);
          else { ORC_FN(cabac_sim_ep)((uint32_t)symbol); g_tree_bits += 1; }
          if (symbol == 0) break;
        }
      return;
    }
    if ((loc->w != 4 || loc->h != 4) && mode_type_curr == MODE_TYPE_ALL) {
      const int ctx_predmode = (left_cu && left_cu->type == CU_INTRA) || (above_cu && above_cu->type == CU_INTRA);
      fbits_update(cb, M_PRED_MODE + ctx_predmode, cur_cu->type == CU_INTRA, &g_tree_bits);
    }
    if (cur_cu->type == CU_INTER) {
      double pu_bits = 0;
      encode_inter_prediction_unit(st, cb, cur_cu, NULL, &pu_bits, loc);
      g_tree_bits += pu_bits;          /* (an estimate only: uvg_encode_mvd's assignment loses part of it; nothing reads it for P / B pictures) */
      hmvp_add(st->hmvp, cur_cu);
      const int has_coeffs = cur_cu->root_cbf || cur_cu->cbf;
      if (!cur_cu->merged) fbits_update(cb, M_ROOT_CBF, has_coeffs, &g_tree_bits);
      if (has_coeffs) {
        int luma_cbf_ctx = 0;
        encode_transform_coeff(f, cb, loc, 0, cy, cu, cv, &luma_cbf_ctx, loc);
      }
      return;
    }
  }
  /* an intra CU */
  {
    const s_cu *left_pu = NULL, *above_pu = NULL;
    if (x > 0) left_pu = cua_at(f, x - 1, y + loc->h - 1);
    if (y % LCU > 0 && y > 0) above_pu = cua_at(f, x + loc->w - 1, y - 1);
    encode_intra_luma(cb, cur_cu->mode, loc, left_pu, above_pu, &g_tree_bits);
  }
  const int is_local_dual_tree = chroma_loc->w != loc->w || chroma_loc->h != loc->h;
  if (!is_local_dual_tree) encode_chroma_intra(cb, cur_cu->mode_chroma, cur_cu->mode, &g_tree_bits);
  int luma_cbf_ctx = 0;
  encode_transform_coeff(f, cb, loc, 0, cy, cu, cv, &luma_cbf_ctx, is_local_dual_tree ? NULL : chroma_loc);
  if (is_local_dual_tree && has_chroma) {
    /* uvg_get_co_located_luma_mode on the cu array: the centre of the chroma area */
    const int luma_dir = cua_at(f, chroma_loc->x + (chroma_loc->w >> 1), chroma_loc->y + (chroma_loc->h >> 1))->mode;
    encode_chroma_intra(cb, cur_cu->mode_chroma, luma_dir, &g_tree_bits);
    encode_transform_coeff(f, cb, chroma_loc, 1, cy, cu, cv, &luma_cbf_ctx, chroma_loc);
  }
}

/*
 * A side effect of the deblocking filter on the picture's cu array that the inter path sees: where filter_deblock_edge_luma derives the
 * boundary strength from motion (both sides inter, no coded luma residual at the edge, a B slice or a bi-predicted side) it first zeroes
 * the vectors of the unused lists of BOTH cu_info_t entries, in place (src/filter.c:745-765).  The entries feed the coder's history
 * table, the next CTUs' neighbour rows and -- through merge_candidate_in_list's comparison of all fields (search_inter.c:1639-1658) -- later
 * decisions.  uvg_filter_deblock_lcu (filter.c:1372-1380) runs between a CTU's search and its coding tree; the units it visits are those
 * of oracle/orc_deblock.c's deblock_lcu (the zeroing is idempotent and its condition does not depend on it, so the order inside the
 * call does not matter).
 */
static void deblock_zeroes_at(s_frame *f, int bx, int by, int dir_hor, int is_b)
{
  if ((!dir_hor && bx == 0) || (dir_hor && by == 0)) return;
  if (bx >= f->p->pic_w || by >= f->p->pic_h) return;
  s_cu *q = &f->cua[(by >> 2) * f->cu_stride + (bx >> 2)];
  if (!(q->luma_deblocking & (dir_hor ? EDGE_HOR : EDGE_VER))) return;
  s_cu *p = dir_hor ? q - f->cu_stride : q - 1;
  if (q->type == CU_INTRA || p->type == CU_INTRA) return;
  if ((q->cbf & 1) || (p->cbf & 1)) return;              /* tu_boundary && nonzero_coeffs: strength 1 without looking at motion */
  if (!(p->mv_dir == 3 || q->mv_dir == 3 || is_b)) return;
  for (int l = 0; l < 2; ++l) {
    if (!(q->mv_dir & (1 << l))) { q->mv[l][0] = 0; q->mv[l][1] = 0; }
    if (!(p->mv_dir & (1 << l))) { p->mv[l][0] = 0; p->mv[l][1] = 0; }
  }
}
static void deblock_zeroes_unused_vectors(s_frame *f, int x_px, int y_px, int is_b)
{
  const int W = f->p->pic_w, H = f->p->pic_h;
  const int end_x = x_px + 64 < W ? x_px + 64 : W, end_y = y_px + 64 < H ? y_px + 64 : H;
  for (int by = y_px; by < end_y; by += 4)
    for (int bx = x_px; bx < end_x; bx += 4) deblock_zeroes_at(f, bx, by, 0, is_b);
  if (x_px > 0)
    for (int bx = x_px - 8; bx < x_px; bx += 4)
      for (int by = y_px; by < end_y; by += 4) deblock_zeroes_at(f, bx, by, 1, is_b);
  for (int by = y_px; by < end_y; by += 4)
    for (int bx = x_px; bx < end_x; bx += 4) {
      if ((bx & 63) >= 56 && bx < W - 8) continue;
      deblock_zeroes_at(f, bx, by, 1, is_b);
    }
}

/* ---------------------------------------------------------------------------------------------------------- the picture -- */
/*
 * One all-intra picture, CTU by CTU in raster order (any order that respects the WPP dependencies gives the same result):
 *   y/u/v: source planes (pic_w x pic_h, pic_w/2 x pic_h/2), tightly packed
 *   rec_*: reconstruction before the in-loop filters, same layout
 *   cu_out[(pic_h/4... rounded up to CTUs) * cu_stride]: see s_cu, cu_stride = 16 * ctus per row
 *   coeff_out: per CTU 64*64 + 2*32*32 levels (lcu_coeff_t order: y, u, v; raster inside the CTU)
 *   models_out: per CTU three model sets: at the CTU's start, at the end of its search, after the real coder
 */
static int search_picture(const orc_search_params *p, const orc_inter_frame *fr, const orc_px *src_y, const orc_px *src_u, const orc_px *src_v,
                          orc_px *rec_y, orc_px *rec_u, orc_px *rec_v, uint8_t *cu_out, int16_t *coeff_out, orc_models_ext *models_out,
                          int32_t *motion_out, uint8_t *inter_extra_out, orc_models_inter *models_inter_out)
{
  fbits_init();
  if (p->pic_w % 8 || p->pic_h % 8) return -1;
  const int non_i = fr && fr->slice_type != 2;
  const int W = p->pic_w, H = p->pic_h, wc = (W + 63) / 64, hc = (H + 63) / 64;
  const int cu_stride = wc * 16;
  s_frame f = {p, (s_cu *)calloc((size_t)cu_stride * hc * 16, sizeof(s_cu)), cu_stride};
  orc_models *row_start = (orc_models *)calloc((size_t)hc, sizeof(orc_models));   /* models handed to the next row after its first CTU */
  s_lcu *lcu = (s_lcu *)malloc(sizeof(s_lcu));
  s_state *st = (s_state *)calloc(1, sizeof(s_state));
  st->p = p;
  st->c_lambda = p->c_lambda;
  st->fr = fr;
  st->src_y = src_y;
  st->cua = f.cua; st->cu_stride = cu_stride;
  int32_t hmvp[41];
  int32_t *col = NULL;
  if (non_i) {
    /* the collocated picture (L0[0]) on the 8x8 grid, as uvg_inter_get_merge_cand / _mv_cand read it through get_temporal_merge_candidates */
    const int gw = (W + 7) / 8, gh = (H + 7) / 8;
    col = (int32_t *)calloc((size_t)gw * gh * 8, sizeof(int32_t));
    if (fr->n_refs && fr->l_size[0] > 0) {
      const int32_t *rc = fr->ref_cu[fr->l[0][0]];
      for (int gy = 0; gy < gh; ++gy)
        for (int gx = 0; gx < gw; ++gx) memcpy(col + ((size_t)gy * gw + gx) * 8, rc + ((size_t)(gy * 2) * fr->ref_cu_stride + gx * 2) * 8, 8 * sizeof(int32_t));
    }
    st->col = col;
    st->hmvp = hmvp;
  }
  orc_models coder;     /* state->cabac.ctx of the CTU row */
  for (int cyi = 0; cyi < hc; ++cyi) {
    if (non_i) memset(hmvp, 0, sizeof hmvp);         /* the row's history table starts empty (encoderstate.c:1021-1028) */
    for (int cxi = 0; cxi < wc; ++cxi) {
      const int x = cxi * 64, y = cyi * 64;
      if (cxi == 0) {
        if (cyi == 0 || !p->wpp) { if (cyi == 0) models_init(&coder, fr ? fr->frame_qp : p->qp, fr ? fr->slice_type : 2); }
        else coder = row_start[cyi - 1];
      }
      orc_models_ext *mo = &models_out[(size_t)(cyi * wc + cxi) * 3];
      orc_models_inter *mi = models_inter_out ? &models_inter_out[(size_t)(cyi * wc + cxi) * 3] : NULL;
      models_to_ext(&coder, &mo[0], mi ? &mi[0] : NULL);
      /* uvg_search_lcu */
      st->search.m = coder;
      st->search.update = 0;
      rdoq_ctx_from(&coder, &st->rdoq);
      /* init_lcu_t (search.c:2230-2330) */
      memset(lcu, 0, sizeof *lcu);
      if (y > 0) for (int i = 0; i < 64 && x + i < W; i += 4) *CU_AT(lcu, i, -1) = *cua_at(&f, x + i, y - 1);
      if (x > 0) for (int i = 0; i < 64 && y + i < H; i += 4) *CU_AT(lcu, -1, i) = *cua_at(&f, x - 1, y + i);
      if (x > 0 && y > 0) *CU_AT(lcu, -1, -1) = *cua_at(&f, x - 1, y - 1);
      if (y > 0 && x + 64 < W && !p->wpp) lcu->cu[TCW * TCW] = *cua_at(&f, x + 64, y - 1);
      if (y > 0) {
        const int x_max = 96 < W - x ? 96 : W - x, x_min = x > 0 ? 0 : 1;
        for (int i = x_min - 1; i < x_max; ++i) lcu->top_y[i + 1] = rec_y[(y - 1) * W + x + i];
        for (int i = x_min - 1; i < x_max / 2; ++i) { lcu->top_u[i + 1] = rec_u[(y / 2 - 1) * (W / 2) + x / 2 + i]; lcu->top_v[i + 1] = rec_v[(y / 2 - 1) * (W / 2) + x / 2 + i]; }
      }
      if (x > 0) {
        const int y_min = y > 0 ? 0 : 1;
        for (int i = y_min - 1; i < 64 && y + i < H; ++i) lcu->left_y[i + 1] = rec_y[(y + i) * W + x - 1];
        for (int i = y_min - 1; i < 32 && y / 2 + i < H / 2; ++i) { lcu->left_u[i + 1] = rec_u[(y / 2 + i) * (W / 2) + x / 2 - 1]; lcu->left_v[i + 1] = rec_v[(y / 2 + i) * (W / 2) + x / 2 - 1]; }
      }
      const int x_max = (x + 64 < W ? x + 64 : W) - x, y_max = (y + 64 < H ? y + 64 : H) - y;
      for (int j = 0; j < y_max; ++j) memcpy(&lcu->ref_y[j * LCU], &src_y[(y + j) * W + x], (size_t)x_max * sizeof(orc_px));
      for (int j = 0; j < y_max / 2; ++j) {
        memcpy(&lcu->ref_u[j * LCU_C], &src_u[(y / 2 + j) * (W / 2) + x / 2], (size_t)(x_max / 2) * sizeof(orc_px));
        memcpy(&lcu->ref_v[j * LCU_C], &src_v[(y / 2 + j) * (W / 2) + x / 2], (size_t)(x_max / 2) * sizeof(orc_px));
      }
      s_loc start;
      loc_ctor(&start, x, y, 64, 64);
      s_tree tree = {0, MODE_TYPE_ALL, 0, 0, 0, 0};
      int32_t hmvp_before[41];
      if (non_i) memcpy(hmvp_before, hmvp, sizeof hmvp);
      search_cu(st, &start, &start, lcu, tree, 1);
      if (non_i) memcpy(hmvp, hmvp_before, sizeof hmvp);         /* the search's additions are dropped (encoderstate.c:757-813); the coder's stay */
      models_to_ext(&st->search.m, &mo[1], mi ? &mi[1] : NULL);
      /* copy_lcu_to_cu_data + coefficients */
      for (int j = 0; j < y_max; j += 4) for (int i = 0; i < x_max; i += 4) f.cua[((y + j) >> 2) * cu_stride + ((x + i) >> 2)] = *CU_AT(lcu, i, j);
      for (int j = 0; j < y_max; ++j) memcpy(&rec_y[(y + j) * W + x], &lcu->rec_y[j * LCU], (size_t)x_max * sizeof(orc_px));
      for (int j = 0; j < y_max / 2; ++j) {
        memcpy(&rec_u[(y / 2 + j) * (W / 2) + x / 2], &lcu->rec_u[j * LCU_C], (size_t)(x_max / 2) * sizeof(orc_px));
        memcpy(&rec_v[(y / 2 + j) * (W / 2) + x / 2], &lcu->rec_v[j * LCU_C], (size_t)(x_max / 2) * sizeof(orc_px));
      }
      if (motion_out || inter_extra_out)          /* as the records of tools/refcheck/ctu_dump.c: the cu array right after the CTU's search */
        for (int j = 0; j < 64; j += 4)
          for (int i = 0; i < 64; i += 4) {
            const size_t at = (size_t)((y + j) >> 2) * cu_stride + ((x + i) >> 2);
            const s_cu *c = &f.cua[at];
            if (motion_out) {
              int32_t *m = &motion_out[at * 8];
              memset(m, 0, 8 * sizeof(int32_t));
              if (c->type == CU_INTER && x + i < W && y + j < H) {
                m[0] = c->mv[0][0]; m[1] = c->mv[0][1]; m[2] = c->mv[1][0]; m[3] = c->mv[1][1]; m[4] = c->mv_ref[0]; m[5] = c->mv_ref[1]; m[6] = c->mv_dir;
                m[7] = c->skipped | c->merged << 1 | c->merge_idx << 2;
              }
            }
            if (inter_extra_out) { uint8_t *e = &inter_extra_out[at * 4]; e[0] = c->root_cbf; e[1] = c->mv_cand0; e[2] = c->mv_cand1; e[3] = 0; }
          }
      if (non_i) deblock_zeroes_unused_vectors(&f, x, y, fr->slice_type == 0);
      int16_t *co = &coeff_out[(size_t)(cyi * wc + cxi) * (64 * 64 + 2 * 32 * 32)];
      memcpy(co, lcu->coeff_y, sizeof lcu->coeff_y); memcpy(co + 4096, lcu->coeff_u, sizeof lcu->coeff_u); memcpy(co + 4096 + 1024, lcu->coeff_v, sizeof lcu->coeff_v);
      /* encoder_state_worker_encode_lcu_bitstream: the real coder adapts the row's models */
      {
        s_cabac cb;
        cb.m = coder; cb.update = 1;
        encode_coding_tree(&f, st, &cb, co, co + 4096, co + 4096 + 1024, &start, &start, tree, 1);
        coder = cb.m;
      }
      models_to_ext(&coder, &mo[2], mi ? &mi[2] : NULL);
      if (cxi == 0) row_start[cyi] = coder;
    }
  }
  /* compact cu array for the caller: 12 bytes + two trees per 4x4 (the layout of tools/refcheck/ctu_dump.c) */
  for (int j = 0; j < hc * 16; ++j)
    for (int i = 0; i < cu_stride; ++i) {
      const s_cu *c = &f.cua[j * cu_stride + i];
      uint8_t *o = &cu_out[((size_t)j * cu_stride + i) * 20];
      o[0] = c->type; o[1] = c->log2_w; o[2] = c->log2_h; o[3] = c->log2_cw; o[4] = c->log2_ch; o[5] = c->cbf; o[6] = (uint8_t)c->mode;
      o[7] = (uint8_t)c->mode_chroma; o[8] = c->luma_deblocking; o[9] = c->chroma_deblocking; o[10] = c->qp; o[11] = 0;
      memcpy(o + 12, &c->split_tree, 4); memcpy(o + 16, &c->mode_type_tree, 4);
    }
  free(f.cua); free(row_start); free(lcu); free(st); free(col);
  return 0;
}

ORC_EXPORT int ORC_FN(search_intra_picture)(const orc_search_params *p, const orc_px *src_y, const orc_px *src_u, const orc_px *src_v,
                                            orc_px *rec_y, orc_px *rec_u, orc_px *rec_v, uint8_t *cu_out, int16_t *coeff_out, orc_models_ext *models_out)
{
  return search_picture(p, NULL, src_y, src_u, src_v, rec_y, rec_u, rec_v, cu_out, coeff_out, models_out, NULL, NULL, NULL);
}

/*
 * One P / B picture of a low-delay encode (BASELINE configs[2]: --gop lp-g4d3t1 --preset medium), CTU by CTU: uvg_search_lcu with the
 * inter search (orc_search_inter.inc) competing with the intra search, the real coder's model adaptation and history table in between.
 * fr: the picture's reference lists and reference pictures (orc_inter_frame).  p->qp / lambda: the PICTURE's (the encoder's QP offsets per
 * GOP layer and its lambda derivation stay with the caller).  Outputs as search_intra_picture, plus motion_out [per 4x4][8] (mv[2][2],
 * mv_ref[2], mv_dir, skipped | merged << 1 | merge_idx << 2), inter_extra_out [per 4x4][4] (root_cbf, mv_cand0, mv_cand1, 0) and the
 * 18 inter-syntax models beside each of the three model sets.
 */
ORC_EXPORT int ORC_FN(search_inter_picture)(const orc_search_params *p, const orc_inter_frame *fr, const orc_px *src_y, const orc_px *src_u,
                                            const orc_px *src_v, orc_px *rec_y, orc_px *rec_u, orc_px *rec_v, uint8_t *cu_out, int16_t *coeff_out,
                                            orc_models_ext *models_out, int32_t *motion_out, uint8_t *inter_extra_out, orc_models_inter *models_inter_out)
{
  return search_picture(p, fr, src_y, src_u, src_v, rec_y, rec_u, rec_v, cu_out, coeff_out, models_out, motion_out, inter_extra_out, models_inter_out);
}

/* a trace of search_cu_inter's calls for debugging against the "cuinter" records of tools/refcheck/ctu_dump.c: 22 doubles per call */
ORC_EXPORT void ORC_FN(search_trace)(double *buf, int cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
ORC_EXPORT int ORC_FN(search_trace_count)(void) { return g_trace_n; }
/* per-call inputs of search_pu_inter (see orc_search_inter.inc: CTXT_I ints, CTXT_D doubles per call) */
ORC_EXPORT void ORC_FN(search_ctx_trace)(int32_t *ints, double *doubles, int cap) { g_ctxt_i = ints; g_ctxt_d = doubles; g_ctxt_cap = cap; g_ctxt_n = 0; }
ORC_EXPORT int ORC_FN(search_ctx_trace_count)(void) { return g_ctxt_n; }


/*
 * The hand-over consumer: what a bitstream coder does with the search's outputs, in count mode.  cu: the compact side information
 * written above (20 bytes per 4x4: the cu_info_t fields uvg_encode_coding_tree reads + the two trees), coeff: lcu_coeff_t per
 * CTU, start[ctu]: the coder's models when the CTU's coding tree begins, range_in[ctu]: the arithmetic coder's range at that
 * point.  Walks uvg_encode_coding_tree (encode_coding_tree.c:1365-1727) for every CTU with the arithmetic coder's range
 * arithmetic (uvg_cabac_encode_bin, cabac.c:76-109) and returns per CTU the bits the coder consumes for the tree
 * (renormalisation shifts of the context-coded bins + one per bypass bin), its range afterwards and the models afterwards.
 */
ORC_EXPORT int ORC_FN(count_picture_bits)(const orc_search_params *p, const uint8_t *cu, const int16_t *coeff, const orc_models_ext *start,
                                          const int64_t *range_in, int64_t *bits_out, int64_t *range_out, orc_models_ext *after)
{
  fbits_init();
  const int W = p->pic_w, H = p->pic_h, wc = (W + 63) / 64, hc = (H + 63) / 64, cu_stride = wc * 16;
  s_frame f = {p, (s_cu *)calloc((size_t)cu_stride * hc * 16, sizeof(s_cu)), cu_stride};
  for (int j = 0; j < hc * 16; ++j)
    for (int i = 0; i < cu_stride; ++i) {
      s_cu *c = &f.cua[j * cu_stride + i];
      const uint8_t *o = &cu[((size_t)j * cu_stride + i) * 20];
      c->type = o[0]; c->log2_w = o[1]; c->log2_h = o[2]; c->log2_cw = o[3]; c->log2_ch = o[4]; c->cbf = o[5]; c->mode = (int8_t)o[6];
      c->mode_chroma = (int8_t)o[7]; c->luma_deblocking = o[8]; c->chroma_deblocking = o[9]; c->qp = o[10];
      memcpy(&c->split_tree, o + 12, 4); memcpy(&c->mode_type_tree, o + 16, 4);
    }
  s_state *st = (s_state *)calloc(1, sizeof(s_state));
  st->p = p;
  for (int k = 0; k < wc * hc; ++k) {
    const int16_t *co = &coeff[(size_t)k * 6144];
    s_cabac cb;
    models_from_ext(&cb.m, &start[k]); cb.update = 1;
    orc_cabac_sim *sim = &ORC_FN(cabac_sim);
    sim->on = 1; sim->range = (uint32_t)range_in[k]; sim->shifts = 0; sim->regular_fbits = 0.0;
    g_tree_bits = 0.0;
    s_loc start_loc;
    loc_ctor(&start_loc, (k % wc) * 64, (k / wc) * 64, 64, 64);
    s_tree tree = {0, MODE_TYPE_ALL, 0, 0, 0, 0};
    encode_coding_tree(&f, st, &cb, co, co + 4096, co + 4096 + 1024, &start_loc, &start_loc, tree, 1);
    sim->on = 0;
    bits_out[k] = (int64_t)sim->shifts + (int64_t)floor(g_tree_bits - sim->regular_fbits + 0.5);
    range_out[k] = sim->range;
    models_to_ext(&cb.m, &after[k], NULL);
  }
  free(f.cua); free(st);
  return 0;
}


/*
 * The same consumer as a bitstream coder: the whole arithmetic coder (uvg_cabac_encode_bin / _bin_ep / _bins_ep / uvg_cabac_write,
 * cabac.c:76-311) over every CTU's coding tree.  state_in[ctu][5]: low, range, bits_left, num_buffered_bytes, buffered_byte when
 * the tree begins; state_out likewise when it ends; bytes_out / byte_off[ctu + 1]: the payload bytes the coder hands to the
 * bitstream during each CTU's tree (before emulation prevention).  Returns the total number of bytes, or -1 if bytes_cap is too small.
 */
ORC_EXPORT long ORC_FN(encode_picture_ctus)(const orc_search_params *p, const uint8_t *cu, const int16_t *coeff, const orc_models_ext *start,
                                            const int64_t *state_in, int64_t *state_out, uint8_t *bytes_out, long bytes_cap, int64_t *byte_off)
{
  fbits_init();
  const int W = p->pic_w, H = p->pic_h, wc = (W + 63) / 64, hc = (H + 63) / 64, cu_stride = wc * 16;
  s_frame f = {p, (s_cu *)calloc((size_t)cu_stride * hc * 16, sizeof(s_cu)), cu_stride};
  for (int j = 0; j < hc * 16; ++j)
    for (int i = 0; i < cu_stride; ++i) {
      s_cu *c = &f.cua[j * cu_stride + i];
      const uint8_t *o = &cu[((size_t)j * cu_stride + i) * 20];
      c->type = o[0]; c->log2_w = o[1]; c->log2_h = o[2]; c->log2_cw = o[3]; c->log2_ch = o[4]; c->cbf = o[5]; c->mode = (int8_t)o[6];
      c->mode_chroma = (int8_t)o[7]; c->luma_deblocking = o[8]; c->chroma_deblocking = o[9]; c->qp = o[10];
      memcpy(&c->split_tree, o + 12, 4); memcpy(&c->mode_type_tree, o + 16, 4);
    }
  s_state *st = (s_state *)calloc(1, sizeof(s_state));
  st->p = p;
  orc_cabac_sim *sim = &ORC_FN(cabac_sim);
  long total = 0;
  byte_off[0] = 0;
  for (int k = 0; k < wc * hc; ++k) {
    const int16_t *co = &coeff[(size_t)k * 6144];
    s_cabac cb;
    models_from_ext(&cb.m, &start[k]); cb.update = 1;
    sim->on = 2; sim->shifts = 0; sim->regular_fbits = 0.0; sim->out_len = 0;
    sim->low = (uint32_t)state_in[5 * k]; sim->range = (uint32_t)state_in[5 * k + 1]; sim->bits_left = (int32_t)state_in[5 * k + 2];
    sim->num_buffered_bytes = (int32_t)state_in[5 * k + 3]; sim->buffered_byte = (uint32_t)state_in[5 * k + 4];
    g_tree_bits = 0.0;
    s_loc start_loc;
    loc_ctor(&start_loc, (k % wc) * 64, (k / wc) * 64, 64, 64);
    s_tree tree = {0, MODE_TYPE_ALL, 0, 0, 0, 0};
    encode_coding_tree(&f, st, &cb, co, co + 4096, co + 4096 + 1024, &start_loc, &start_loc, tree, 1);
    sim->on = 0;
    state_out[5 * k] = sim->low; state_out[5 * k + 1] = sim->range; state_out[5 * k + 2] = sim->bits_left;
    state_out[5 * k + 3] = sim->num_buffered_bytes; state_out[5 * k + 4] = sim->buffered_byte;
    if (total + (long)sim->out_len > bytes_cap) { free(f.cua); free(st); return -1; }
    memcpy(bytes_out + total, sim->out, sim->out_len);
    total += (long)sim->out_len;
    byte_off[k + 1] = total;
  }
  free(f.cua); free(st);
  return total;
}


/* ---- the whole slice data: every WPP row's substream (encoderstate.c:862-939) -------------------------------------------------- */
typedef struct { uint16_t s0[2], s1[2]; uint8_t rate[2]; } sao_models2;      /* sao_merge_flag_model, sao_type_idx_model */
static void sao_models2_init(sao_models2 *m, int qp)
{
  for (int i = 0; i < 2; ++i) {
    const int v = k_ctx_init_sao[2][i];
    const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
    int s = ((slope * (qp - 16)) >> 1) + offset;
    s = s < 1 ? 1 : (s > 127 ? 127 : s);
    m->s0[i] = (uint16_t)((s << 8) & 0x7fe0); m->s1[i] = (uint16_t)((s << 8) & 0x7ffe);
    m->rate[i] = k_ctx_init_sao[3][i];
  }
}
static void sao_bin(sao_models2 *m, int c, int bin)                          /* CABAC_BIN on one of the two SAO models */
{
  ORC_FN(cabac_sim_bin)((m->s0[c] + m->s1[c]) >> 8, bin);
  const int r0 = m->rate[c] >> 4, r1 = m->rate[c] & 15;
  m->s0[c] = (uint16_t)(m->s0[c] - ((m->s0[c] >> r0) & 0x7fe0u));
  m->s1[c] = (uint16_t)(m->s1[c] - ((m->s1[c] >> r1) & 0x7ffeu));
  if (bin) { m->s0[c] = (uint16_t)(m->s0[c] + ((0x7fffu >> r0) & 0x7fe0u)); m->s1[c] = (uint16_t)(m->s1[c] + ((0x7fffu >> r1) & 0x7ffeu)); }
}
/* encode_sao_color (encoderstate.c:523-573); info: sao_info_t as 17 ints (type, eo_class, ddistortion, merge_left, merge_up,
 * band_position[2], offsets[10]) */
static void encode_sao_color(sao_models2 *m, const int32_t *info, int color)
{
  const int type = info[0], off = color == 2 ? 5 : 0;
  const int max_off = (1 << ((ORC_BIT_DEPTH < 10 ? ORC_BIT_DEPTH : 10) - 5)) - 1;
  if (color != 2) {
    sao_bin(m, 1, type != 0);
    if (type == 1) ORC_FN(cabac_sim_ep)(0);
    else if (type == 2) ORC_FN(cabac_sim_ep)(1);
  }
  if (type == 0) return;
  for (int i = 1; i <= 4; ++i) {                                              /* uvg_cabac_write_unary_max_symbol_ep (cabac.c:388-413) */
    unsigned symbol = (unsigned)abs(info[7 + i + off]);
    const int code_last = (unsigned)max_off > symbol;
    ORC_FN(cabac_sim_ep)(symbol ? 1 : 0);
    if (!symbol) continue;
    while (--symbol) ORC_FN(cabac_sim_ep)(1);
    if (code_last) ORC_FN(cabac_sim_ep)(0);
  }
  if (type == 1) {
    for (int i = 1; i <= 4; ++i) if (info[7 + i + off] != 0) ORC_FN(cabac_sim_ep)(info[7 + i + off] < 0 ? 1 : 0);
    ORC_FN(cabac_sim_eps)((uint32_t)info[5 + (color == 2 ? 1 : 0)], 5);
  } else if (color != 2) {
    ORC_FN(cabac_sim_eps)((uint32_t)info[1], 2);
  }
}

/* ---- the CTU-level ALF syntax: uvg_encode_alf_bits (alf.c:1365-1413) between a CTU's SAO syntax and its coding tree (encoderstate.c:880) ----
 * models: alf_ctb_flag [9] (component * 3 + enabled neighbours), the "APS, not a fixed set" flag, the chroma alternatives [2], the CC-ALF control [6] */
typedef struct { uint16_t s0[18], s1[18]; uint8_t rate[18]; } alf_models;
static void alf_models_init(alf_models *m, int qp)
{
  for (int i = 0; i < 18; ++i) {
    const int v = k_ctx_init_alf[2][i];
    const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
    int s = ((slope * (qp - 16)) >> 1) + offset;
    s = s < 1 ? 1 : (s > 127 ? 127 : s);
    m->s0[i] = (uint16_t)((s << 8) & 0x7fe0); m->s1[i] = (uint16_t)((s << 8) & 0x7ffe);
    m->rate[i] = k_ctx_init_alf[3][i];
  }
}
static void alf_bin(alf_models *m, int c, int bin)
{
  ORC_FN(cabac_sim_bin)((m->s0[c] + m->s1[c]) >> 8, bin);
  const int r0 = m->rate[c] >> 4, r1 = m->rate[c] & 15;
  m->s0[c] = (uint16_t)(m->s0[c] - ((m->s0[c] >> r0) & 0x7fe0u));
  m->s1[c] = (uint16_t)(m->s1[c] - ((m->s1[c] >> r1) & 0x7ffeu));
  if (bin) { m->s0[c] = (uint16_t)(m->s0[c] + ((0x7fffu >> r0) & 0x7fe0u)); m->s1[c] = (uint16_t)(m->s1[c] + ((0x7fffu >> r1) & 0x7ffeu)); }
}
static void trunc_bin_ep(unsigned symbol, unsigned max_value)                     /* uvg_cabac_encode_trunc_bin (cabac.c:203-229), max_value <= 256 */
{
  int thresh = 0;
  while ((2u << thresh) <= max_value) ++thresh;                                  /* uvg_tb_max: floor(log2(max_value)) */
  const unsigned val = 1u << thresh, b = max_value - val;
  if (symbol < val - b) ORC_FN(cabac_sim_eps)(symbol, thresh);
  else ORC_FN(cabac_sim_eps)(symbol + val - b, thresh + 1);
}
/* meta: the "alf" record's header (ctu_dump.c): [3] alf_type, [4..6] slice enable Y / Cb / Cr, [7] luma APSs, [17..18] CC-ALF on, [19..20] CC-ALF filters;
 * flags [7][n]: CTU enable Y / Cb / Cr, alternative Cb / Cr, CC-ALF control Cb / Cr; n_alts: num_alternatives_chroma of the slice's chroma APS */
static void encode_alf_ctu(alf_models *m, const int32_t *meta, const uint8_t *flags, const int16_t *set_idx, int n_alts, int k, int wc, int n)
{
  const int left = k % wc ? k - 1 : -1, above = k >= wc ? k - wc : -1;
  for (int c = 0; c < 3; ++c) {
    const uint8_t *en = flags + (size_t)c * n;
    if (meta[4 + c]) alf_bin(m, c * 3 + (left >= 0 && en[left]) + (above >= 0 && en[above]), en[k]);      /* code_alf_ctu_enable_flag (:1147) */
    if (c == 0) {
      if (en[k] && meta[4]) {                                                    /* code_alf_ctu_filter_index (:1209) */
        const unsigned set = (unsigned)set_idx[k], n_aps = (unsigned)meta[7];
        if (n_aps > 0) {
          alf_bin(m, 9, set >= 16);
          if (set >= 16) { if (n_aps > 1) trunc_bin_ep(set - 16, n_aps); }
          else trunc_bin_ep(set, 16);
        } else trunc_bin_ep(set, 16);
      }
    } else if (meta[4 + c] && en[k]) {                                           /* code_alf_ctu_alternative_ctu (:1255) */
      const int ones = flags[(size_t)(2 + c) * n + k];
      for (int i = 0; i < ones; ++i) alf_bin(m, 10 + c - 1, 1);
      if (ones < n_alts - 1) alf_bin(m, 10 + c - 1, 0);
    }
  }
  if (meta[3] == 2)
    for (int c = 0; c < 2; ++c) {
      if (!meta[17 + c]) continue;                                                /* code_cc_alf_filter_control_idc (:1321) */
      const uint8_t *ctl = flags + (size_t)(5 + c) * n;
      const int idc = ctl[k], count = meta[19 + c];
      alf_bin(m, 12 + (left >= 0 && ctl[left]) + (above >= 0 && ctl[above]) + 3 * c, idc != 0);
      if (idc > 0) {
        for (int v = idc - 1; v > 0; --v) ORC_FN(cabac_sim_ep)(1);
        if (idc < count) ORC_FN(cabac_sim_ep)(0);
      }
    }
}

/*
 * sao: per CTU the two sao_info_t (34 ints; NULL = SAO off).  Out: every row's substream, emulation prevention applied
 * (uvg_bitstream_put_byte, bitstream.c:215-226), rows concatenated, row_off[rows + 1]; after[ctu] (optional): the models when the
 * CTU is coded.  The slice data of the picture is exactly these bytes.
 */
static long encode_rows_impl(const orc_search_params *p, const uint8_t *cu, const int16_t *coeff, const int32_t *sao, const int32_t *alf_meta, const uint8_t *alf_flags,
                             const int16_t *alf_set_idx, int alf_n_alts, uint8_t *bytes_out, long bytes_cap, int64_t *row_off, orc_models_ext *after)
{
  fbits_init();
  const int W = p->pic_w, H = p->pic_h, wc = (W + 63) / 64, hc = (H + 63) / 64, cu_stride = wc * 16;
  s_frame f = {p, (s_cu *)calloc((size_t)cu_stride * hc * 16, sizeof(s_cu)), cu_stride};
  for (int j = 0; j < hc * 16; ++j)
    for (int i = 0; i < cu_stride; ++i) {
      s_cu *c = &f.cua[j * cu_stride + i];
      const uint8_t *o = &cu[((size_t)j * cu_stride + i) * 20];
      c->type = o[0]; c->log2_w = o[1]; c->log2_h = o[2]; c->log2_cw = o[3]; c->log2_ch = o[4]; c->cbf = o[5]; c->mode = (int8_t)o[6];
      c->mode_chroma = (int8_t)o[7]; c->luma_deblocking = o[8]; c->chroma_deblocking = o[9]; c->qp = o[10];
      memcpy(&c->split_tree, o + 12, 4); memcpy(&c->mode_type_tree, o + 16, 4);
    }
  s_state *st = (s_state *)calloc(1, sizeof(s_state));
  st->p = p;
  orc_cabac_sim *sim = &ORC_FN(cabac_sim);
  orc_models *row_m = (orc_models *)calloc((size_t)hc, sizeof(orc_models));
  sao_models2 *row_s = (sao_models2 *)calloc((size_t)hc, sizeof(sao_models2));
  alf_models *row_a = (alf_models *)calloc((size_t)hc, sizeof(alf_models));
  long total = 0;
  row_off[0] = 0;
  for (int cy = 0; cy < hc; ++cy) {
    s_cabac cb;
    sao_models2 sm;
    alf_models am;
    if (cy == 0) { models_init(&cb.m, p->qp, 2); sao_models2_init(&sm, p->qp); alf_models_init(&am, p->qp); }
    else { cb.m = row_m[cy - 1]; sm = row_s[cy - 1]; am = row_a[cy - 1]; }
    cb.update = 1;
    sim->on = 2; sim->shifts = 0; sim->regular_fbits = 0.0;
    ORC_FN(cabac_sim_start)();
    for (int cx = 0; cx < wc; ++cx) {
      const int k = cy * wc + cx;
      if (sao) {                                                              /* encode_sao (encoderstate.c:593-608) */
        const int32_t *l = sao + (size_t)k * 34, *c = l + 17;
        if (cx > 0) sao_bin(&sm, 0, l[3]);
        if (cy > 0 && !l[3]) sao_bin(&sm, 0, l[4]);
        if (!l[3] && !l[4]) { encode_sao_color(&sm, l, 0); encode_sao_color(&sm, c, 1); encode_sao_color(&sm, c, 2); }
      }
      if (alf_meta) encode_alf_ctu(&am, alf_meta, alf_flags, alf_set_idx, alf_n_alts, k, wc, wc * hc);
      const int16_t *co = &coeff[(size_t)k * 6144];
      s_loc start_loc;
      loc_ctor(&start_loc, cx * 64, cy * 64, 64, 64);
      s_tree tree = {0, MODE_TYPE_ALL, 0, 0, 0, 0};
      g_tree_bits = 0.0;
      encode_coding_tree(&f, st, &cb, co, co + 4096, co + 4096 + 1024, &start_loc, &start_loc, tree, 1);
      if (after) models_to_ext(&cb.m, &after[k], NULL);
      if (cx == 0) { row_m[cy] = cb.m; row_s[cy] = sm; row_a[cy] = am; }                      /* the next row's start (encoderstate.c:966-975) */
    }
    ORC_FN(cabac_sim_row_end)();
    sim->on = 0;
    int zeros = 0;                                                            /* emulation prevention */
    for (size_t i = 0; i < sim->out_len; ++i) {
      const uint8_t b = sim->out[i];
      if (zeros == 2 && b < 4) { if (total >= bytes_cap) goto fail; bytes_out[total++] = 3; zeros = 0; }
      zeros = b == 0 ? zeros + 1 : 0;
      if (total >= bytes_cap) goto fail;
      bytes_out[total++] = b;
    }
    row_off[cy + 1] = total;
  }
  free(f.cua); free(st); free(row_m); free(row_s); free(row_a);
  return total;
fail:
  free(f.cua); free(st); free(row_m); free(row_s); free(row_a);
  return -1;
}
ORC_EXPORT long ORC_FN(encode_picture_rows)(const orc_search_params *p, const uint8_t *cu, const int16_t *coeff, const int32_t *sao,
                                            uint8_t *bytes_out, long bytes_cap, int64_t *row_off, orc_models_ext *after)
{
  return encode_rows_impl(p, cu, coeff, sao, NULL, NULL, NULL, 0, bytes_out, bytes_cap, row_off, after);
}
/* ... of a picture of an --alf full / --alf on run: the CTU-level ALF syntax between SAO and the coding tree (see encode_alf_ctu for the arguments) */
ORC_EXPORT long ORC_FN(encode_picture_rows_alf)(const orc_search_params *p, const uint8_t *cu, const int16_t *coeff, const int32_t *sao, const int32_t *alf_meta,
                                                const uint8_t *alf_flags, const int16_t *alf_set_idx, int alf_n_alts, uint8_t *bytes_out, long bytes_cap, int64_t *row_off)
{
  return encode_rows_impl(p, cu, coeff, sao, alf_meta, alf_flags, alf_set_idx, alf_n_alts, bytes_out, bytes_cap, row_off, NULL);
}

/*
 * The motion search of one reference picture for one prediction unit, as a function of its inputs: what search_pu_inter_ref does between
 * uvg_inter_get_mv_cand and the unit_stats_map entries (select_starting_point, early_terminate, hexagon_search; src/search_inter.c:1404-1446)
 * followed by search_frac and select_mv_cand as search_pu_inter applies them to the best unit of a list (:1908-1960).
 * job (24 ints): x, y, size, ref, mv_cand[2][2], extra_mv[2], n_start, start[6][2], (pad);  refs: the reference luma planes (pic_w x pic_h, tight).
 * out_i: mv[2], int_mv[2], mv_cand, skipped_hexagon;  out_d: cost, bits, int_cost, int_bits.
 */
ORC_EXPORT void ORC_FN(me_search_job)(int pic_w, int pic_h, double lambda_sqrt, int fme_level, const orc_px *src_y, const orc_px *const *refs, int n_refs,
                                      const int32_t *job, int32_t *out_i, double *out_d)
{
  orc_search_params p;
  memset(&p, 0, sizeof p);
  p.pic_w = pic_w; p.pic_h = pic_h; p.lambda_sqrt = lambda_sqrt;
  orc_inter_frame fr;
  memset(&fr, 0, sizeof fr);
  fr.fme_level = fme_level;
  fr.n_refs = n_refs;
  for (int i = 0; i < n_refs && i < 16; ++i) fr.ref_y[i] = refs[i];
  s_state st;
  memset(&st, 0, sizeof st);
  st.p = &p; st.fr = &fr; st.src_y = src_y;
  s_info info;
  memset(&info, 0, sizeof info);
  info.st = &st;
  info.origin.x = job[0]; info.origin.y = job[1]; info.width = info.height = job[2];
  info.ref_idx = job[3];
  info.mv_cand[0][0] = job[4]; info.mv_cand[0][1] = job[5]; info.mv_cand[1][0] = job[6]; info.mv_cand[1][1] = job[7];
  info.num_merge_cand = job[10] < 0 ? 0 : job[10];
  for (int i = 0; i < job[10]; ++i) { info.merge_cand[i].dir = 1; info.merge_cand[i].mv[0][0] = job[11 + 2 * i]; info.merge_cand[i].mv[0][1] = job[12 + 2 * i]; }
  s_vec best_mv = {job[8], job[9]};
  double best_cost = MAX_DOUBLE, best_bits = 2147483647;
  int skip_me = 0;
  if (job[10] >= 0) {
    select_starting_point(&info, best_mv, &best_cost, &best_bits, &best_mv);
    skip_me = early_terminate(&info, &best_cost, &best_bits, &best_mv);
    if (!skip_me) hexagon_search(&info, 0xffffffffu, &best_cost, &best_bits, &best_mv);
  } else { best_cost = 0; best_bits = 0; }            /* search_frac alone around best_mv = extra_mv */
  out_i[2] = best_mv.x; out_i[3] = best_mv.y; out_d[2] = best_cost; out_d[3] = best_bits;
  if (fme_level > 0) search_frac(&info, &best_cost, &best_bits, &best_mv);
  out_i[0] = best_mv.x; out_i[1] = best_mv.y; out_d[0] = best_cost; out_d[1] = best_bits;
  out_i[4] = select_mv_cand(info.mv_cand, best_mv.x, best_mv.y, NULL);
  out_i[5] = skip_me;
}

/*
 * The luma prediction of one motion and its SATD against the source: uvg_inter_pred_pu (predict_luma only) + uvg_satd_any_size as the merge
 * analysis and the bi-prediction test use them (src/search_inter.c:1758-1775, 2018-2031).  motion (9 ints): x, y, dir, ref[2] (picture
 * indices), mv[2][2];  pred_out: size x size samples (may be NULL).
 */
ORC_EXPORT unsigned ORC_FN(inter_pred_satd)(int pic_w, int pic_h, const orc_px *src_y, const orc_px *const *refs, int n_refs, const int32_t *motion, int size,
                                            orc_px *pred_out)
{
  orc_search_params p;
  memset(&p, 0, sizeof p);
  p.pic_w = pic_w; p.pic_h = pic_h;
  orc_inter_frame fr;
  memset(&fr, 0, sizeof fr);
  fr.n_refs = n_refs;
  for (int i = 0; i < n_refs && i < 16; ++i) fr.ref_y[i] = refs[i];
  s_state st;
  memset(&st, 0, sizeof st);
  st.p = &p; st.fr = &fr; st.src_y = src_y;
  s_loc loc;
  loc_ctor(&loc, motion[0], motion[1], size, size);
  static __thread orc_px pred[64 * 64];
  static __thread int16_t b0[64 * 64], b1[64 * 64];
  const int dir = motion[2];
  int32_t mv[2][2] = {{motion[5], motion[6]}, {motion[7], motion[8]}};
  if (dir == 3) {
    const unsigned f0 = recon_unipred(&st, motion[3], mv[0], &loc, 1, 1, 0, b0, NULL, NULL);
    const unsigned f1 = recon_unipred(&st, motion[4], mv[1], &loc, 1, 1, 0, b1, NULL, NULL);
    ORC_FN(bipred_average)(pred, size, b0, b1, (int)((f0 & 1) | ((f1 & 1) << 1)), size, size);
  } else {
    recon_unipred(&st, motion[3 + dir - 1], mv[dir - 1], &loc, 0, 1, 0, pred, NULL, NULL);
  }
  if (pred_out) memcpy(pred_out, pred, (size_t)size * size * sizeof(orc_px));
  return ORC_FN(satd_any_size)(size, size, src_y + (size_t)motion[1] * pic_w + motion[0], pic_w, pred, size);
}
