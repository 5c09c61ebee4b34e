/*
 * ORACLE (test infrastructure only; never linked into the product): merge candidates of an inter CU, a plain-C restatement of
 *   uvg_inter_get_merge_cand            src/inter.c:1989-2192
 *   get_spatial_merge_candidates        src/inter.c:1368-1455  (A0, A1, B0, B1, B2 with the coding-order tests is_cand_coded :770-876)
 *   get_temporal_merge_candidates       src/inter.c:1031-1097  (C0 below-right, C1 centre, on the collocated picture's 8x8 grid)
 *   add_temporal_candidate              src/inter.c:1547-1601  with apply_mv_scaling_pocs :1146-1165, get_scaled_mv :1099-1103 and the
 *                                       vector compression round_mv_comp :1111-1144
 *   add_merge_candidate / is_duplicate_candidate / different_mer / round_avg_mv  :1782-1830, 1908-1925
 *   inter_clear_cu_unused               src/inter.c:749-758    (the neighbours looked at lose their unused list's vector / index)
 * Pinned by records of the real encoder's calls (tools/refcheck/ctu_dump.c wraps uvg_inter_get_merge_cand and stores everything
 * it reads: tests/golden/ref_merge_*.npz, tests/test_oracle_inter_cand.py).  Multi-type splits are off in the configurations of this
 * project (quad tree only), so is_cand_coded only meets QT_SPLIT.  The functions do not depend on the bit depth; the file is built
 * for both like every oracle file.
 */
#include "orc_common.h"
#include <string.h>

typedef struct { int32_t type, mv[2][2], ref[2], dir; } cand_cu;          /* the fields of cu_info_t this path reads (8 ints) */
typedef struct { int32_t dir, ref[2], mv[2][2]; } merge_cand;             /* inter_merge_cand_t (7 ints) */
enum { CU_INTER_T = 2, TCW = 17, LCU_W = 64 };

static cand_cu *lcu_at(cand_cu *tab, int x_px, int y_px) { return &tab[TCW + 1 + (x_px >> 2) + (y_px >> 2) * TCW]; }   /* LCU_GET_CU_AT_PX */

static void clear_unused(cand_cu *c)
{
  for (int l = 0; l < 2; ++l)
    if (!(c->dir & (1 << l))) { c->mv[l][0] = 0; c->mv[l][1] = 0; c->ref[l] = 255; }
}

/* is_cand_coded for quad-tree splits: the first level at which the two positions fall into different quadrants decides */
static int cand_coded(int cur_x, int cur_y, int cand_x, int cand_y, uint32_t split_tree)
{
  int l2 = 6;
  if ((cur_y >> l2) != (cand_y >> l2)) return (cand_y >> l2) < (cur_y >> l2);
  if ((cur_x >> l2) != (cand_x >> l2)) return (cand_x >> l2) < (cur_x >> l2);
  for (int depth = 0; depth < 8; ++depth) {
    const uint32_t split = (split_tree >> (depth * 3)) & 7;
    if (split != 1) return 0;            /* (the reference asserts: the two positions are in the same block) */
    --l2;
    const int cur = ((cur_x >> l2) & 1) + 2 * ((cur_y >> l2) & 1), cand = ((cand_x >> l2) & 1) + 2 * ((cand_y >> l2) & 1);
    if (cand != cur) return cand < cur;
  }
  return 0;
}

static int duplicate(const cand_cu *a, const cand_cu *b)
{
  if (!b) return 0;
  if (a->dir != b->dir) return 0;
  for (int l = 0; l < 2; ++l)
    if ((a->dir & (1 << l)) && (a->mv[l][0] != b->mv[l][0] || a->mv[l][1] != b->mv[l][1] || a->ref[l] != b->ref[l])) return 0;
  return 1;
}

static int add_merge(const cand_cu *c, const cand_cu *d1, const cand_cu *d2, merge_cand *out)
{
  if (!c || duplicate(c, d1) || duplicate(c, d2)) return 0;
  out->mv[0][0] = c->mv[0][0]; out->mv[0][1] = c->mv[0][1]; out->mv[1][0] = c->mv[1][0]; out->mv[1][1] = c->mv[1][1];
  out->ref[0] = c->ref[0] & 255; out->ref[1] = c->ref[1] & 255; out->dir = c->dir;
  return 1;
}

static int different_mer(int x, int y, int x2, int y2, int level) { return (x >> level) != (x2 >> level) || (y >> level) != (y2 >> level); }

static int floor_log2(uint32_t v) { int r = 0; while (v >>= 1) ++r; return r; }
static int round_mv_comp(int32_t val)          /* 6-bit mantissa / 4-bit exponent round trip of the stored temporal vectors */
{
  const uint32_t sign = (uint32_t)(val >> 31);
  const int scale = floor_log2(((uint32_t)val ^ sign) | 31u) - 5;
  int exponent;
  uint32_t mantissa;
  if (scale >= 0) {
    const int round = (1 << scale) >> 1;
    const int n = (val + round) >> scale;
    exponent = scale + (int)(((uint32_t)n ^ sign) >> 5);
    mantissa = ((uint32_t)n & 31u) | (sign << 5);
  } else {
    exponent = 0;
    mantissa = (uint32_t)val;
  }
  const int packed = exponent | (int)(mantissa << 4);
  const int e = packed & 15;
  const uint32_t m = (uint32_t)(packed >> 4);
  return e == 0 ? (int)m : (int)((m ^ 32u) << (e - 1));
}

static int32_t scaled_mv(int32_t mv, int scale)
{
  const int32_t s = scale * mv;
  return orc_clip3(-131072, 131071, (s + 127 + (s < 0)) >> 8);
}
static void scale_pocs(int cur_poc, int cur_ref_poc, int nb_poc, int nb_ref_poc, int32_t mv[2])
{
  int dc = cur_poc - cur_ref_poc, dn = nb_poc - nb_ref_poc;
  if (dc == dn) return;
  dc = orc_clip3(-128, 127, dc);
  dn = orc_clip3(-128, 127, dn);
  const int scale = orc_clip3(-4096, 4095, (dc * ((0x4000 + (orc_iabs(dn) >> 1)) / dn) + 32) >> 6);
  mv[0] = scaled_mv(mv[0], scale);
  mv[1] = scaled_mv(mv[1], scale);
}

/* get_spatial_merge_candidates (src/inter.c:1368-1455) */
static void spatial_candidates(cand_cu *tab, int x, int y, int w, int h, int pic_w, int pic_h, int wpp, uint32_t split_tree,
                               cand_cu **a0_, cand_cu **a1_, cand_cu **b0_, cand_cu **b1_, cand_cu **b2_)
{
  const int lx = x & 63, ly = y & 63;
  cand_cu *a0 = NULL, *a1 = NULL, *b0 = NULL, *b1 = NULL, *b2 = NULL;
  if (x != 0) {
    cand_cu *c = lcu_at(tab, lx - 1, ly + h - 1);
    if (c->type == CU_INTER_T) { clear_unused(c); a1 = c; }
    if (ly + h < LCU_W && y + h < pic_h) {
      c = lcu_at(tab, lx - 1, ly + h);
      if (c->type == CU_INTER_T && cand_coded(x, y, x - 1, y + h, split_tree)) { clear_unused(c); a0 = c; }
    }
  }
  if (y != 0) {
    cand_cu *c = NULL;
    if (x + w < pic_w) {
      if (lx + w < LCU_W) c = lcu_at(tab, lx + w, ly - 1);
      else if (!wpp && ly == 0) c = &tab[TCW * TCW];              /* LCU_GET_TOP_RIGHT_CU */
    }
    if (c && c->type == CU_INTER_T && cand_coded(x, y, x + w, y - 1, split_tree)) { clear_unused(c); b0 = c; }
    c = lcu_at(tab, lx + w - 1, ly - 1);
    if (c->type == CU_INTER_T) { clear_unused(c); b1 = c; }
    if (x != 0) {
      c = lcu_at(tab, lx - 1, ly - 1);
      if (c->type == CU_INTER_T) { clear_unused(c); b2 = c; }
    }
  }
  *a0_ = a0; *a1_ = a1; *b0_ = b0; *b1_ = b1; *b2_ = b2;
}

/* get_temporal_merge_candidates (src/inter.c:1031-1097) for list L0, index 0: C0 if there, else C1 */
static const int32_t *temporal_cu(const int32_t *col, int x, int y, int w, int h, int pic_w, int pic_h, int l0_size)
{
  const int gw = (pic_w + 7) / 8;
  const int32_t *c0 = NULL, *c1 = NULL;
  if (l0_size <= 0) return NULL;
  const int xbr = x + w, ybr = y + h;
  if (xbr < pic_w && ybr < pic_h && (ybr % LCU_W) != 0) {
    const int32_t *c = col + ((size_t)(ybr >> 3) * gw + (xbr >> 3)) * 8;
    if (c[0] == CU_INTER_T) c0 = c;
  }
  const int xc = x + w / 2, yc = y + h / 2;
  if (xc < pic_w && yc < pic_h) {
    const int32_t *c = col + ((size_t)(yc >> 3) * gw + (xc >> 3)) * 8;
    if (c[0] == CU_INTER_T) c1 = c;
  }
  return c0 ? c0 : c1;
}

/* add_temporal_candidate (src/inter.c:1547-1601): the collocated vector, compressed, scaled from its POC distance to the current one */
static int temporal_vector(const int32_t *tc, int reflist, int poc, const int32_t *pocs, int used, int cur_ref_poc, int col_poc, int32_t mv[2])
{
  if (!tc) return 0;
  int col_list = reflist;
  for (int i = 0; i < used; ++i) if (pocs[i] > poc) { col_list = 1; break; }
  if ((tc[5] & (col_list + 1)) == 0) col_list = 1 - col_list;
  mv[0] = round_mv_comp(tc[1 + 2 * col_list]);
  mv[1] = round_mv_comp(tc[2 + 2 * col_list]);
  scale_pocs(poc, cur_ref_poc, col_poc, tc[6 + col_list], mv);
  return 1;
}

/*
 * ctx: the 64 ints of a "merge" record (tools/refcheck/ctu_dump.c): [1..4] x, y, width, height of the CU; [5] POC; [6] slice type
 * (0 = B); [7..8] picture size; [9] tmvp; [10] max merge candidates; [11] log2 parallel merge level; [12] wpp; [13] references in
 * use, [14..29] their POCs; [30..31] list sizes, [32..39] / [40..47] L0 / L1 (indices into the POC array); [49] the CU's split tree.
 * lcu: 17 * 17 + 1 entries of 8 ints (the lcu_t's table at the moment of the call; MODIFIED like the reference does);
 * col: the collocated picture (L0[0]) on the 8x8 grid, 8 ints per position: type, mv[2][2], dir, the POC each list's vector points to;
 * hmvp: [0] entries in the row's table, then 5 entries of 8 ints.  out: 6 candidates of 7 ints; returns their number.
 */
ORC_EXPORT int ORC_FN(merge_candidates)(const int32_t *ctx, int32_t *lcu, const int32_t *col, const int32_t *hmvp, int32_t *out)
{
  cand_cu *tab = (cand_cu *)lcu;
  merge_cand *mc = (merge_cand *)out;
  const int x = ctx[1], y = ctx[2], w = ctx[3], h = ctx[4], poc = ctx[5], is_b = ctx[6] == 0, pic_w = ctx[7], pic_h = ctx[8];
  const int tmvp = ctx[9], max_cands = ctx[10], mer = ctx[11], wpp = ctx[12], used = ctx[13];
  const int32_t *pocs = ctx + 14, *lsize = ctx + 30, *L[2] = {ctx + 32, ctx + 40};
  const uint32_t split_tree = (uint32_t)ctx[49];
  memset(mc, 0, 6 * sizeof *mc);
  /* ---- spatial ---- */
  cand_cu *a0, *a1, *b0, *b1, *b2;
  spatial_candidates(tab, x, y, w, h, pic_w, pic_h, wpp, split_tree, &a0, &a1, &b0, &b1, &b2);
  int n = 0;
  if (different_mer(x, y, x, y - 1, mer) && add_merge(b1, NULL, NULL, &mc[n])) n++;
  if (different_mer(x, y, x - 1, y, mer) && add_merge(a1, b1, NULL, &mc[n])) n++;
  if (different_mer(x, y, x + 1, y - 1, mer) && add_merge(b0, b1, NULL, &mc[n])) n++;
  if (different_mer(x, y, x - 1, y + 1, mer) && add_merge(a0, a1, NULL, &mc[n])) n++;
  if (n < 4 && different_mer(x, y, x - 1, y - 1, mer) && add_merge(b2, a1, b1, &mc[n])) n++;
  /* ---- temporal ---- */
  if (tmvp && n < max_cands && used) {
    mc[n].dir = 0;
    const int32_t *tc = temporal_cu(col, x, y, w, h, pic_w, pic_h, lsize[0]);
    for (int reflist = 0; reflist <= (is_b ? 1 : 0); ++reflist) {
      int32_t mv[2];
      /* current reference: index 0 of L0 for either list (sic, :2041-2048); the collocated picture is L0[0] */
      if (lsize[0] <= 0 || !temporal_vector(tc, reflist, poc, pocs, used, pocs[L[0][0]], pocs[L[0][0]], mv)) continue;
      mc[n].mv[reflist][0] = mv[0]; mc[n].mv[reflist][1] = mv[1];
      mc[n].ref[reflist] = 0;
      mc[n].dir |= 1 << reflist;
      if (pocs[L[reflist][0]] > poc) { mc[n].mv[reflist][0] *= -1; mc[n].mv[reflist][1] *= -1; }
    }
    if (mc[n].dir != 0) n++;
  }
  if (n >= max_cands) return n;
  /* ---- history ---- */
  if (n < max_cands - 1) {
    const cand_cu *lut = (const cand_cu *)(hmvp + 1);
    for (int i = 0; i < hmvp[0]; ++i) {
      if (i > 1 || (!duplicate(&lut[i], a1) && !duplicate(&lut[i], b1))) {
        mc[n].mv[0][0] = lut[i].mv[0][0]; mc[n].mv[0][1] = lut[i].mv[0][1];
        mc[n].dir = lut[i].dir;
        mc[n].ref[0] = lut[i].ref[0] & 255;
        if (is_b) { mc[n].mv[1][0] = lut[i].mv[1][0]; mc[n].mv[1][1] = lut[i].mv[1][1]; mc[n].ref[1] = lut[i].ref[1] & 255; }
        n++;
        if (n == max_cands - 1) break;
      }
    }
  }
  /* ---- pairwise average of the first two ---- */
  if (n > 1 && n < max_cands) {
    int inter_dir = 0;
    for (int l = 0; l < (is_b ? 2 : 1); ++l) {
      const int ri = (mc[0].dir & (l + 1)) ? mc[0].ref[l] : -1, rj = (mc[1].dir & (l + 1)) ? mc[1].ref[l] : -1;
      if (ri == -1 && rj == -1) continue;
      inter_dir += 1 << l;
      if (ri != -1 && rj != -1) {
        int32_t ax = mc[0].mv[l][0] + mc[1].mv[l][0], ay = mc[0].mv[l][1] + mc[1].mv[l][1];
        ax = (ax + 1 - (ax >= 0)) >> 1; ay = (ay + 1 - (ay >= 0)) >> 1;          /* round_avg_mv, shift 1 */
        mc[n].mv[l][0] = ax; mc[n].mv[l][1] = ay; mc[n].ref[l] = ri & 255;
      } else if (ri != -1) { mc[n].mv[l][0] = mc[0].mv[l][0]; mc[n].mv[l][1] = mc[0].mv[l][1]; mc[n].ref[l] = ri & 255; }
      else { mc[n].mv[l][0] = mc[1].mv[l][0]; mc[n].mv[l][1] = mc[1].mv[l][1]; mc[n].ref[l] = rj & 255; }
    }
    mc[n].dir = inter_dir;
    if (inter_dir > 0) n++;
  }
  if (n >= max_cands) return n;
  /* ---- zero vectors ---- */
  int num_ref = used;
  if (n < max_cands && is_b) {
    int neg = 0, pos = 0;
    for (int j = 0; j < used; ++j) { if (pocs[j] < poc) neg++; else pos++; }
    num_ref = neg < pos ? neg : pos;
  }
  int zero_idx = 0;
  while (n < max_cands) {
    mc[n].mv[0][0] = 0; mc[n].mv[0][1] = 0;
    mc[n].ref[0] = (zero_idx >= num_ref - 1) ? 0 : zero_idx;
    mc[n].dir = 1;
    if (is_b) { mc[n].ref[1] = mc[n].ref[0]; mc[n].mv[1][0] = 0; mc[n].mv[1][1] = 0; mc[n].dir = 3; }
    zero_idx++;
    n++;
  }
  return n;
}

/* add_mvp_candidate without scaling (src/inter.c:1185-1219): a neighbour's vector that points to the picture being searched */
static int add_mvp(const cand_cu *c, int reflist, int target, const int32_t *const L[2], int32_t mv[2])
{
  if (!c) return 0;
  for (int i = 0; i < 2; ++i) {
    const int cl = i == 0 ? reflist : !reflist;
    if (!(c->dir & (1 << cl))) continue;
    if (L[cl][c->ref[cl]] == target) { mv[0] = c->mv[cl][0]; mv[1] = c->mv[cl][1]; return 1; }
  }
  return 0;
}
static int32_t round_quarter(int32_t v)         /* uvg_round_precision(INTERNAL_MV_PREC, 2): to quarter samples and back */
{
  v = v >= 0 ? (v + 1) >> 2 : (v + 2) >> 2;
  return (int32_t)((uint32_t)v << 2);
}

/*
 * uvg_inter_get_mv_cand (src/inter.c:1606-1737): the two AMVP predictors for list ctx[50] and the reference index ctx[51 + list] being
 * searched.  Same inputs as merge_candidates; out: mv_cand[2][2].
 */
ORC_EXPORT void ORC_FN(amvp_candidates)(const int32_t *ctx, int32_t *lcu, const int32_t *col, const int32_t *hmvp, int32_t *out)
{
  cand_cu *tab = (cand_cu *)lcu;
  const int x = ctx[1], y = ctx[2], w = ctx[3], h = ctx[4], poc = ctx[5], pic_w = ctx[7], pic_h = ctx[8];
  const int tmvp = ctx[9], wpp = ctx[12], used = ctx[13], reflist = ctx[50];
  const int32_t *pocs = ctx + 14, *lsize = ctx + 30;
  const int32_t *const L[2] = {ctx + 32, ctx + 40};
  const int target = L[reflist][ctx[51 + reflist]];            /* ref_LX[reflist][cur_cu->inter.mv_ref[reflist]] */
  cand_cu *a0, *a1, *b0, *b1, *b2;
  spatial_candidates(tab, x, y, w, h, pic_w, pic_h, wpp, (uint32_t)ctx[49], &a0, &a1, &b0, &b1, &b2);
  const int32_t *tc = temporal_cu(col, x, y, w, h, pic_w, pic_h, used ? lsize[0] : 0);
  int32_t mv[2][2] = {{0, 0}, {0, 0}};
  int n = 0, nb = 0;
  if (add_mvp(a0, reflist, target, L, mv[n])) n++;
  else if (add_mvp(a1, reflist, target, L, mv[n])) n++;
  if (add_mvp(b0, reflist, target, L, mv[n])) nb++;
  else if (add_mvp(b1, reflist, target, L, mv[n])) nb++;
  else if (add_mvp(b2, reflist, target, L, mv[n])) nb++;
  n += nb;
  if (n > 0) { mv[0][0] = round_quarter(mv[0][0]); mv[0][1] = round_quarter(mv[0][1]); }
  if (n > 1) { mv[1][0] = round_quarter(mv[1][0]); mv[1][1] = round_quarter(mv[1][1]); }
  if (n == 2 && mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1]) n = 1;
  if (tmvp && poc > 1 && used && n < 2 && tc && lsize[0] > 0 && temporal_vector(tc, reflist, poc, pocs, used, pocs[target], pocs[L[0][0]], mv[n])) n++;
  if (n < 2) {
    const cand_cu *lut = (const cand_cu *)(hmvp + 1);
    const int num = hmvp[0];
    for (int i = 0; i < (num < 4 ? num : 4) && n < 2; ++i)
      for (int ps = 0; ps < 2 && n < 2; ++ps) {
        const int cl = ps == 0 ? reflist : !reflist;
        const cand_cu *c = &lut[num - 1 - i];
        if (!(c->dir & (1 << cl))) continue;
        if (L[cl][c->ref[cl]] == target) { mv[n][0] = c->mv[cl][0]; mv[n][1] = c->mv[cl][1]; n++; }
      }
  }
  while (n < 2) { mv[n][0] = 0; mv[n][1] = 0; n++; }
  out[0] = round_quarter(mv[0][0]); out[1] = round_quarter(mv[0][1]); out[2] = round_quarter(mv[1][0]); out[3] = round_quarter(mv[1][1]);
}
