// Merge / AMVP candidate derivation on the device: uvg_inter_get_merge_cand (src/inter.c:1989-2192) and uvg_inter_get_mv_cand
// (src/inter.c:1606-1737) with what they call -- the spatial neighbours A0 / A1 / B0 / B1 / B2 with the coding-order test
// (get_spatial_merge_candidates :1368-1455, is_cand_coded :770-876), the temporal candidate from the collocated picture's 8x8 grid with
// POC scaling and the 10-bit storage round trip (:1031-1165, 1547-1601), the history table, the pairwise average, the zero vectors.
//
// One lane derives one list: the work per call is a few hundred dependent integer operations on 5 + 2 table entries, so the
// parallelism is across calls (the batch entry points of inter_cand.hip: one lane per call; the CTU kernel: the lanes of a wave over
// the CUs of a depth).  The tables are read through an accessor so that the same code runs on a table in global memory (batch) or in
// the workgroup's LDS image (CTU kernel).
#pragma once
#include <stdint.h>

namespace icand {

// the fields of cu_info_t the derivation reads, per 4x4 unit (8 ints: what the batch ABI passes; the CTU kernel packs the same)
struct unit { int32_t type, mv[2][2], ref[2], dir; };
struct merge_cand { int32_t dir, ref[2], mv[2][2]; };      // inter_merge_cand_t
enum { TYPE_INTER = 2, TCW = 17 };

// the picture-level inputs of one call
struct frame_ctx {
  int32_t x, y, w, h;              // the CU
  int32_t poc, is_b, pic_w, pic_h;
  int32_t tmvp, max_cands, mer_level, wpp;
  int32_t n_refs;                  // pictures in the reference array
  int32_t ref_pocs[16];
  int32_t l_size[2];
  int32_t l[2][8];                 // ref_LX: list index -> position in the reference array
  uint32_t split_tree;             // the CU's split tree (the coding-order test of A0 / B0)
};

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- small arithmetic pieces ----
__device__ inline int floor_log2_u(uint32_t v) { return 31 - __clz((int)v); }
// the stored temporal vector: 6-bit mantissa, 4-bit exponent, and back (round_mv_comp, inter.c:1111-1144)
__device__ inline int32_t mv_storage_roundtrip(int32_t val)
{
  const uint32_t sign = (uint32_t)(val >> 31);
  const int scale = floor_log2_u(((uint32_t)val ^ sign) | 31u) - 5;
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
  return e == 0 ? (int32_t)m : (int32_t)((m ^ 32u) << (e - 1));
}
__device__ inline int32_t scale_one(int32_t mv, int scale)
{
  const int32_t s = scale * mv;
  return clip3(-131072, 131071, (s + 127 + (s < 0)) >> 8);
}
__device__ inline void scale_by_pocs(int cur_poc, int cur_ref_poc, int nb_poc, int nb_ref_poc, int32_t mv[2])
{
  int dc = cur_poc - cur_ref_poc, dn = nb_poc - nb_ref_poc;
  if (dc == dn) return;
  dc = clip3(-128, 127, dc);
  dn = clip3(-128, 127, dn);
  const int adn = dn < 0 ? -dn : dn;
  const int scale = clip3(-4096, 4095, (dc * ((0x4000 + (adn >> 1)) / dn) + 32) >> 6);
  mv[0] = scale_one(mv[0], scale);
  mv[1] = scale_one(mv[1], scale);
}
__device__ inline int32_t to_quarter_and_back(int32_t v)       // uvg_round_precision(INTERNAL_MV_PREC, 2)
{
  v = v >= 0 ? (v + 1) >> 2 : (v + 2) >> 2;
  return (int32_t)((uint32_t)v << 2);
}

// is_cand_coded for quad-tree splits: the first level at which the two positions fall into different quadrants decides
__device__ inline bool coded_before(int cur_x, int cur_y, int cand_x, int cand_y, uint32_t split_tree)
{
  int l2 = 6;
  if ((cur_y >> l2) != (cand_y >> l2)) return (cand_y >> l2) < (cur_y >> l2);
  if ((cur_x >> l2) != (cand_x >> l2)) return (cand_x >> l2) < (cur_x >> l2);
  for (int depth = 0; depth < 8; ++depth) {
    if (((split_tree >> (depth * 3)) & 7u) != 1u) return false;
    --l2;
    const int cur = ((cur_x >> l2) & 1) + 2 * ((cur_y >> l2) & 1), cand = ((cand_x >> l2) & 1) + 2 * ((cand_y >> l2) & 1);
    if (cand != cur) return cand < cur;
  }
  return false;
}

__device__ inline bool same_motion(const unit &a, const unit *b)
{
  if (!b) return false;
  if (a.dir != b->dir) return false;
  for (int l = 0; l < 2; ++l)
    if ((a.dir & (1 << l)) && (a.mv[l][0] != b->mv[l][0] || a.mv[l][1] != b->mv[l][1] || a.ref[l] != b->ref[l])) return false;
  return true;
}

// ---- the spatial neighbours ----
// TAB: unit &at(int index) over the lcu_t's 17 x 17 + 1 table (index = 18 + x4 + y4 * 17 for CTU-local 4x4 coordinates >= -1;
// the last entry is the CTU above right, filled only without WPP).  A neighbour that is looked at loses the vector and reference of
// the lists it does not use (inter_clear_cu_unused, inter.c:749-758): the table is modified like the reference's.
struct neighbours { unit *a0, *a1, *b0, *b1, *b2; };

template <typename TAB> __device__ inline unit *unit_at(TAB &tab, int lx, int ly) { return &tab.at(TCW + 1 + (lx >> 2) + (ly >> 2) * TCW); }
__device__ inline void drop_unused(unit *c)
{
  for (int l = 0; l < 2; ++l)
    if (!(c->dir & (1 << l))) { c->mv[l][0] = 0; c->mv[l][1] = 0; c->ref[l] = 255; }
}
template <typename TAB> __device__ inline neighbours spatial(TAB &tab, const frame_ctx &f)
{
  const int x = f.x, y = f.y, w = f.w, h = f.h, lx = x & 63, ly = y & 63;
  neighbours n = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (x != 0) {
    unit *c = unit_at(tab, lx - 1, ly + h - 1);
    if (c->type == TYPE_INTER) { drop_unused(c); n.a1 = c; }
    if (ly + h < 64 && y + h < f.pic_h) {
      c = unit_at(tab, lx - 1, ly + h);
      if (c->type == TYPE_INTER && coded_before(x, y, x - 1, y + h, f.split_tree)) { drop_unused(c); n.a0 = c; }
    }
  }
  if (y != 0) {
    unit *c = nullptr;
    if (x + w < f.pic_w) {
      if (lx + w < 64) c = unit_at(tab, lx + w, ly - 1);
      else if (!f.wpp && ly == 0) c = &tab.at(TCW * TCW);
    }
    if (c && c->type == TYPE_INTER && coded_before(x, y, x + w, y - 1, f.split_tree)) { drop_unused(c); n.b0 = c; }
    c = unit_at(tab, lx + w - 1, ly - 1);
    if (c->type == TYPE_INTER) { drop_unused(c); n.b1 = c; }
    if (x != 0) {
      c = unit_at(tab, lx - 1, ly - 1);
      if (c->type == TYPE_INTER) { drop_unused(c); n.b2 = c; }
    }
  }
  return n;
}

// ---- the temporal candidate ----
// COL: the collocated picture (L0[0]) on its 8x8 grid, 8 ints per position: type, mv[2][2], dir, the POC each list's vector points to
struct col_unit { int32_t type, mv[2][2], dir, poc[2]; };
template <typename COL> __device__ inline bool temporal_unit(COL &col, const frame_ctx &f, col_unit *out)
{
  if (f.l_size[0] <= 0) return false;
  const int gw = (f.pic_w + 7) / 8;
  bool have = false;
  const int xbr = f.x + f.w, ybr = f.y + f.h;
  if (xbr < f.pic_w && ybr < f.pic_h && (ybr % 64) != 0) {
    const col_unit c = col.at((ybr >> 3) * gw + (xbr >> 3));
    if (c.type == TYPE_INTER) { *out = c; have = true; }
  }
  if (!have) {
    const int xc = f.x + f.w / 2, yc = f.y + f.h / 2;
    if (xc < f.pic_w && yc < f.pic_h) {
      const col_unit c = col.at((yc >> 3) * gw + (xc >> 3));
      if (c.type == TYPE_INTER) { *out = c; have = true; }
    }
  }
  return have;
}
__device__ inline void temporal_vector(const col_unit &tc, int reflist, const frame_ctx &f, int cur_ref_poc, int col_poc, int32_t mv[2])
{
  int col_list = reflist;
  for (int i = 0; i < f.n_refs; ++i) if (f.ref_pocs[i] > f.poc) { col_list = 1; break; }
  if ((tc.dir & (col_list + 1)) == 0) col_list = 1 - col_list;
  mv[0] = mv_storage_roundtrip(tc.mv[col_list][0]);
  mv[1] = mv_storage_roundtrip(tc.mv[col_list][1]);
  scale_by_pocs(f.poc, cur_ref_poc, col_poc, tc.poc[col_list], mv);
}

__device__ inline bool other_mer(int x, int y, int x2, int y2, int level) { return (x >> level) != (x2 >> level) || (y >> level) != (y2 >> level); }
__device__ inline bool take_spatial(const unit *c, const unit *d1, const unit *d2, merge_cand *out)
{
  if (!c || same_motion(*c, d1) || same_motion(*c, d2)) return false;
  out->mv[0][0] = c->mv[0][0]; out->mv[0][1] = c->mv[0][1]; out->mv[1][0] = c->mv[1][0]; out->mv[1][1] = c->mv[1][1];
  out->ref[0] = c->ref[0] & 255; out->ref[1] = c->ref[1] & 255; out->dir = c->dir;
  return true;
}

// (max_cands = cfg.max_merge: the reference's own construction assumes at least 5 -- four spatial candidates are taken before the count is
// first compared with it, and its closing loops test for equality, inter.c:2028-2176; with fewer it overruns its array.  The comparisons
// here are ">=" / "<" so that a smaller value cannot run away; uvghip_ctu_search_pb refuses it.)
// uvg_inter_get_merge_cand.  hmvp: [0] entries in the CTU row's table, then 5 units (most recent first).  -> number of candidates
template <typename TAB, typename COL>
__device__ inline int merge_candidates(const frame_ctx &f, TAB &tab, COL &col, const int32_t *hmvp, merge_cand *mc)
{
  for (int i = 0; i < 6; ++i) { mc[i].dir = 0; mc[i].ref[0] = mc[i].ref[1] = 0; mc[i].mv[0][0] = mc[i].mv[0][1] = mc[i].mv[1][0] = mc[i].mv[1][1] = 0; }
  const neighbours nb = spatial(tab, f);
  const int x = f.x, y = f.y, mer = f.mer_level, max_cands = f.max_cands < 6 ? f.max_cands : 6;      // (mc[] has six entries whatever the caller's context says)
  int n = 0;
  if (other_mer(x, y, x, y - 1, mer) && take_spatial(nb.b1, nullptr, nullptr, &mc[n])) n++;
  if (other_mer(x, y, x - 1, y, mer) && take_spatial(nb.a1, nb.b1, nullptr, &mc[n])) n++;
  if (other_mer(x, y, x + 1, y - 1, mer) && take_spatial(nb.b0, nb.b1, nullptr, &mc[n])) n++;
  if (other_mer(x, y, x - 1, y + 1, mer) && take_spatial(nb.a0, nb.a1, nullptr, &mc[n])) n++;
  if (n < 4 && other_mer(x, y, x - 1, y - 1, mer) && take_spatial(nb.b2, nb.a1, nb.b1, &mc[n])) n++;
  if (f.tmvp && n < max_cands && f.n_refs) {
    mc[n].dir = 0;
    col_unit tc;
    const bool have = temporal_unit(col, f, &tc);
    for (int reflist = 0; reflist <= (f.is_b ? 1 : 0); ++reflist) {
      if (!have || f.l_size[0] <= 0) continue;
      int32_t mv[2];
      // the current reference is index 0 of L0 for either list (inter.c:2041-2048); the collocated picture is L0[0]
      temporal_vector(tc, reflist, f, f.ref_pocs[f.l[0][0]], f.ref_pocs[f.l[0][0]], mv);
      mc[n].mv[reflist][0] = mv[0]; mc[n].mv[reflist][1] = mv[1];
      mc[n].ref[reflist] = 0;
      mc[n].dir |= 1 << reflist;
      if (f.ref_pocs[f.l[reflist][0]] > f.poc) { mc[n].mv[reflist][0] *= -1; mc[n].mv[reflist][1] *= -1; }
    }
    if (mc[n].dir != 0) n++;
  }
  if (n >= max_cands) return n;
  if (n < max_cands - 1) {               // history
    const unit *lut = reinterpret_cast<const unit *>(hmvp + 1);
    for (int i = 0; i < hmvp[0]; ++i) {
      if (i > 1 || (!same_motion(lut[i], nb.a1) && !same_motion(lut[i], nb.b1))) {
        mc[n].mv[0][0] = lut[i].mv[0][0]; mc[n].mv[0][1] = lut[i].mv[0][1];
        mc[n].dir = lut[i].dir;
        mc[n].ref[0] = lut[i].ref[0] & 255;
        if (f.is_b) { mc[n].mv[1][0] = lut[i].mv[1][0]; mc[n].mv[1][1] = lut[i].mv[1][1]; mc[n].ref[1] = lut[i].ref[1] & 255; }
        n++;
        if (n == max_cands - 1) break;
      }
    }
  }
  if (n > 1 && n < max_cands) {           // the average of the first two
    int inter_dir = 0;
    for (int l = 0; l < (f.is_b ? 2 : 1); ++l) {
      const int ri = (mc[0].dir & (l + 1)) ? mc[0].ref[l] : -1, rj = (mc[1].dir & (l + 1)) ? mc[1].ref[l] : -1;
      if (ri == -1 && rj == -1) continue;
      inter_dir += 1 << l;
      if (ri != -1 && rj != -1) {
        int32_t ax = mc[0].mv[l][0] + mc[1].mv[l][0], ay = mc[0].mv[l][1] + mc[1].mv[l][1];
        ax = (ax + 1 - (ax >= 0)) >> 1; ay = (ay + 1 - (ay >= 0)) >> 1;
        mc[n].mv[l][0] = ax; mc[n].mv[l][1] = ay; mc[n].ref[l] = ri & 255;
      } else if (ri != -1) { mc[n].mv[l][0] = mc[0].mv[l][0]; mc[n].mv[l][1] = mc[0].mv[l][1]; mc[n].ref[l] = ri & 255; }
      else { mc[n].mv[l][0] = mc[1].mv[l][0]; mc[n].mv[l][1] = mc[1].mv[l][1]; mc[n].ref[l] = rj & 255; }
    }
    mc[n].dir = inter_dir;
    if (inter_dir > 0) n++;
  }
  if (n >= max_cands) return n;
  int num_ref = f.n_refs;                 // zero vectors
  if (n < max_cands && f.is_b) {
    int neg = 0, pos = 0;
    for (int j = 0; j < f.n_refs; ++j) { if (f.ref_pocs[j] < f.poc) neg++; else pos++; }
    num_ref = neg < pos ? neg : pos;
  }
  int zero_idx = 0;
  while (n < max_cands) {
    mc[n].mv[0][0] = 0; mc[n].mv[0][1] = 0;
    mc[n].ref[0] = (zero_idx >= num_ref - 1) ? 0 : zero_idx;
    mc[n].dir = 1;
    if (f.is_b) { mc[n].ref[1] = mc[n].ref[0]; mc[n].mv[1][0] = 0; mc[n].mv[1][1] = 0; mc[n].dir = 3; }
    zero_idx++;
    n++;
  }
  return n;
}

// a neighbour's vector that points to the picture being searched (add_mvp_candidate without scaling, inter.c:1185-1219)
__device__ inline bool predictor_from(const unit *c, int reflist, int target, const frame_ctx &f, int32_t mv[2])
{
  if (!c) return false;
  for (int i = 0; i < 2; ++i) {
    const int cl = i == 0 ? reflist : !reflist;
    if (!(c->dir & (1 << cl))) continue;
    if (f.l[cl][c->ref[cl] & 7] == target) { mv[0] = c->mv[cl][0]; mv[1] = c->mv[cl][1]; return true; }
  }
  return false;
}

// what amvp_candidates indexes dynamically: the caller gives it a place that is not a register (LDS)
struct amvp_ws { int32_t mv[2][2]; col_unit tc; };
// uvg_inter_get_mv_cand: the two predictors of list `reflist` for the reference index ref_idx[reflist] being searched -> out[2][2]
template <typename TAB, typename COL>
__device__ inline void amvp_candidates(const frame_ctx &f, TAB &tab, COL &col, const int32_t *hmvp, int reflist, const int32_t ref_idx[2], int32_t out[4],
                                       amvp_ws *ws)
{
  const int target = f.l[reflist][ref_idx[reflist] & 7];
  const neighbours nb = spatial(tab, f);
  col_unit &tc = ws->tc;
  const bool have_tc = f.n_refs ? temporal_unit(col, f, &tc) : false;
  int32_t (&mv)[2][2] = ws->mv;
  mv[0][0] = mv[0][1] = mv[1][0] = mv[1][1] = 0;
  int n = 0, nbn = 0;
  if (predictor_from(nb.a0, reflist, target, f, mv[n])) n++;
  else if (predictor_from(nb.a1, reflist, target, f, mv[n])) n++;
  if (predictor_from(nb.b0, reflist, target, f, mv[n])) nbn++;
  else if (predictor_from(nb.b1, reflist, target, f, mv[n])) nbn++;
  else if (predictor_from(nb.b2, reflist, target, f, mv[n])) nbn++;
  n += nbn;
  if (n > 0) { mv[0][0] = to_quarter_and_back(mv[0][0]); mv[0][1] = to_quarter_and_back(mv[0][1]); }
  if (n > 1) { mv[1][0] = to_quarter_and_back(mv[1][0]); mv[1][1] = to_quarter_and_back(mv[1][1]); }
  if (n == 2 && mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1]) n = 1;
  if (f.tmvp && f.poc > 1 && f.n_refs && n < 2 && have_tc && f.l_size[0] > 0) {
    temporal_vector(tc, reflist, f, f.ref_pocs[target], f.ref_pocs[f.l[0][0]], mv[n]);
    n++;
  }
  if (n < 2) {
    const unit *lut = reinterpret_cast<const unit *>(hmvp + 1);
    const int num = hmvp[0];
    for (int i = 0; i < (num < 4 ? num : 4) && n < 2; ++i)
      for (int ps = 0; ps < 2 && n < 2; ++ps) {
        const int cl = ps == 0 ? reflist : !reflist;
        const unit &c = lut[num - 1 - i];
        if (!(c.dir & (1 << cl))) continue;
        if (f.l[cl][c.ref[cl] & 7] == target) { mv[n][0] = c.mv[cl][0]; mv[n][1] = c.mv[cl][1]; n++; }
      }
  }
  while (n < 2) { mv[n][0] = 0; mv[n][1] = 0; n++; }
  out[0] = to_quarter_and_back(mv[0][0]); out[1] = to_quarter_and_back(mv[0][1]);
  out[2] = to_quarter_and_back(mv[1][0]); out[3] = to_quarter_and_back(mv[1][1]);
}

}  // namespace icand
