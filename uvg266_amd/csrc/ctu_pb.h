// The closed-loop search of one CTU of a P / B picture by one wave: uvg_search_lcu / search_cu (src/search.c:1299-2479) with the inter
// search (uvg_search_cu_inter, src/search_inter.c:2329-2406 -> search_pu_inter :1671-2101 -> search_pu_inter_ref :1300-1500) competing
// with the intra search of ctu_core.h, for the low-delay --preset medium configuration (BASELINE configs[2]: rd = 0, me = hexbs,
// subme = 4, bipred, early-skip, me-early-termination, max-merge 6, tmvp, WPP; one reference list entry per reference picture).
// Followed by the side effect of the deblocking filter on the stored motion (src/filter.c:745-765) and by the real coder's model
// adaptation and history table (uvg_encode_coding_tree, src/encode_coding_tree.c:1365-1727) so that the next CTU of the row starts
// from what the reference would hand it.
//
// Built on ctu_core.h compiled with -DCTU_PB (its LDS model sets then carry the 18 inter-syntax models, its level state the parked
// candidate's motion).  Same execution model: sample work on all lanes of the wave, everything that is a recurrence through the
// models or a floating-point sum whose order the reference fixes on lane 0; wave-uniform control flow everywhere else (the costs
// a decision needs are wave sums every lane holds).  The same source compiles for the host with one lane (tests/emul/).
//
// What differs from the intra walk (ctu_core.h search_ctu):
//  * ONE wave walks the tree in the reference's own order -- a CU is evaluated unsplit, THEN its split is tried (the history table and
//    the pruning test make the order visible) -- there is no depth pipeline;
//  * the 64x64 CU exists (pu-depth-inter starts at 0): it is evaluated in place, set aside in the workgroup's global scratch while
//    the split is tried, and brought back if the split loses;
//  * ONE table of the CTU's motion (pb_state::mot) serves every depth.  The reference keeps one lcu_t per depth, and a neighbour a
//    candidate derivation looks at loses the vectors of its unused lists in THAT copy only (inter_clear_cu_unused, inter.c:749-758);
//    here it loses them for good.  Only units on the right / bottom edge of a CU are ever looked at (never the CU's first unit, which
//    is what the history table and the coder read), and every look clears before it reads: decisions, reconstruction and bitstream
//    are the same; the stored vectors of UNUSED lists of such units can differ from the reference's cu_array (tests compare used lists).
#pragma once
#if !defined(CTU_PB)
#error "ctu_pb.h needs -DCTU_PB"
#endif
#include "ctu_core.h"
#if defined(__HIPCC__)
#include "satd_tile_dev.h"
#endif

#if defined(__HIPCC__) && defined(CTU_PROFILE)
#define PB_T0() const unsigned long long pb_t0__ = __builtin_amdgcn_s_memtime()
#if defined(CTU_PROFILE_WALK)      // only the walk's wave counts the phases of eval_pb (slots < 14): they add up to its time
#define PB_T1(W, slot) do { if (CTU_TID == 0 && ((slot) >= 14 || CTU_WAVE == 0)) (W)->prof_pb[slot] += __builtin_amdgcn_s_memtime() - pb_t0__; } while (0)
#else
#define PB_T1(W, slot) do { if (CTU_TID == 0) (W)->prof_pb[slot] += __builtin_amdgcn_s_memtime() - pb_t0__; } while (0)
#endif
#else
#define PB_T0() ((void)0)
#define PB_T1(W, slot) ((void)0)
#endif
// prof_pb slots: 0 candidate lists, 1 merge analysis (prediction + SATD), 2 early skip test, 3 integer motion search, 4 fractional search,
// 5 bi-prediction, 6 intra rough search + chroma trial, 7 the inter CU's prediction + residual, 8 its bits + cost, 9 the intra CU (eval_cu),
// 10 unpark / 64x64 save + restore, 11 load, 12 store + deblock side effect, 13 coder pass, 14 total, 15 4x4 leaves

namespace ctu {

enum { MI_SKIP = NMODELS, MI_PRED_MODE = NMODELS + 3, MI_MERGE_FLAG = NMODELS + 5, MI_MERGE_IDX = NMODELS + 6, MI_INTER_DIR = NMODELS + 7,
       MI_REF_PIC = NMODELS + 13, MI_MVD = NMODELS + 15, MI_MVP_IDX = NMODELS + 17, M_ROOT_CBF = 243 };

// the picture's inter state (encoder_state_t::frame, cfg)
struct pb_job {
  int32_t slice_type;                  // 0 B, 1 P
  int32_t poc, n_refs, ref_pocs[16], l_size[2], l[2][16];
  int32_t tmvp, max_merge, merge_level, frame_qp;
  int32_t bipred, fme_level, early_skip, depth_inter_min, depth_inter_max;
  const void *ref_y[16], *ref_u[16], *ref_v[16];      // the reference pictures (after their in-loop filters)
  int32_t ref_stride, ref_stride_c;
  const int32_t *ref_cu[16];           // their motion, [per 4x4][8]: type, mv[2][2], mv_dir, the POC the L0 / L1 vector points to
  int32_t ref_cu_stride;
  uvghip_inter4_t *inter4;             // OUT: the picture's second side table (cu_stride entries per row)
  uint32_t *trees;                     // OUT, optional: split_tree | mode_type_tree << 16 per 4x4
  int32_t *motion_out;                 // OUT, optional: this picture in the ref_cu layout (what later pictures read)
  int32_t *hmvp_rows;                  // [CTU row][41]: the coder's history table, carried from CTU to CTU of the row
  int32_t inflight_margin;             // 0: cfg.owf == 0; else 1 + the in-loop filters' delay in samples: frames in flight (mv_within)
};

struct pb_tab { icand::unit *p; __device__ icand::unit &at(int i) { return p[i]; } };
// the collocated picture = L0[0]'s motion on the 8x8 grid (get_temporal_merge_candidates, inter.c:935-1010)
struct pb_col {
  const int32_t *p; int stride, gw;
  const int32_t (*cache)[8]; const int32_t *cache_idx;        // two entries fetched ahead for the CU being evaluated (search_pu_inter)
  __device__ icand::col_unit at(int i) const
  {
    const int gy = i / gw, gx = i - gy * gw;
    const int32_t *o = p + ((size_t)(gy * 2) * stride + gx * 2) * 8;
    if (cache_idx[0] == i) o = cache[0]; else if (cache_idx[1] == i) o = cache[1];
    icand::col_unit c;
    c.type = o[0]; c.mv[0][0] = o[1]; c.mv[0][1] = o[2]; c.mv[1][0] = o[3]; c.mv[1][1] = o[4]; c.dir = o[5]; c.poc[0] = o[6]; c.poc[1] = o[7];
    return c;
  }
};
template <typename PX> CTU_DEV pb_col col_of(lds<PX> *S, const job<PX> &J)
{
  const pb_job &B = *J.pb;
  pb_col c = {B.ref_cu[B.l_size[0] > 0 ? B.l[0][0] : 0], B.ref_cu_stride, (J.P.pic_w + 7) / 8, pbq(S).colc, pbq(S).colc_idx};
  return c;
}
// The collocated picture's units a CU's temporal candidate can come from (temporal_unit, inter_cand_dev.h: below-right of the CU, else
// its centre) into LDS, by 16 lanes at once: the merge list and every AMVP derivation of the CU read them, lane 0 alone would wait for
// device memory each time.
template <typename PX> CTU_DEV void prefetch_col(lds<PX> *S, const job<PX> &J, int x, int y, int n)
{
  const pb_job &B = *J.pb;
  pb_state &Q = pbq(S);
  const int W = J.P.pic_w, H = J.P.pic_h, gw = (W + 7) / 8;
  const int xbr = x + n, ybr = y + n, xc = x + n / 2, yc = y + n / 2;
  const int i0 = (xbr < W && ybr < H && (ybr % 64) != 0) ? (ybr >> 3) * gw + (xbr >> 3) : -1;
  const int i1 = (xc < W && yc < H) ? (yc >> 3) * gw + (xc >> 3) : -1;
  const int32_t *tab = B.ref_cu[B.l_size[0] > 0 ? B.l[0][0] : 0];
  PAR_FOR(e, 16) {
    const int k = e >> 3, i = k ? i1 : i0;
    if (i >= 0) { const int gy = i / gw, gx = i - gy * gw; Q.colc[k][e & 7] = tab[((size_t)(gy * 2) * B.ref_cu_stride + gx * 2) * 8 + (e & 7)]; }
    if ((e & 7) == 0) Q.colc_idx[k] = i;
  }
  CTU_SYNC();
}
CTU_DEV int u_idx(int lx, int ly) { return ((ly >> 2) + 1) * 17 + (lx >> 2) + 1; }       // lx, ly >= -4

CTU_DEV int wave_sum(int v)
{
#if defined(__HIPCC__)
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
#endif
  return v;
}

// ---- contexts of the skip flag and the prediction mode (uvg_get_skip_context, search_inter.c:1228-1255) ----
template <typename PX> CTU_DEV void pb_flag_ctx(lds<PX> *S, int x, int y, int lx, int ly, int *skip_ctx, int *pred_ctx)
{
  int cs = 0, cp = 0;
  if (x) { const int u = u_idx(lx - 1, ly); cs += S->pb.fl[u][0] ? 1 : 0; cp |= S->cu[u].type == CU_INTRA; }
  if (y) { const int u = u_idx(lx, ly - 1); cs += S->pb.fl[u][0] ? 1 : 0; cp |= S->cu[u].type == CU_INTRA; }
  *skip_ctx = cs; *pred_ctx = cp;
}
// an intra CU of a P / B slice: skip flag 0 and prediction mode 1 ahead of the intra modes (mock_encode_coding_unit, :1788-1834); lane 0
template <typename PX> CTU_DEV void pb_intra_flag_bits(lds<PX> *S, const job<PX> &J, uint32_t *m_, int update, int x, int y, int lx, int ly, int n, double &bits)
{
  if (n == 4) return;
  CTU_LDS uint32_t *const m = LDSP(uint32_t, m_);
  int cs, cp;
  pb_flag_ctx(S, x, y, lx, ly, &cs, &cp);
  m_code(m, update, MI_SKIP + cs, 0, bits);
  m_code(m, update, MI_PRED_MODE + cp, 1, bits);
}

// ---- the history table (uvg_hmvp_add_mv, inter.c:1831-1905); lane 0 ----
CTU_DEV bool hmvp_dup(const int32_t *a, const int32_t *b)
{
  if (a[7] != b[7]) return false;
  for (int l = 0; l < 2; ++l)
    if ((a[7] & (1 << l)) && (a[1 + 2 * l] != b[1 + 2 * l] || a[2 + 2 * l] != b[2 + 2 * l] || a[5 + l] != b[5 + l])) return false;
  return true;
}
CTU_DEV void hmvp_add(int32_t *hm, const int32_t *e)        // e: icand::unit as 8 ints
{
  if (e[0] == CU_INTRA) return;
  int32_t *lut = hm + 1;
  const int size = hm[0];
  int duplicate = -1;
  for (int i = 0; i < size; ++i) if (hmvp_dup(e, lut + 8 * i)) { duplicate = i; break; }
  if (duplicate != 0) {
    int end = duplicate == -1 ? 5 : duplicate;
    if (end > 4) end = 4;
    if (end == 0 && size == 1) end = 1;
    for (int i = end - 1; i >= 0; --i) for (int k = 0; k < 8; ++k) lut[8 * (i + 1) + k] = lut[8 * i + k];
  }
  for (int k = 0; k < 8; ++k) lut[k] = e[k];
  if (duplicate == -1 && hm[0] < 5) hm[0]++;
}

// ---- motion compensation (inter_recon_unipred / uvg_inter_recon_bipred, inter.c:400-602; the filters of ipol-generic.c) ----
template <typename PX> CTU_DEV int ref_px(CTU_GLB const PX *ref, int stride, int pw, int ph, int x, int y)
{
  return (int)ref[(size_t)clampi(y, 0, ph - 1) * stride + clampi(x, 0, pw - 1)];
}
// A w x h block whose top-left INTEGER position in the reference plane is (x0, y0), at phase (fx, fy) (luma: 1/16 with the 8-tap
// filter, chroma: 1/32 with the 4-tap filter): horizontal pass into tmp (rows y0 - off .. of w int16 each), vertical pass into dst.
// out 0: samples; 1: the 14-bit intermediates (int16); 2: samples of the bi-prediction with the other list's intermediates in `other`.
// A lane takes SEG outputs in a row (horizontal pass) or in a column (vertical pass) at a time: SEG + TAPS - 1 loads for SEG outputs.
// Two planes of the same geometry (U and V) go through the passes together when ref2 is given: their rows are one batch of loads.
template <typename PX, int TAPS, int SEG> CTU_DEV void ipol_passes(CTU_GLB const PX *ref, CTU_GLB const PX *ref2, int stride, int pw, int ph, int x0, int y0, int w, int h,
                                                                    int fx, int fy, int out, void *dst_, void *dst2_, int dp, const int16_t *other_, const int16_t *other2_,
                                                                    int op, CTU_LDS int16_t *tmp)
{
  const int depth = (int)px_info<PX>::depth;
  const int off = TAPS == 4 ? 1 : 3;
  const int8_t *const fh = TAPS == 4 ? VVC_CHROMA_FILTER + 4 * fx : VVC_LUMA_FILTER + 8 * fx;
  const int8_t *const fv = TAPS == 4 ? VVC_CHROMA_FILTER + 4 * fy : VVC_LUMA_FILTER + 8 * fy;
  const int shift1 = depth - 8;
  const int rows = h + TAPS - 1, segs = w / SEG, planes = ref2 ? 2 : 1, plane_sz = rows * w;
  int ch[TAPS], cv[TAPS];
  for (int k = 0; k < TAPS; ++k) { ch[k] = fh[k]; cv[k] = fv[k]; }
  PAR_FOR(e0, planes * rows * segs) {
    const int pl = e0 >= rows * segs, e = e0 - pl * rows * segs;
    const int r = e / segs, q0 = (e - r * segs) * SEG;
    CTU_GLB const PX *row = (pl ? ref2 : ref) + (size_t)clampi(y0 + r - off, 0, ph - 1) * stride;
    CTU_LDS int16_t *o = tmp + pl * plane_sz + r * w + q0;
    const int xs = x0 + q0 - off;
    if (fx == 0) {
      if (xs + off >= 0 && xs + off + SEG <= pw) { for (int j = 0; j < SEG; ++j) o[j] = (int16_t)((64 * (int)row[xs + off + j]) >> shift1); }
      else for (int j = 0; j < SEG; ++j) o[j] = (int16_t)((64 * (int)row[clampi(xs + off + j, 0, pw - 1)]) >> shift1);
    } else {
      int px[SEG + TAPS - 1];
      if (xs >= 0 && xs + SEG + TAPS - 1 <= pw) { for (int j = 0; j < SEG + TAPS - 1; ++j) px[j] = (int)row[xs + j]; }
      else for (int j = 0; j < SEG + TAPS - 1; ++j) px[j] = (int)row[clampi(xs + j, 0, pw - 1)];
      for (int j = 0; j < SEG; ++j) {
        int acc = 0;
        for (int k = 0; k < TAPS; ++k) acc += ch[k] * px[j + k];
        o[j] = (int16_t)(acc >> shift1);
      }
    }
  }
  CTU_SYNC();
  const int wp_shift = 14 - depth, wp_off = 1 << (wp_shift - 1), bi_shift = 15 - depth, bi_off = 1 << (bi_shift - 1);
  const int vsegs = h / SEG;
  PAR_FOR(e0, planes * vsegs * w) {
    const int pl = e0 >= vsegs * w, e = e0 - pl * vsegs * w;
    const int sg = e / w, q = e - sg * w, r0 = sg * SEG;
    const CTU_LDS int16_t *tp = tmp + pl * plane_sz;
    int hi[SEG];
    if (fy == 0) { for (int j = 0; j < SEG; ++j) hi[j] = (int)(int16_t)((64 * (int)tp[(r0 + j + off) * w + q]) >> 6); }
    else {
      int t[SEG + TAPS - 1];
      for (int j = 0; j < SEG + TAPS - 1; ++j) t[j] = (int)tp[(r0 + j) * w + q];
      for (int j = 0; j < SEG; ++j) {
        int acc = 0;
        for (int k = 0; k < TAPS; ++k) acc += cv[k] * t[j + k];
        hi[j] = (int)(int16_t)(acc >> 6);
      }
    }
    void *const d_ = pl ? dst2_ : dst_;
    const int16_t *const oth_ = pl ? other2_ : other_;
    for (int j = 0; j < SEG; ++j) {
      const int r = r0 + j;
      if (out == 1) LDSP(int16_t, d_)[r * dp + q] = (int16_t)hi[j];
      else if (out == 0) LDSP(PX, d_)[r * dp + q] = (PX)clampi((hi[j] + wp_off) >> wp_shift, 0, (int)px_info<PX>::maxv);
      else LDSP(PX, d_)[r * dp + q] = (PX)clampi((hi[j] + (int)LDSP(const int16_t, oth_)[r * op + q] + bi_off) >> bi_shift, 0, (int)px_info<PX>::maxv);
    }
  }
  CTU_SYNC();
}
template <typename PX> CTU_NOINLINE CTU_DEV void ipol_block(const PX *ref_, int stride, int pw, int ph, int x0, int y0, int w, int h, int fx, int fy, int is_chroma,
                                                            int out, void *dst_, int dp, const int16_t *other_, int op, int16_t *tmp_,
                                                            const PX *ref2_ = nullptr, void *dst2_ = nullptr, const int16_t *other2_ = nullptr)
{
  CTU_GLB const PX *const ref = (CTU_GLB const PX *)ref_, *const ref2 = (CTU_GLB const PX *)ref2_;
  CTU_LDS int16_t *const tmp = LDSP(int16_t, tmp_);
  if (!is_chroma) ipol_passes<PX, 8, 8>(ref, ref2, stride, pw, ph, x0, y0, w, h, fx, fy, out, dst_, dst2_, dp, other_, other2_, op, tmp);
  else if (w >= 8) ipol_passes<PX, 4, 8>(ref, ref2, stride, pw, ph, x0, y0, w, h, fx, fy, out, dst_, dst2_, dp, other_, other2_, op, tmp);
  else ipol_passes<PX, 4, 4>(ref, ref2, stride, pw, ph, x0, y0, w, h, fx, fy, out, dst_, dst2_, dp, other_, other2_, op, tmp);
}

// the prediction of the n x n CU at picture position (x, y) with motion m (icand::unit fields: mv, ref = list indices, dir) into
// ry / ru / rv (luma and/or chroma); blocks wider than 32 go quadrant by quadrant (the scratch is the 32x32 depth's)
template <typename PX> CTU_NOINLINE CTU_DEV void pred_cu(lds<PX> *S, const job<PX> &J, int x, int y, int n, const icand::unit *m_, int luma, int chroma,
                                                         PX *ry, int rpy, PX *ru, PX *rv, int rpc)
{
  wctx *const V = wv_of(S);
  const pb_job &B = *J.pb;
  const icand::unit m = *m_;
  CTU_SYNC();
  const int W = J.P.pic_w, H = J.P.pic_h;
  const int q = n > 32 ? 32 : n, nq = n / q;
  const int n_lists = m.dir == 3 ? 2 : 1;
  for (int qy = 0; qy < nq; ++qy)
    for (int qx = 0; qx < nq; ++qx)
      for (int pass = 0; pass < n_lists; ++pass) {
        const int l = m.dir == 3 ? pass : m.dir - 1;
        const int ri = B.l[l][m.ref[l] & 15];
        const int mvx = m.mv[l][0], mvy = m.mv[l][1];
        const int out = m.dir == 3 ? (pass == 0 ? 1 : 2) : 0;
        if (luma) {
          const int bx = x + qx * q, by = y + qy * q;
          PX *d = ry + (qy * q) * rpy + qx * q;
          ipol_block<PX>((const PX *)B.ref_y[ri], B.ref_stride, W, H, bx + (mvx >> 4), by + (mvy >> 4), q, q, mvx & 15, mvy & 15, 0, out,
                         out == 1 ? (void *)V->lv0 : (void *)d, out == 1 ? q : rpy, V->lv0, q, V->t0);
        }
        if (chroma) {
          const int cq = q >> 1, bx = (x >> 1) + qx * cq, by = (y >> 1) + qy * cq;
          PX *du = ru + (qy * cq) * rpc + qx * cq, *dv = rv + (qy * cq) * rpc + qx * cq;
          ipol_block<PX>((const PX *)B.ref_u[ri], B.ref_stride_c, W >> 1, H >> 1, bx + (mvx >> 5), by + (mvy >> 5), cq, cq, mvx & 31, mvy & 31, 1, out,
                         out == 1 ? (void *)V->lv1 : (void *)du, out == 1 ? cq : rpc, V->lv1, cq, V->t0,
                         (const PX *)B.ref_v[ri], out == 1 ? (void *)V->lv2 : (void *)dv, V->lv2);       // U and V together
        }
      }
}

// ---- distortion: uvg_image_calc_sad at an integer displacement (image.c:322-472: the reference's case analysis of the overhang is
// a per-sample clamp of the reference position), uvg_satd_any_size of the block against a prediction in LDS ----
template <typename PX> CTU_DEV unsigned sad_at(const job<PX> &J, int ref_pic, int x, int y, int n, int dx, int dy)
{
  const pb_job &B = *J.pb;
  CTU_GLB const PX *const ref = (CTU_GLB const PX *)B.ref_y[ref_pic];
  CTU_GLB const PX *const cur = (CTU_GLB const PX *)J.src_y + (size_t)y * J.src_stride + x;
  const int W = J.P.pic_w, H = J.P.pic_h, l2 = ilog2_dev(n);
  int acc = 0;
  PAR_FOR(e, n * n) {
    const int r = e >> l2, q = e & (n - 1);
    acc += iabs_((int)cur[(size_t)r * J.src_stride + q] - ref_px(ref, B.ref_stride, W, H, x + dx + q, y + dy + r));
  }
  return (unsigned)wave_sum(acc) >> ((int)px_info<PX>::depth - 8);
}
template <typename PX> CTU_DEV unsigned satd_vs_source(const job<PX> &J, int x, int y, int n, const PX *pred_, int pp)
{
  CTU_LDS const PX *const pred = LDSP(const PX, pred_);
  CTU_GLB const PX *const cur = (CTU_GLB const PX *)J.src_y + (size_t)y * J.src_stride + x;
  const int tx_n = n >> 3, tiles = tx_n * tx_n;
  int acc = 0;
  PAR_FOR(t, tiles) {
    const int ty = t / tx_n, tx = t - ty * tx_n;
#if defined(__HIPCC__)
    uint32_t d[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      // the source row of the tile as sample pairs: whole words (the block's x is a multiple of 8)
      uint32_t cp[4];
      const CTU_GLB uint32_t *c32 = (const CTU_GLB uint32_t *)(cur + (size_t)(ty * 8 + r) * J.src_stride + tx * 8);
      if (sizeof(PX) == 1) {
        const uint32_t a = c32[0], b = c32[1];
        cp[0] = (a & 0xffu) | ((a & 0xff00u) << 8); cp[1] = ((a >> 16) & 0xffu) | ((a >> 24) << 16);
        cp[2] = (b & 0xffu) | ((b & 0xff00u) << 8); cp[3] = ((b >> 16) & 0xffu) | ((b >> 24) << 16);
      } else { cp[0] = c32[0]; cp[1] = c32[1]; cp[2] = c32[2]; cp[3] = c32[3]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const CTU_LDS PX *p = pred + (ty * 8 + r) * pp + tx * 8 + 2 * q;
        d[r][q] = pk_sub(cp[q], (uint32_t)p[0] | ((uint32_t)p[1] << 16));
      }
    }
    acc += (int)satd8_tile_lane(d);
#else
    int d[64];
    for (int r = 0; r < 8; ++r)
      for (int q = 0; q < 8; ++q) d[r * 8 + q] = (int)cur[(size_t)(ty * 8 + r) * J.src_stride + tx * 8 + q] - (int)pred[(ty * 8 + r) * pp + tx * 8 + q];
    acc += (int)satd8_tile(d);
#endif
  }
  return (unsigned)wave_sum(acc) >> ((int)px_info<PX>::depth - 8);
}

// ---- motion vector costs (search_inter.c:378-488; every call site of this configuration has num_cand = 0) ----
CTU_DEV unsigned golomb_bits(unsigned s)
{
  unsigned bins = 0;
  if (s >= 1u << 8) { bins += 16; s >>= 8; }
  if (s >= 1u << 4) { bins += 8; s >>= 4; }
  if (s >= 1u << 2) { bins += 4; s >>= 2; }
  if (s >= 1u << 1) bins += 2;
  return bins;
}
CTU_DEV int to_quarter(int v) { return v >= 0 ? (v + 1) >> 2 : (v + 2) >> 2; }
CTU_DEV double mvd_coding_cost(int dx, int dy)            // get_mvd_coding_cost: fixed point with 15 fractional bits, exact in double
{
  const int ax = iabs_(dx), ay = iabs_(dy);
  return (double)(4 + (ax == 1) + (ay == 1) + (int)golomb_bits((unsigned)ax) + (int)golomb_bits((unsigned)ay));
}
// select_mv_cand: the cheaper predictor; -> its index, *cost_out its cost
CTU_DEV int select_mv_cand(const int32_t (&c)[2][2], int mx, int my, double *cost_out)
{
  const bool same = c[0][0] == c[1][0] && c[0][1] == c[1][1];
  if (same && !cost_out) return 0;
  const double c1 = mvd_coding_cost(to_quarter(mx - c[0][0]), to_quarter(my - c[0][1]));
  const double c2 = same ? c1 : mvd_coding_cost(to_quarter(mx - c[1][0]), to_quarter(my - c[1][1]));
  if (cost_out) *cost_out = c1 < c2 ? c1 : c2;
  return c2 < c1 ? 1 : 0;
}
CTU_DEV double calc_mvd_cost(double lambda_sqrt, int x, int y, int mv_shift, const int32_t (&c)[2][2], double *bitcost)
{
  x *= 1 << mv_shift;
  y *= 1 << mv_shift;
  double mvd_cost = 0;
  select_mv_cand(c, x, y, &mvd_cost);
  *bitcost = mvd_cost;
  return mvd_cost * lambda_sqrt;
}

// fracmv_within_tile (search_inter.c:94-149) with cfg.wpp on and no mv constraint: with frames in flight (cfg.owf) the block a vector (1/16
// samples) refers to -- a margin for the interpolation taps and the in-loop filters' delay included -- may lie at most max_inter_ref_lcu.down = 1
// CTU rows below the block's own CTU and, rows and columns together, 2 CTUs down-right (encoder.c:244-245): the reference picture is still
// being coded below that.
CTU_DEV bool mv_within(const pb_job &B, int x, int y, int n, int mvx, int mvy)
{
  if (!B.inflight_margin) return true;
  const bool frac_luma = (mvx & 15) != 0 || (mvy & 15) != 0, frac_chroma = (mvx & 31) != 0 || (mvy & 31) != 0;
  const int margin = 2 + (frac_luma ? 4 : (frac_chroma ? 2 : 0)) + B.inflight_margin - 1;
  const int lx = ((x + n + margin) * 16 + mvx) / (64 << 4) - x / 64;
  const int ly = ((y + n + margin) * 16 + mvy) / (64 << 4) - y / 64;
  return ly <= 1 && lx + ly <= 2;
}

// ---- the integer and fractional motion search of one reference picture (wave-uniform) ----
struct me_best { double cost, bits; int mx, my; };        // vector in 1/16 units
template <typename PX> struct me_info {
  const job<PX> *J;
  int ref_pic, x, y, n;
  int32_t cand[2][2];
};
template <typename PX> CTU_DEV int check_mv_cost(const me_info<PX> &I, int dx, int dy, me_best &b)
{
  if (!mv_within(*I.J->pb, I.x, I.y, I.n, dx * 16, dy * 16)) return 0;
  double bitcost = 0;
  double cost = (double)sad_at(*I.J, I.ref_pic, I.x, I.y, I.n, dx, dy);
  if (cost >= b.cost) return 0;
  cost += calc_mvd_cost(I.J->P.lambda_sqrt, dx, dy, 4, I.cand, &bitcost);
  if (cost >= b.cost) return 0;
  b.mx = dx * 16; b.my = dy * 16; b.cost = cost; b.bits = bitcost;
  return 1;
}
template <typename PX> CTU_DEV bool mv_in_merge(lds<PX> *S, int mx, int my)
{
  for (int i = 0; i < pbq(S).n_mc; ++i) {
    const icand::merge_cand &c = pbq(S).mc[i];
    if (c.dir == 3) continue;
    const int l = c.dir - 1;
    if (c.mv[l][0] == mx * 16 && c.mv[l][1] == my * 16) return true;
  }
  return false;
}
template <typename PX> CTU_NOINLINE CTU_DEV void me_integer(lds<PX> *S, const me_info<PX> &I, int ex, int ey, me_best &b)
{
  // select_starting_point (search_inter.c:297-375)
  check_mv_cost(I, 0, 0, b);
  ex >>= 4; ey >>= 4;
  if ((ex != 0 || ey != 0) && !mv_in_merge(S, ex, ey)) check_mv_cost(I, ex, ey, b);
  for (int i = 0; i < pbq(S).n_mc; ++i) {
    const icand::merge_cand &c = pbq(S).mc[i];
    if (c.dir == 3) continue;
    const int l = c.dir - 1;
    const int px = (c.mv[l][0] + 8) >> 4, py = (c.mv[l][1] + 8) >> 4;
    if (px == 0 && py == 0) continue;
    check_mv_cost(I, px, py, b);
  }
  // early_terminate (:491-539), me-early-termination = on
  bool skip_me = false;
  {
    const int sx[7] = {0, -1, 0, 1, 0, -1, 0}, sy[7] = {-1, 0, 1, 0, -1, 0, 0};
    int mx = b.mx >> 4, my = b.my >> 4;
    int first_index = 0, last_index = 3;
    for (int k = 0; k < 2 && !skip_me; ++k) {
      const double threshold = b.cost;
      int best_index = 6;
      for (int i = first_index; i <= last_index; i++)
        if (check_mv_cost(I, mx + sx[i], my + sy[i], b)) best_index = i;
      mx += sx[best_index]; my += sy[best_index];
      if (b.cost >= threshold) { skip_me = true; break; }
      first_index = (best_index + 3) % 4;
      last_index = first_index + 2;
    }
  }
  if (skip_me) return;
  // hexagon_search (:767-847), unlimited steps
  {
    const int lx_[9] = {0, 1, 2, 1, -1, -2, -1, 1, 2}, ly_[9] = {0, -2, 0, 2, 2, 0, -2, -2, 0};
    const int qx[9] = {0, 0, -1, 1, 0, -1, 1, -1, 1}, qy[9] = {0, -1, 0, 0, 1, -1, -1, 1, 1};
    int mx = b.mx >> 4, my = b.my >> 4;
    int best_index = 0;
    for (int i = 1; i < 7; ++i)
      if (check_mv_cost(I, mx + lx_[i], my + ly_[i], b)) best_index = i;
    while (best_index != 0) {
      const int start = best_index == 1 ? 6 : (best_index == 8 ? 1 : best_index - 1);
      mx += lx_[best_index]; my += ly_[best_index];
      best_index = 0;
      for (int i = 0; i < 3; ++i)
        if (check_mv_cost(I, mx + lx_[start + i], my + ly_[start + i], b)) best_index = start + i;
    }
    for (int i = 1; i < 9; ++i) check_mv_cost(I, mx + qx[i], my + qy[i], b);
  }
}
// search_frac (:1029-1226): every candidate block of a step is the prediction at that fractional position (tools/refcheck proves the identity against the four-block functions)
template <typename PX> CTU_NOINLINE CTU_DEV void me_frac(lds<PX> *S, const me_info<PX> &I, PX *pred, int pp, me_best &b)
{
  wctx *const V = wv_of(S);
  const job<PX> &J = *I.J;
  const pb_job &B = *J.pb;
  const int sqx[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1}, sqy[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
  const int W = J.P.pic_w, H = J.P.pic_h, n = I.n, fme_level = B.fme_level;
  const int q = n > 32 ? 32 : n, nq = n / q;
  int mx = b.mx >> 4, my = b.my >> 4;
  double bitcosts[4] = {0, 0, 0, 0};
  unsigned costs[4] = {0, 0, 0, 0};
  unsigned best_index = 0;
  for (int qy = 0; qy < nq; ++qy)
    for (int qx = 0; qx < nq; ++qx)
      ipol_block<PX>((const PX *)B.ref_y[I.ref_pic], B.ref_stride, W, H, I.x + qx * q + mx, I.y + qy * q + my, q, q, 0, 0, 0, 0, pred + qy * q * pp + qx * q, pp,
                     nullptr, 0, V->t0);
  costs[0] = satd_vs_source(J, I.x, I.y, n, pred, pp);
  costs[0] += (uint32_t)calc_mvd_cost(J.P.lambda_sqrt, mx, my, 4, I.cand, &bitcosts[0]);
  double cost = costs[0], bitcost = bitcosts[0];
  mx *= 2; my *= 2;
  int i = 1;
  for (int step = 0; step < fme_level; ++step) {
    const int mv_shift = step < 2 ? 3 : 2;
    bool within[4];
    for (int j = 0; j < 4; ++j) {
      const int fxp = (mx + sqx[i + j]) * (1 << mv_shift), fyp = (my + sqy[i + j]) * (1 << mv_shift);
      within[j] = mv_within(B, I.x, I.y, n, fxp, fyp);
      if (!within[j]) continue;          // (the reference measures the candidate and then ignores it, search_inter.c:1160-1194)
      for (int qy = 0; qy < nq; ++qy)
        for (int qx = 0; qx < nq; ++qx)
          ipol_block<PX>((const PX *)B.ref_y[I.ref_pic], B.ref_stride, W, H, I.x + qx * q + (fxp >> 4), I.y + qy * q + (fyp >> 4), q, q, fxp & 15, fyp & 15, 0, 0,
                         pred + qy * q * pp + qx * q, pp, nullptr, 0, V->t0);
      costs[j] = satd_vs_source(J, I.x, I.y, n, pred, pp);
      costs[j] += (uint32_t)calc_mvd_cost(J.P.lambda_sqrt, mx + sqx[i + j], my + sqy[i + j], mv_shift, I.cand, &bitcosts[j]);
    }
    for (int j = 0; j < 4; ++j)
      if (within[j] && costs[j] < cost) { cost = costs[j]; bitcost = bitcosts[j]; best_index = (unsigned)(i + j); }
    i += 4;
    if (step == 1 || step == fme_level - 1) {
      mx += sqx[best_index]; my += sqy[best_index];
      if (step == (fme_level - 1 < 1 ? fme_level - 1 : 1)) { mx *= 2; my *= 2; best_index = 0; i = 1; }
    }
  }
  b.mx = mx * 4; b.my = my * 4; b.cost = cost; b.bits = bitcost;
}

CTU_DEV int scaled_mv(int mv, int scale) { const int s = scale * mv; return clampi((s + 127 + (s < 0)) >> 8, -131072, 131071); }
CTU_DEV void apply_mv_scaling(int cur_poc, int cur_ref_poc, int nb_poc, int nb_ref_poc, int *mx, int *my)      // search_inter.c:1260-1294
{
  int dc = cur_poc - cur_ref_poc, dn = nb_poc - nb_ref_poc;
  if (dc == dn) return;
  if (dn == 0) return;
  dc = clampi(dc, -128, 127);
  dn = clampi(dn, -128, 127);
  const int scale = clampi((dc * ((0x4000 + (iabs_(dn) >> 1)) / dn) + 32) >> 6, -4096, 4095);
  *mx = scaled_mv(*mx, scale);
  *my = scaled_mv(*my, scale);
}

// sort_keys_by_cost (search_inter.c:1619-1634): insertion sort of the keys, stable; lane 0
CTU_DEV void sort_keys(const pb_cand *units, int8_t *keys, int size)
{
  for (int i = 1; i < size; ++i) {
    const int8_t cur = keys[i];
    const double cur_cost = units[cur].cost;
    int j = i;
    while (j > 0 && cur_cost < units[keys[j - 1]].cost) { keys[j] = keys[j - 1]; --j; }
    keys[j] = cur;
  }
}

// ---- where a CU's samples and levels go while it is evaluated: the decided planes (64x64) or the depth's candidate buffers ----
template <typename PX> struct cu_target {
  PX *ry, *ru, *rv;
  int16_t *ky, *ku, *kv;
  int rpy, rpc, kpy, kpc;
};
template <typename PX> CTU_DEV cu_target<PX> target_of(lds<PX> *S, const job<PX> &J, int L)
{
  cu_target<PX> T;
  if (L == 0 && S->depth_wave == 2) {          // four waves: the 64x64 CU is a candidate like the others (the walk is already in its children)
    T.ry = S->cand64_px; T.ru = S->cand64_px + 4096; T.rv = S->cand64_px + 5120;
    T.ky = S->cand64_co; T.ku = S->cand64_co + 4096; T.kv = S->cand64_co + 5120;
    T.rpy = T.kpy = LCU; T.rpc = T.kpc = LCU_C;
  } else if (L == 0) {
    T.ry = S->Dy + PY + 1; T.ru = S->Du + PC + 1; T.rv = S->Dv + PC + 1;
    T.ky = J.coeff; T.ku = J.coeff + 4096; T.kv = J.coeff + 5120;
    T.rpy = PY; T.rpc = PC; T.kpy = LCU; T.kpc = LCU_C;
  } else {
    const int n = 64 >> L;
    T.ry = S->cand_px + cand_px_off(L, 0); T.ru = S->cand_px + cand_px_off(L, 1); T.rv = S->cand_px + cand_px_off(L, 2);
    int16_t *const kb = J.W->cand_co;
    T.ky = kb + cand_px_off(L, 0); T.ku = kb + cand_px_off(L, 1); T.kv = kb + cand_px_off(L, 2);
    T.rpy = T.kpy = n; T.rpc = T.kpc = n >> 1;
  }
  return T;
}

// the candidate lists of the CU at (x, y), n x n, through csrc/inter_cand_dev.h; lane 0
template <typename PX> CTU_DEV void set_cand_ctx(lds<PX> *S, int x, int y, int n, uint32_t split_tree)
{
  { icand::frame_ctx &f = pbq(S).f; f.x = x; f.y = y; f.w = n; f.h = n; f.split_tree = split_tree; }
}
// (During a CU's evaluation nothing the derivation reads changes -- the history table is the search's at the CU's entry, the neighbours
// only lose unused vectors -- so the predictors of a (list, reference index) are derived once per CU; the coder's calls, on its own
// history table, are not cached.)
template <typename PX> CTU_DEV void amvp_for(lds<PX> *S, const job<PX> &J, const int32_t *hmvp, int reflist, int ref0, int ref1, int32_t (*out)[2])
{
  pb_state &Q = pbq(S);
  const int ri = (reflist ? ref1 : ref0) & 7;
  const bool cacheable = hmvp == Q.hm;
  if (cacheable && (Q.amvp_key[0] != Q.f.x || Q.amvp_key[1] != Q.f.y || Q.amvp_key[2] != Q.f.w)) {
    Q.amvp_key[0] = Q.f.x; Q.amvp_key[1] = Q.f.y; Q.amvp_key[2] = Q.f.w; Q.amvp_have[0] = Q.amvp_have[1] = 0;
  }
  if (cacheable && ((Q.amvp_have[reflist] >> ri) & 1u)) {
    const int32_t *c = Q.amvp_cache[reflist][ri];
    out[0][0] = c[0]; out[0][1] = c[1]; out[1][0] = c[2]; out[1][1] = c[3];
    return;
  }
  pb_tab tab = {S->pb.mot};
  pb_col col = col_of(S, J);
  Q.ref_idx2[0] = ref0; Q.ref_idx2[1] = ref1;
  icand::amvp_candidates(Q.f, tab, col, hmvp, reflist, Q.ref_idx2, Q.out4, &Q.ws);
  out[0][0] = Q.out4[0]; out[0][1] = Q.out4[1]; out[1][0] = Q.out4[2]; out[1][1] = Q.out4[3];
  if (cacheable) { int32_t *c = Q.amvp_cache[reflist][ri]; c[0] = Q.out4[0]; c[1] = Q.out4[1]; c[2] = Q.out4[2]; c[3] = Q.out4[3]; Q.amvp_have[reflist] |= 1u << ri; }
}

CTU_DEV void cand_from_merge(pb_cand *pu, const icand::merge_cand &c)          // the motion fields search_pu_inter copies out of a merge candidate
{
  pu->m.dir = c.dir;
  pu->m.ref[0] = c.ref[0] & 255; pu->m.ref[1] = c.ref[1] & 255;
  pu->m.mv[0][0] = c.mv[0][0]; pu->m.mv[0][1] = c.mv[0][1]; pu->m.mv[1][0] = c.mv[1][0]; pu->m.mv[1][1] = c.mv[1][1];
}
CTU_DEV bool same_merge(const icand::merge_cand &a, const icand::merge_cand &b)
{
  return a.dir == b.dir && a.ref[0] == b.ref[0] && a.mv[0][0] == b.mv[0][0] && a.mv[0][1] == b.mv[0][1] && a.ref[1] == b.ref[1] && a.mv[1][0] == b.mv[1][0] &&
         a.mv[1][1] == b.mv[1][1];
}

// the residual of the CU's transform units against the prediction in T: uvg_quantize_lcu_residual (transform.c:1487-1603) for an inter
// CU; -> the flags of up to four units (cbf4) and their union.  early: only the flags are wanted (no reconstruction).  luma_done: the
// luma levels of every unit are in T.ky already (the early-skip test of this very candidate quantised them): reconstruction only.
template <typename PX> CTU_NOINLINE CTU_DEV int quantize_inter(lds<PX> *S, const job<PX> &J, int x, int y, int n, const cu_target<PX> &T, int luma, int chroma, int early,
                                                              int32_t *cbf4, int luma_done = 0)
{
  const int q = n > 32 ? 32 : n, nq = n / q;
  int any = 0;
  for (int i = 0; i < nq * nq; ++i) {
    const int ox = (i & 1) * q, oy = (i >> 1) * q, tx = x + ox, ty = y + oy;
    const int fl = 1 | (early ? 2 : 0) | 4;
    int cbf = cbf4[i];
    if (luma) {
      cbf &= ~1;
      cbf |= recon_tu(S, J, 0, tx, ty, tx & 63, ty & 63, q, 0, 0, T.ry + oy * T.rpy + ox, T.rpy, T.ky + oy * T.kpy + ox, T.kpy, q, fl | (luma_done ? 8 : 0));
    }
    if (chroma) {
      cbf &= ~6;
      const int cu = recon_tu(S, J, 1, tx, ty, tx & 63, ty & 63, q, 0, 0, T.ru + (oy >> 1) * T.rpc + (ox >> 1), T.rpc, T.ku + (oy >> 1) * T.kpc + (ox >> 1), T.kpc, q, fl);
      const int cv = recon_tu(S, J, 2, tx, ty, tx & 63, ty & 63, q, 0, cu, T.rv + (oy >> 1) * T.rpc + (ox >> 1), T.rpc, T.kv + (oy >> 1) * T.kpc + (ox >> 1), T.kpc, q, fl);
      cbf |= cu << 1 | cv << 2;
    }
    cbf4[i] = cbf;
    any |= cbf;
  }
  return any;
}

// The merge analysis of an 8x8 CU, all candidates at once.  One candidate's prediction occupies 15 of the wave's 64 lanes and waits for
// memory at every step; the (up to six) candidates are independent, so their rows are fetched and filtered together, then their
// columns, then each candidate's 8x8 SATD on a lane of its own.  Scratch: the 32x32 depth's transform buffers, idle while one wave walks
// the CTU -- tmp [candidate][list][15 rows][8] int16, the predictions [candidate][64] behind the first list's intermediates.
// Same arithmetic as ipol_passes<PX, 8, 8> / uvg_bipred_average.  -> S->wv[3].rq_i[c]: the SATD of candidate c; its samples stay in pred8(c)
template <typename PX> CTU_DEV PX *pred8_of(lds<PX> *S, int c) { return (S->depth_wave ? S->b8_pred : (PX *)(S->wv[3].lv0 + 512)) + c * 64; }
template <typename PX> CTU_DEV int32_t *satd8_of(lds<PX> *S) { return S->depth_wave ? S->b8_satd : S->wv[3].rq_i; }
template <typename PX> CTU_NOINLINE CTU_DEV void merge_batch8(lds<PX> *S, const job<PX> &J, int x, int y, int n_mc)
{
  const pb_job &B = *J.pb;
  pb_state &Q = pbq(S);
  CTU_LDS int16_t *const tmp = LDSP(int16_t, S->depth_wave ? S->b8_tmp : S->wv[3].t0);
  int32_t *const satd_out = satd8_of(S);
  const int depth = (int)px_info<PX>::depth, shift1 = depth - 8;
  const int W = J.P.pic_w, H = J.P.pic_h;
  PAR_FOR(e, n_mc * 2 * 15) {
    const int c = e / 30, l = (e / 15) & 1, r = e % 15;
    const icand::merge_cand &m = Q.mc[c];
    if (!(m.dir & (1 << l))) continue;
    CTU_GLB const PX *const ref = (CTU_GLB const PX *)B.ref_y[B.l[l][m.ref[l] & 15]];
    const int mvx = m.mv[l][0], mvy = m.mv[l][1], fx = mvx & 15;
    const int x0 = x + (mvx >> 4), y0 = y + (mvy >> 4);
    CTU_GLB const PX *row = ref + (size_t)clampi(y0 + r - 3, 0, H - 1) * B.ref_stride;
    CTU_LDS int16_t *o = tmp + ((c * 2 + l) * 15 + r) * 8;
    const int xs = x0 - 3;
    if (fx == 0) {
      for (int j = 0; j < 8; ++j) o[j] = (int16_t)((64 * (int)row[clampi(xs + 3 + j, 0, W - 1)]) >> shift1);
    } else {
      const int8_t *fh = VVC_LUMA_FILTER + 8 * fx;
      int px[15];
      if (xs >= 0 && xs + 15 <= W) { for (int j = 0; j < 15; ++j) px[j] = (int)row[xs + j]; }
      else for (int j = 0; j < 15; ++j) px[j] = (int)row[clampi(xs + j, 0, W - 1)];
      for (int j = 0; j < 8; ++j) {
        int acc = 0;
        for (int k = 0; k < 8; ++k) acc += fh[k] * px[j + k];
        o[j] = (int16_t)(acc >> shift1);
      }
    }
  }
  CTU_SYNC();
  const int wp_shift = 14 - depth, wp_off = 1 << (wp_shift - 1), bi_shift = 15 - depth, bi_off = 1 << (bi_shift - 1);
  PAR_FOR(e, n_mc * 8) {
    const int c = e >> 3, q = e & 7;
    const icand::merge_cand &m = Q.mc[c];
    int hi[2][8];
    for (int l = 0; l < 2; ++l) {
      if (!(m.dir & (1 << l))) continue;
      const int fy = m.mv[l][1] & 15;
      const CTU_LDS int16_t *t = tmp + (c * 2 + l) * 120 + q;
      if (fy == 0) { for (int j = 0; j < 8; ++j) hi[l][j] = (int)t[(j + 3) * 8]; }
      else {
        const int8_t *fv = VVC_LUMA_FILTER + 8 * fy;
        int v[15];
        for (int j = 0; j < 15; ++j) v[j] = (int)t[j * 8];
        for (int j = 0; j < 8; ++j) {
          int acc = 0;
          for (int k = 0; k < 8; ++k) acc += fv[k] * v[j + k];
          hi[l][j] = (int)(int16_t)(acc >> 6);
        }
      }
    }
    CTU_LDS PX *p = LDSP(PX, pred8_of(S, c)) + q;
    for (int j = 0; j < 8; ++j) {
      int v;
      if (m.dir == 3) v = (hi[0][j] + hi[1][j] + bi_off) >> bi_shift;
      else v = (hi[m.dir - 1][j] + wp_off) >> wp_shift;
      p[j * 8] = (PX)clampi(v, 0, (int)px_info<PX>::maxv);
    }
  }
  CTU_SYNC();
  PAR_FOR(c, n_mc) {
    CTU_GLB const PX *const cur = (CTU_GLB const PX *)J.src_y + (size_t)y * J.src_stride + x;
    CTU_LDS const PX *const pred = LDSP(const PX, pred8_of(S, c));
#if defined(__HIPCC__)
    uint32_t d[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t cp[4];
      const CTU_GLB uint32_t *c32 = (const CTU_GLB uint32_t *)(cur + (size_t)r * J.src_stride);
      if (sizeof(PX) == 1) {
        const uint32_t a = c32[0], b = c32[1];
        cp[0] = (a & 0xffu) | ((a & 0xff00u) << 8); cp[1] = ((a >> 16) & 0xffu) | ((a >> 24) << 16);
        cp[2] = (b & 0xffu) | ((b & 0xff00u) << 8); cp[3] = ((b >> 16) & 0xffu) | ((b >> 24) << 16);
      } else { cp[0] = c32[0]; cp[1] = c32[1]; cp[2] = c32[2]; cp[3] = c32[3]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) d[r][q] = pk_sub(cp[q], (uint32_t)pred[r * 8 + 2 * q] | ((uint32_t)pred[r * 8 + 2 * q + 1] << 16));
    }
    satd_out[c] = (int32_t)(satd8_tile_lane(d) >> (depth - 8));
#else
    int d[64];
    for (int r = 0; r < 8; ++r)
      for (int q = 0; q < 8; ++q) d[r * 8 + q] = (int)cur[(size_t)r * J.src_stride + q] - (int)pred[r * 8 + q];
    satd_out[c] = (int32_t)(satd8_tile(d) >> (depth - 8));
#endif
  }
  CTU_SYNC();
}

// search_pu_inter (search_inter.c:1671-2101): merge analysis, the early skip test, the motion search per reference picture, the
// fractional search of each list's best, the bi-prediction of the two; the maps stay in S->pb.  -> 1: early skip (S->pb.cur is the CU)
template <typename PX> CTU_NOINLINE CTU_DEV int search_pu_inter(lds<PX> *S, const job<PX> &J, int L, const cu_target<PX> &T)
{
  wctx *const V = wv_of(S);
  const params &P = J.P;
  const pb_job &B = *J.pb;
  pb_state &Q = pbq(S);
  level_state &N = S->lvl[L];
  const int n = 64 >> L, x = N.x, y = N.y, lx = x & 63, ly = y & 63;
  CTU_LDS const uint32_t *const mdl = LDSP(const uint32_t, V->cur);
  { PB_T0();
  if (B.tmvp && B.n_refs) prefetch_col(S, J, x, y, n);
  SERIAL {
    Q.early_tag = 0;
    memset(&Q.cur, 0, sizeof Q.cur);                     // cur_pu: the CU's entry after search_cu's memset (type NOTSET)
    set_cand_ctx(S, x, y, n, N.split_tree);
    pb_tab tab = {S->pb.mot};
    pb_col col = col_of(S, J);
    Q.n_mc = icand::merge_candidates(Q.f, tab, col, Q.hm, Q.mc);
    Q.merge_size = 0;
    for (int i = 0; i < 6; ++i) { Q.merge_keys[i] = -1; Q.merge[i].cost = CTU_MAX_DOUBLE; }
  }
  CTU_SYNC();
  PB_T1(J.W, 0); }
  const double merge_flag_cost = m_fbits(mdl, MI_MERGE_FLAG, 1);
  const int n_mc = Q.n_mc;
  const bool batch8 = n == 8;
  { PB_T0();
  if (batch8) merge_batch8(S, J, x, y, n_mc);
  for (int merge_idx = 0; merge_idx < n_mc; ++merge_idx) {
    SERIAL cand_from_merge(&Q.cur, Q.mc[merge_idx]);
    CTU_SYNC();
    const int dir = Q.cur.m.dir;
    if (dir == 3 && !B.bipred) continue;
    if (dir == 3 && !(n + n > 12)) continue;
    bool dup = false;
    for (int i = 0; i < Q.merge_size && !dup; ++i) dup = same_merge(Q.mc[merge_idx], Q.mc[Q.merge[Q.merge_keys[i]].merge_idx]);
    // (candidates whose vectors reach beyond what is final in the reference picture are not tried, search_inter.c:1749-1757)
    if (((dir & 1) && !mv_within(B, x, y, n, Q.cur.m.mv[0][0], Q.cur.m.mv[0][1])) || ((dir & 2) && !mv_within(B, x, y, n, Q.cur.m.mv[1][0], Q.cur.m.mv[1][1])) || dup) continue;
    unsigned satd;
    if (batch8) satd = (unsigned)satd8_of(S)[merge_idx];
    else {
      pred_cu(S, J, x, y, n, &Q.cur.m, 1, 0, T.ry, T.rpy, T.ru, T.rv, T.rpc);
      satd = satd_vs_source(J, x, y, n, T.ry, T.rpy);
    }
    SERIAL {
      const int e = Q.merge_size;
      pb_cand &u = Q.merge[e];
      u = Q.cur;
      u.m.type = CU_INTER; u.merge_idx = (uint8_t)merge_idx; u.merged = 1; u.skipped = 0;
      double bits = merge_flag_cost + merge_idx + m_fbits(mdl, MI_MERGE_IDX, merge_idx != 0);
      double cost = (double)satd;
      cost += bits * P.lambda_sqrt;
      u.cost = cost; u.bits = bits;
      Q.merge_keys[e] = (int8_t)e;
      Q.merge_size = e + 1;
    }
    CTU_SYNC();
  }
  SERIAL sort_keys(Q.merge, Q.merge_keys, Q.merge_size);
  CTU_SYNC();
  PB_T1(J.W, 1); }
  const int num_rdo_cands = Q.merge_size < 1 ? Q.merge_size : 1;
  if (B.early_skip) {
    PB_T0();
    for (int k = 0; k < num_rdo_cands; ++k) {
      const int merge_idx = Q.merge[Q.merge_keys[k]].merge_idx;
      SERIAL cand_from_merge(&Q.cur, Q.mc[merge_idx]);
      CTU_SYNC();
      if (batch8) {
        PAR_FOR(e, 64) LDSP(PX, T.ry)[(e >> 3) * T.rpy + (e & 7)] = LDSP(const PX, pred8_of(S, merge_idx))[e];
        CTU_SYNC();
      } else pred_cu(S, J, x, y, n, &Q.cur.m, 1, 0, T.ry, T.rpy, T.ru, T.rv, T.rpc);
      int32_t cbf4[4] = {0, 0, 0, 0};
      const int luma_any = quantize_inter(S, J, x, y, n, T, 1, 0, 1, cbf4);
      // (the candidate's luma levels stay in T.ky: if it becomes the CU, finish_inter need not quantise the same residual again)
      SERIAL Q.early_tag = merge_idx + 1;
      CTU_SYNC();
      if (luma_any) continue;
      pred_cu(S, J, x, y, n, &Q.cur.m, 0, 1, T.ry, T.rpy, T.ru, T.rv, T.rpc);
      if (quantize_inter(S, J, x, y, n, T, 0, 1, 1, cbf4)) continue;
      SERIAL {
        Q.cur.m.type = CU_INTER; Q.cur.merge_idx = (uint8_t)merge_idx; Q.cur.skipped = 1;
        Q.cur.cost = 0.0; Q.cur.bits = merge_idx;
        Q.merge_size = 1;
        Q.merge[0] = Q.cur;
      }
      CTU_SYNC();
      PB_T1(J.W, 2);
      return 1;
    }
    PB_T1(J.W, 2);
  }
  // ---- AMVP: every reference picture (search_pu_inter_ref) ----
  SERIAL { Q.amvp_size[0] = Q.amvp_size[1] = Q.amvp_size[2] = 0; }
  CTU_SYNC();
  for (int ref_pic = 0; ref_pic < B.n_refs; ++ref_pic) {
    int active[2] = {0, 0}, idx[2] = {-1, -1};
    for (int rl = 0; rl < 2; ++rl)
      for (int i = 0; i < B.l_size[rl]; ++i)
        if (B.l[rl][i] == ref_pic) { active[rl] = 1; idx[rl] = i; break; }
    int ref_list = active[0] ? 0 : 1;
    int LX_idx = idx[ref_list];
    { PB_T0();
    SERIAL {
      Q.cur.m.ref[ref_list] = LX_idx & 255;
      amvp_for(S, J, Q.hm, ref_list, Q.cur.m.ref[0], Q.cur.m.ref[1], Q.mv_cand);
    }
    CTU_SYNC();
    PB_T1(J.W, 0); }
    me_info<PX> I;
    I.J = &J; I.ref_pic = ref_pic; I.x = x; I.y = y; I.n = n;
    I.cand[0][0] = Q.mv_cand[0][0]; I.cand[0][1] = Q.mv_cand[0][1]; I.cand[1][0] = Q.mv_cand[1][0]; I.cand[1][1] = Q.mv_cand[1][1];
    me_best b = {CTU_MAX_DOUBLE, 2147483647.0, 0, 0};
    // the starting point from the reference picture's own motion at the block's centre
    {
      const int mid_x = x + (n >> 1), mid_y = y + (n >> 1);
      const int32_t *rc = B.ref_cu[ref_pic] + ((size_t)(mid_y >> 2) * B.ref_cu_stride + (mid_x >> 2)) * 8;
      if (rc[0] == CU_INTER) {
        int px, py;
        if (rc[5] & 1) { px = rc[1]; py = rc[2]; } else { px = rc[3]; py = rc[4]; }
        if (B.l_size[ref_list] > 0) {
          int col_list = ref_list;
          for (int i = 0; i < B.n_refs; i++) if (B.ref_pocs[i] > B.poc) { col_list = 1; break; }
          if ((rc[5] & (col_list + 1)) == 0) col_list = 1 - col_list;
          const int nb_idx = B.l[ref_list][LX_idx & 15];
          apply_mv_scaling(B.poc, B.ref_pocs[B.l[ref_list][LX_idx & 15]], B.ref_pocs[nb_idx], rc[6 + col_list], &px, &py);
        }
        if (mv_within(B, x, y, n, px, py)) { b.mx = px; b.my = py; }
      }
    }
    { PB_T0();
    me_integer(S, I, b.mx, b.my, b);
    PB_T1(J.W, 3); }
    if (B.fme_level == 0 && b.cost < CTU_MAX_DOUBLE) {
      const int q = n > 32 ? 32 : n, nq = n / q;
      for (int qy = 0; qy < nq; ++qy)
        for (int qx = 0; qx < nq; ++qx)
          ipol_block<PX>((const PX *)B.ref_y[ref_pic], B.ref_stride, P.pic_w, P.pic_h, x + qx * q + (b.mx >> 4), y + qy * q + (b.my >> 4), q, q, 0, 0, 0, 0,
                         T.ry + qy * q * T.rpy + qx * q, T.rpy, nullptr, 0, V->t0);
      b.cost = (double)satd_vs_source(J, x, y, n, T.ry, T.rpy);
      b.cost += b.bits * P.lambda_sqrt;
    }
    for (; ref_list < 2 && active[ref_list]; ++ref_list) {
      LX_idx = idx[ref_list];
      const int cu_mv_cand = select_mv_cand(I.cand, b.mx, b.my, nullptr);
      if (mv_within(B, x, y, n, b.mx, b.my) && b.cost < CTU_MAX_DOUBLE) {
        SERIAL {
          const int e = Q.amvp_size[ref_list];
          pb_cand &u = Q.amvp[ref_list][e];
          u = Q.cur;
          u.m.type = CU_INTER; u.merged = 0; u.skipped = 0;
          u.m.dir = ref_list + 1;
          u.m.ref[ref_list] = LX_idx & 255;
          u.m.mv[ref_list][0] = b.mx; u.m.mv[ref_list][1] = b.my;
          if (ref_list == 0) u.cand0 = (uint8_t)cu_mv_cand; else u.cand1 = (uint8_t)cu_mv_cand;
          u.cost = b.cost; u.bits = b.bits;
          Q.amvp_keys[ref_list][e] = (int8_t)e;
          Q.amvp_size[ref_list] = e + 1;
        }
        CTU_SYNC();
      }
    }
  }
  SERIAL {
    sort_keys(Q.amvp[0], Q.amvp_keys[0], Q.amvp_size[0]);
    sort_keys(Q.amvp[1], Q.amvp_keys[1], Q.amvp_size[1]);
    // best_keys / best_unipred; both lists pointing at the same picture: the list whose runner-up is worse loses its best
    Q.i0 = Q.amvp_size[0] > 0 ? Q.amvp_keys[0][0] : 0;
    Q.i1 = Q.amvp_size[1] > 0 ? Q.amvp_keys[1][0] : 0;
    if (B.bipred && Q.amvp_size[0] > 0 && Q.amvp_size[1] > 0) {
      const int r0 = B.l[0][Q.amvp[0][Q.i0].m.ref[0] & 15], r1 = B.l[1][Q.amvp[1][Q.i1].m.ref[1] & 15];
      if (r0 == r1) {
        const double s0 = Q.amvp_size[0] > 1 ? Q.amvp[0][Q.amvp_keys[0][1]].cost : CTU_MAX_DOUBLE;
        const double s1 = Q.amvp_size[1] > 1 ? Q.amvp[1][Q.amvp_keys[1][1]].cost : CTU_MAX_DOUBLE;
        const int list = (s0 <= s1) ? 1 : 0;
        Q.amvp[list][list ? Q.i1 : Q.i0].cost = CTU_MAX_DOUBLE;
        sort_keys(Q.amvp[list], Q.amvp_keys[list], Q.amvp_size[list]);
        Q.amvp_size[list]--;
        if (list) Q.i1 = Q.amvp_keys[1][0]; else Q.i0 = Q.amvp_keys[0][0];
      }
    }
  }
  CTU_SYNC();
  for (int list = 0; list < 2; ++list) {
    const int n_best = Q.amvp_size[list] < 1 ? Q.amvp_size[list] : 1;
    if (B.fme_level > 0) {
      for (int i = 0; i < n_best; ++i) {
        const int key = Q.amvp_keys[list][i];
        const int LX_idx = Q.amvp[list][key].m.ref[list];
        SERIAL amvp_for(S, J, Q.hm, list, Q.amvp[list][key].m.ref[0], Q.amvp[list][key].m.ref[1], Q.mv_cand);
        CTU_SYNC();
        me_info<PX> I;
        I.J = &J; I.ref_pic = B.l[list][LX_idx & 15]; I.x = x; I.y = y; I.n = n;
        I.cand[0][0] = Q.mv_cand[0][0]; I.cand[0][1] = Q.mv_cand[0][1]; I.cand[1][0] = Q.mv_cand[1][0]; I.cand[1][1] = Q.mv_cand[1][1];
        me_best b = {CTU_MAX_DOUBLE, 2147483647.0, Q.amvp[list][key].m.mv[list][0], Q.amvp[list][key].m.mv[list][1]};
        { PB_T0();
        me_frac(S, I, T.ry, T.rpy, b);
        PB_T1(J.W, 4); }
        const int cu_mv_cand = select_mv_cand(I.cand, b.mx, b.my, nullptr);
        const int extra_bits = list + LX_idx;
        b.cost += extra_bits * P.lambda_sqrt;
        b.bits += extra_bits;
        if (mv_within(B, x, y, n, b.mx, b.my)) SERIAL {
          pb_cand &u = Q.amvp[list][key];
          u.m.mv[list][0] = b.mx; u.m.mv[list][1] = b.my;
          if (list == 0) u.cand0 = (uint8_t)cu_mv_cand; else u.cand1 = (uint8_t)cu_mv_cand;
          u.cost = b.cost; u.bits = b.bits;
        }
        CTU_SYNC();
      }
      SERIAL for (int i = n_best; i < Q.amvp_size[list]; ++i) Q.amvp[list][Q.amvp_keys[list][i]].cost = CTU_MAX_DOUBLE;
      CTU_SYNC();
    }
    SERIAL { sort_keys(Q.amvp[list], Q.amvp_keys[list], Q.amvp_size[list]); Q.amvp_size[list] = n_best; }
    CTU_SYNC();
  }
  const int can_use_bipred = B.slice_type == 0 && B.bipred && n + n >= 16;
  if (can_use_bipred) {
    PB_T0();
    if (Q.amvp_size[0] > 0 && Q.amvp_size[1] > 0) {
      SERIAL {
        pb_cand &u = Q.amvp[2][0];
        const pb_cand &u0 = Q.amvp[0][Q.i0], &u1 = Q.amvp[1][Q.i1];
        u = Q.cur;
        u.m.dir = 3;
        u.m.ref[0] = u0.m.ref[0]; u.m.ref[1] = u1.m.ref[1];
        u.m.mv[0][0] = u0.m.mv[0][0]; u.m.mv[0][1] = u0.m.mv[0][1];
        u.m.mv[1][0] = u1.m.mv[1][0]; u.m.mv[1][1] = u1.m.mv[1][1];
        u.merged = 0; u.skipped = 0;
        // (sic) the predictors of the LAST list fetched price both vectors (search_inter.c:2010-2036)
        for (int reflist = 0; reflist < 2; reflist++) amvp_for(S, J, Q.hm, reflist, u.m.ref[0], u.m.ref[1], Q.mv_cand);
      }
      CTU_SYNC();
      pred_cu(S, J, x, y, n, &Q.amvp[2][0].m, 1, 0, T.ry, T.rpy, T.ru, T.rv, T.rpc);
      const unsigned satd = satd_vs_source(J, x, y, n, T.ry, T.rpy);
      SERIAL {
        pb_cand &u = Q.amvp[2][0];
        double cost = (double)satd;
        double bitcost[2] = {0, 0};
        cost += calc_mvd_cost(P.lambda_sqrt, u.m.mv[0][0], u.m.mv[0][1], 0, Q.mv_cand, &bitcost[0]);
        cost += calc_mvd_cost(P.lambda_sqrt, u.m.mv[1][0], u.m.mv[1][1], 0, Q.mv_cand, &bitcost[1]);
        const int extra_bits = u.m.ref[0] + u.m.ref[1] + 2;
        cost += P.lambda_sqrt * extra_bits;
        if (cost < CTU_MAX_DOUBLE) {
          u.cand0 = (uint8_t)select_mv_cand(Q.mv_cand, u.m.mv[0][0], u.m.mv[0][1], nullptr);
          u.cand1 = (uint8_t)select_mv_cand(Q.mv_cand, u.m.mv[1][0], u.m.mv[1][1], nullptr);
          u.cost = cost;
          u.bits = bitcost[0] + bitcost[1] + extra_bits;
          Q.amvp_keys[2][0] = 0;
          Q.amvp_size[2] = 1;
        }
      }
      CTU_SYNC();
    }
    PB_T1(J.W, 5);
  }
  SERIAL {
    int cs, cp;
    pb_flag_ctx(S, x, y, lx, ly, &cs, &cp);
    const double no_skip_flag = m_fbits(mdl, MI_SKIP + cs, 0);
    const double pred_mode_bits = m_fbits(mdl, MI_PRED_MODE + cp, 0);
    const double total_bits = no_skip_flag + pred_mode_bits;
    for (int i = 0; i < 3; i++)
      if (Q.amvp_size[i] > 0) {
        pb_cand &u = Q.amvp[i][Q.amvp_keys[i][0]];
        u.bits += total_bits;
        u.cost += total_bits * P.lambda_sqrt;
      }
  }
  CTU_SYNC();
  return 0;
}

// uvg_search_cu_inter (search_inter.c:2329-2406): the best of the maps becomes S->pb.cur; -> its cost (MAX_DOUBLE: none)
template <typename PX> CTU_NOINLINE CTU_DEV double search_cu_inter(lds<PX> *S, const job<PX> &J, int L, const cu_target<PX> &T)
{
  pb_state &Q = pbq(S);
  if (search_pu_inter(S, J, L, T)) return Q.merge[0].cost;           // early skip: cost 0
  SERIAL {
    double inter_cost = CTU_MAX_DOUBLE;
    const pb_cand *best = nullptr;
    for (int d = 0; d < 3; ++d)
      if (Q.amvp_size[d] > 0) {
        const pb_cand &u = Q.amvp[d][Q.amvp_keys[d][0]];
        if (u.cost < inter_cost) { best = &u; inter_cost = u.cost; }
      }
    if (Q.merge_size > 0 && Q.merge[Q.merge_keys[0]].cost < inter_cost) { best = &Q.merge[Q.merge_keys[0]]; inter_cost = best->cost; }
    if (best) Q.cur = *best;
    Q.d0 = inter_cost;
  }
  CTU_SYNC();
  return Q.d0;
}

// uvg_encode_mvd in count mode (encode_coding_tree.c:1865-1910): the two models adapt, the rest is bypass; -> the bits of THIS call
CTU_DEV double mvd_bits_update(CTU_LDS uint32_t *m, int update, int mvd_hor, int mvd_ver)
{
  const unsigned ah = (unsigned)iabs_(mvd_hor), av = (unsigned)iabs_(mvd_ver);
  double t = 0.0;
  m_code(m, update, MI_MVD + 0, mvd_hor != 0, t);
  m_code(m, update, MI_MVD + 0, mvd_ver != 0, t);
  if (mvd_hor != 0) m_code(m, update, MI_MVD + 1, ah > 1, t);
  if (mvd_ver != 0) m_code(m, update, MI_MVD + 1, av > 1, t);
  for (int c = 0; c < 2; ++c) {
    const unsigned a = c ? av : ah;
    if (!a) continue;
    if (a > 1) {
      unsigned symbol = a - 2, count = 1;
      int num_bins = 0;
      while (symbol >= (1u << count)) { ++num_bins; symbol -= 1u << count; ++count; }
      ++num_bins;
      num_bins += (int)count;
      t += num_bins;
    }
    t += 1;
  }
  return t;
}
// merge_idx (encode_coding_tree.c:790-806 / :1806-1820)
CTU_DEV void merge_idx_bits(CTU_LDS uint32_t *m, int update, int max_merge, int merge_idx, double &bits)
{
  if (max_merge > 1)
    for (int ui = 0; ui < max_merge - 1; ui++) {
      const int symbol = ui != merge_idx;
      if (ui == 0) m_code(m, update, MI_MERGE_IDX, symbol, bits);
      else bits += 1;
      if (symbol == 0) break;
    }
}
// uvg_encode_inter_prediction_unit in count mode (:769-900) for the CU c at (x, y), n x n, predictors from the history table hm; lane 0.
// NOTE uvg_encode_mvd ASSIGNS the caller's bit count (:1908): what was accumulated before a vector difference is lost.
template <typename PX> CTU_DEV void inter_pu_bits(lds<PX> *S, const job<PX> &J, CTU_LDS uint32_t *m, int update, const pb_cand &c, const int32_t *hm, int x, int y,
                                                  int n, uint32_t split_tree, double &bits_out)
{
  const pb_job &B = *J.pb;
  double bits = 0;
  m_code(m, update, MI_MERGE_FLAG, c.merged, bits);
  if (c.merged) merge_idx_bits(m, update, B.max_merge, c.merge_idx, bits);
  else {
    if (B.slice_type == 0) {
      const int inter_dir = c.m.dir;
      if (n + n > 12) m_code(m, update, MI_INTER_DIR + (7 - ((2 * ilog2_dev(n) + 1) >> 1)), inter_dir == 3, bits);
      if (inter_dir < 3) m_code(m, update, MI_INTER_DIR + 5, inter_dir == 2, bits);
    }
    for (int l = 0; l < 2; l++) {
      if (!(c.m.dir & (1 << l))) continue;
      const int size = B.l_size[l];
      if (size > 1) {
        const int ref_frame = c.m.ref[l];
        m_code(m, update, MI_REF_PIC + 0, ref_frame != 0, bits);
        if (ref_frame > 0 && size > 2) {
          m_code(m, update, MI_REF_PIC + 1, ref_frame > 1, bits);
          if (ref_frame > 1 && size > 3)
            for (int idx = 3; idx < size; idx++) { const int val = ref_frame > idx - 1 ? 1 : 0; bits += 1; if (!val) break; }
        }
      }
      set_cand_ctx(S, x, y, n, split_tree);
      int32_t pred[2][2];
      amvp_for(S, J, hm, l, c.m.ref[0], c.m.ref[1], pred);
      const int which = l == 0 ? c.cand0 : c.cand1;
      const int mvd_hor = to_quarter(c.m.mv[l][0] - pred[which][0]), mvd_ver = to_quarter(c.m.mv[l][1] - pred[which][1]);
      bits_out = mvd_bits_update(m, update, mvd_hor, mvd_ver);
      m_code(m, update, MI_MVP_IDX, which, bits);
    }
  }
  bits_out += bits;
}

// mark_deblocking (search.c:1075-1174) of an n x n inter CU; lane 0
template <typename PX> CTU_DEV void mark_deblocking_inter(lds<PX> *S, int x, int y, int lx, int ly, int n, int is_skip)
{
  if (x) {
    for (int xx = lx; xx < lx + n; xx += 32) {
      for (int yy = ly; yy < ly + n; yy += 4) { cu_at(S, xx, yy)->luma_edges |= 1; cu_at(S, xx, yy)->chroma_edges |= 1; }
      if (is_skip) break;
    }
  } else if (n == 64 && !is_skip) {
    for (int yy = ly; yy < ly + n; yy += 4) { cu_at(S, 32, yy)->luma_edges |= 1; cu_at(S, 32, yy)->chroma_edges |= 1; }
  }
  if (y) {
    for (int yy = ly; yy < ly + n; yy += 32) {
      for (int xx = lx; xx < lx + n; xx += 4) { cu_at(S, xx, yy)->luma_edges |= 2; cu_at(S, xx, yy)->chroma_edges |= 2; }
      if (is_skip) break;
    }
  } else if (n == 64 && !is_skip) {
    for (int xx = lx; xx < lx + n; xx += 4) { cu_at(S, xx, 32)->luma_edges |= 2; cu_at(S, xx, 32)->chroma_edges |= 2; }
  }
}
// lcu_fill_cu_info + lcu_fill_cbf (search.c:314-353, 402-420) + mark_deblocking of the depth's inter CU from lvl[L]; lane 0
template <typename PX> CTU_DEV void place_inter_cu(lds<PX> *S, int L)
{
  const level_state &N = S->lvl[L];
  const int n = 64 >> L, lx = N.x & 63, ly = N.y & 63, l2 = ilog2_dev(n);
  const uint32_t mtt = cu_mtt(N.mode_type_tree, L);
  for (int yy = ly; yy < ly + n; yy += 4)
    for (int xx = lx; xx < lx + n; xx += 4) {
      cu4 *c = cu_at(S, xx, yy);
      const int u = u_idx(xx, yy);
      c->type = CU_INTER; c->log2 = (uint8_t)l2; c->log2_c = (uint8_t)(l2 - 1); c->mode = 0; c->mode_chroma = 0;
      c->luma_edges = 0; c->chroma_edges = 0;
      c->cbf = (uint8_t)(n == 64 ? N.cbf4[((yy - ly) >> 5) * 2 + ((xx - lx) >> 5)] : N.cbf);
      S->scr->tree[(yy >> 2) * 16 + (xx >> 2)] = (uint16_t)N.split_tree;
      S->scr->mtt[(yy >> 2) * 16 + (xx >> 2)] = (uint16_t)mtt;
      icand::unit &m = S->pb.mot[u];
      m.type = N.mot[0]; m.mv[0][0] = N.mot[1]; m.mv[0][1] = N.mot[2]; m.mv[1][0] = N.mot[3]; m.mv[1][1] = N.mot[4]; m.ref[0] = N.mot[5]; m.ref[1] = N.mot[6]; m.dir = N.mot[7];
      for (int k = 0; k < 8; ++k) S->pb.fl[u][k] = N.fl[k];
      if (xx != lx || yy != ly) S->pb.fl[u][3] = 0;                      // root_cbf lives in the CU's first entry only
    }
  mark_deblocking_inter(S, N.x, N.y, lx, ly, n, N.fl[0]);
}

// The chosen inter candidate (S->pb.cur) becomes the CU: the part of search_cu behind the mode decision (search.c:1598-1720):
// quarter-sample rounding, prediction, residual, "merged without residual = skipped", then uvg_mock_encode_coding_unit and
// cu_rd_cost_tr_split_accurate with the models adapting.  Everything the CU is goes to lvl[L].
template <typename PX> CTU_NOINLINE CTU_DEV void finish_inter(lds<PX> *S, const job<PX> &J, int L, const cu_target<PX> &T)
{
  wctx *const V = wv_of(S);
  const params &P = J.P;
  const pb_job &B = *J.pb;
  pb_state &Q = pbq(S);
  level_state &N = S->lvl[L];
  const int n = 64 >> L, x = N.x, y = N.y, lx = x & 63, ly = y & 63;
  const int q = n > 32 ? 32 : n, ntu = n > 32 ? 4 : 1;
  int32_t cbf4[4] = {0, 0, 0, 0};
  int root_cbf = 0;
  { PB_T0();
  if (!Q.cur.skipped) {
    SERIAL {
      if (!Q.cur.merged)
        for (int l = 0; l < 2; ++l)
          if (Q.cur.m.dir & (1 << l)) { Q.cur.m.mv[l][0] = icand::to_quarter_and_back(Q.cur.m.mv[l][0]); Q.cur.m.mv[l][1] = icand::to_quarter_and_back(Q.cur.m.mv[l][1]); }
    }
    CTU_SYNC();
    pred_cu(S, J, x, y, n, &Q.cur.m, 1, 1, T.ry, T.rpy, T.ru, T.rv, T.rpc);
    // the CU is the merge candidate the early-skip test quantised: same prediction, same residual, same RDOQ -- its luma levels are there
    const int luma_done = Q.cur.merged && Q.early_tag == (int)Q.cur.merge_idx + 1;
    const int any = quantize_inter(S, J, x, y, n, T, 1, 1, 0, cbf4, luma_done);
    if (n > 32) root_cbf = (any & 7) != 0;
    const int cbf = (cbf4[0] & 7) != 0 || root_cbf;
    SERIAL { if (Q.cur.merged && !cbf) { Q.cur.merged = 0; Q.cur.skipped = 1; } }
    CTU_SYNC();
  } else {
    // an early skip: the prediction of the test is the reconstruction, there are no levels (quantize_tr_residual zeroes them)
    for (int color = 0; color < 3; ++color) {
      const int c = color != 0, w = n >> c, l2 = ilog2_dev(w);
      int16_t *k = color == 0 ? T.ky : (color == 1 ? T.ku : T.kv);
      const int kp = c ? T.kpc : T.kpy;
      PAR_FOR(e, w * w) k[(e >> l2) * kp + (e & (w - 1))] = 0;
    }
    CTU_SYNC();
  }
  PB_T1(J.W, 7); }
  const pb_cand cu = Q.cur;
  CTU_SYNC();
  PB_T0();
  // ---- bits: uvg_mock_encode_coding_unit (encode_coding_tree.c:1730-1862), update = 1 ----
  SERIAL {
    CTU_LDS uint32_t *const m = LDSP(uint32_t, V->cur);
    double bits = 0;
    split_flag_bits(S, P, V->cur, 1, x, y, lx, ly, n, 0, bits);
    int cs, cp;
    pb_flag_ctx(S, x, y, lx, ly, &cs, &cp);
    m_code(m, 1, MI_SKIP + cs, cu.skipped, bits);
    if (cu.skipped) merge_idx_bits(m, 1, B.max_merge, cu.merge_idx, bits);
    else {
      m_code(m, 1, MI_PRED_MODE + cp, 0, bits);
      inter_pu_bits(S, J, m, 1, Q.cur, Q.hm, x, y, n, N.split_tree, bits);
    }
    Q.d0 = bits * P.lambda;
  }
  CTU_SYNC();
  // ---- cu_rd_cost_tr_split_accurate (search.c:724-986) for an inter CU ----
  const int skip_residual = cu.skipped || cbf4[0] == 0;           // (pred_cu->cbf: the flags of the CU's FIRST transform unit)
  double total = 0;                                               // lane 0's
  if (n > 32) {
    SERIAL { double lb = 0; if (!cu.merged) m_code(LDSP(uint32_t, V->cur), 1, M_ROOT_CBF, (cbf4[0] & 7) != 0, lb); Q.d1 = lb; }
    CTU_SYNC();
  }
  for (int i = 0; i < ntu; ++i) {
    const int ox = (i & 1) * q, oy = (i >> 1) * q;
    const int cbf = cbf4[i], cb_y = cbf & 1, cb_u = (cbf >> 1) & 1, cb_v = (cbf >> 2) & 1;
    if (n > 32) {            // the unit's levels again (quantize_inter left the last unit's in the scratch)
      for (int color = 0; color < 3; ++color) {
        const int c = color != 0, w = q >> c, l2 = ilog2_dev(w);
        const int16_t *k = (color == 0 ? T.ky : (color == 1 ? T.ku : T.kv)) + (oy >> c) * (c ? T.kpc : T.kpy) + (ox >> c);
        const int kp = c ? T.kpc : T.kpy;
        PAR_FOR(e, w * w) lv_of(V, color)[e] = CTU_GLOAD(&k[(e >> l2) * kp + (e & (w - 1))]);
      }
      CTU_SYNC();
    } else if (cu.skipped) {
      PAR_FOR(e, q * q) { lv_of(V, 0)[e] = 0; if (e < (q >> 1) * (q >> 1)) { lv_of(V, 1)[e] = 0; lv_of(V, 2)[e] = 0; } }
      CTU_SYNC();
    }
    ssd_block(S, J, 0, (lx + ox), (ly + oy), q, 0, T.ry + oy * T.rpy + ox, T.rpy);
    ssd_block(S, J, 1, (lx + ox), (ly + oy), q, 1, T.ru + (oy >> 1) * T.rpc + (ox >> 1), T.rpc);
    ssd_block(S, J, 2, (lx + ox), (ly + oy), q, 2, T.rv + (oy >> 1) * T.rpc + (ox >> 1), T.rpc);
    double luma_bits = 0, chroma_bits = 0, coeff_bits_ = 0;
    LANE0 {
      CTU_LDS uint32_t *const m = LDSP(uint32_t, V->cur);
      if (!cu.merged) m_code(m, 1, M_ROOT_CBF, (cbf & 7) != 0, luma_bits);
      if (!skip_residual) {
        m_code(m, 1, M_CBF_CB + 0, cb_u, chroma_bits);
        m_code(m, 1, M_CBF_CR + cb_u, cb_v, chroma_bits);
      }
      if ((cb_u || cb_v) && !skip_residual) m_code(m, 1, M_CBF_LUMA + 0, cb_y, luma_bits);
    }
    WSYNC();
    const unsigned luma_ssd = (unsigned)V->red[0];
    if (cb_y) coeff_bits_ += coeff_bits(S, V->cur, 1, lv_of(V, 0), q, 0);
    const unsigned ssd_u = (unsigned)((unsigned)V->red[1] * P.cw_u), ssd_v = (unsigned)((unsigned)V->red[2] * P.cw_v);
    const unsigned chroma_ssd = ssd_u + ssd_v;
    chroma_bits += coeff_bits(S, V->cur, 1, lv_of(V, 1), q >> 1, 1);
    chroma_bits += coeff_bits(S, V->cur, 1, lv_of(V, 2), q >> 1, 2);
    const double bits = luma_bits + coeff_bits_;
    total += luma_ssd * 1.0 + chroma_ssd * 1.0 + (bits + chroma_bits) * P.lambda;
  }
  SERIAL {
    double cost = Q.d0;
    if (n > 32) cost += total + Q.d1 * P.lambda; else cost += total;
    N.cost = cost; N.type = CU_INTER; N.mode = 0; N.cbf = cbf4[0];
    for (int i = 0; i < 4; ++i) N.cbf4[i] = cbf4[i];
    N.mot[0] = CU_INTER; N.mot[1] = cu.m.mv[0][0]; N.mot[2] = cu.m.mv[0][1]; N.mot[3] = cu.m.mv[1][0]; N.mot[4] = cu.m.mv[1][1];
    N.mot[5] = cu.m.ref[0]; N.mot[6] = cu.m.ref[1]; N.mot[7] = cu.m.dir;
    N.fl[0] = cu.skipped; N.fl[1] = cu.merged; N.fl[2] = cu.merge_idx; N.fl[3] = (uint8_t)root_cbf; N.fl[4] = cu.cand0; N.fl[5] = cu.cand1;
    N.fl[6] = (uint8_t)cu.m.ref[0]; N.fl[7] = (uint8_t)cu.m.ref[1];
  }
  CTU_SYNC();
  PB_T1(J.W, 8);
}

// The CU of depth L (0..3, completely inside the picture) evaluated unsplit: search_cu up to the split loop (search.c:1395-1774) --
// inter search, the intra search unless the inter result is good enough, the winner reconstructed and costed with the models adapting.
// Depths 1..3 go to the depth's candidate buffers on the depth's copy of the entry models (nothing decided is touched); the 64x64 CU
// goes straight into the decided planes / side information (the caller sets it aside before the split).  -> S->lvl[L]
template <typename PX> CTU_NOINLINE CTU_DEV void eval_pb(lds<PX> *S, const job<PX> &J, int L, int can_inter, int can_intra)
{
  const params &P = J.P;
  const pb_job &B = *J.pb;
  pb_state &Q = pbq(S);
  level_state &N = S->lvl[L];
  const int n = 64 >> L, x = N.x, y = N.y, lx = x & 63, ly = y & 63;
  LANE0 S->vsel[CTU_WAVE] = L == 0 ? 3 : 4 - L;
  CTU_SYNC();
  wctx *const V = wv_of(S);
  LANE0 { V->cur = L == 0 ? S->pb.work0 : S->work[L - 1]; Q.hm = S->pb.hmvp_entry[L]; }
  CTU_SYNC();
  const cu_target<PX> T = target_of(S, J, L);
  double cost = CTU_MAX_DOUBLE;
  int type = CU_NOTSET, early_skipped = 0;
  if (can_inter) {
    const double c = search_cu_inter(S, J, L, T);
    if (c < cost) { cost = c; type = CU_INTER; }
    early_skipped = type == CU_INTER && Q.cur.skipped;
  }
  // no intra search when -- rd = 0 only -- the inter cost per sample is below INTRA_THRESHOLD = 8, or after an early skip (search.c:1413-1419).
  const int skip_intra = (P.rd == 0 && type != CU_NOTSET && cost / (double)(n * n) < 8) || (B.early_skip && early_skipped);
  int mode = 0;
  if (can_intra && !skip_intra) {
    PB_T0();
    build_refs(S, P, 0, x, y, lx, ly, n, n);
    search_intra_rough(S, J, x, y, lx, ly, n);
    mode = V->u_mode;
    double intra_cost = V->rs_cand[0];                    // the rough cost of the best mode
    if (intra_cost < cost) {
      // the chroma of the intra candidate decides with its RD cost (search.c:1526-1560, uvg_cu_rd_cost_chroma :625-722): models untouched
      const int cu_ = recon_tu(S, J, 1, x, y, lx, ly, n, mode, 0, T.ru, T.rpc, T.ku, T.kpc, n);
      const int cv_ = recon_tu(S, J, 2, x, y, lx, ly, n, mode, cu_, T.rv, T.rpc, T.kv, T.kpc, n);
      ssd_block(S, J, 1, lx, ly, n, 1, T.ru, T.rpc);
      ssd_block(S, J, 2, lx, ly, n, 2, T.rv, T.rpc);
      double tr_tree_bits = 0, cbits = 0;
      LANE0 {
        CTU_LDS const uint32_t *const m = LDSP(const uint32_t, V->cur);
        tr_tree_bits += m_fbits(m, M_CBF_CB + 0, cu_);
        tr_tree_bits += m_fbits(m, M_CBF_CR + cu_, cv_);
      }
      cbits += coeff_bits(S, V->cur, 0, lv_of(V, 1), n >> 1, 1);
      cbits += coeff_bits(S, V->cur, 0, lv_of(V, 2), n >> 1, 2);
      SERIAL {
        const int ssd = V->red[1] + V->red[2];
        const double bits = tr_tree_bits + cbits;
        Q.d0 = intra_cost + ((double)ssd * 1.0 + bits * P.c_lambda);
      }
      CTU_SYNC();
      intra_cost = Q.d0;
    }
    if (intra_cost < cost) { cost = intra_cost; type = CU_INTRA; }
    PB_T1(J.W, 6);
  }
  if (type == CU_INTRA) {
    if (L == 0) { SERIAL { N.cost = CTU_MAX_DOUBLE; N.type = CU_NOTSET; } CTU_SYNC(); }     // (refused by the host: pu-depth-intra starts below 64)
    else { PB_T0(); eval_cu(S, J, L, 1, mode); PB_T1(J.W, 9); }
  } else if (type == CU_INTER) {
    finish_inter(S, J, L, T);
    if (L == 0 && S->depth_wave != 2) { SERIAL place_inter_cu(S, 0); CTU_SYNC(); }          // (in place; four waves: placed when its split has lost, unpark64_pb)
  } else {
    SERIAL { N.cost = CTU_MAX_DOUBLE; N.type = CU_NOTSET; }
    CTU_SYNC();
  }
  LANE0 S->vsel[CTU_WAVE] = 0;
  CTU_SYNC();
}

// the depth's unsplit candidate becomes the decision (work_tree_copy_up in reverse)
template <typename PX> CTU_NOINLINE CTU_DEV void unpark_pb(lds<PX> *S, const job<PX> &J, int L)
{
  const level_state &N = S->lvl[L];
  if (N.type != CU_INTER) { unpark(S, J, L); return; }
  const int n = 64 >> L, lx = N.x & 63, ly = N.y & 63;
  for (int color = 0; color < 3; ++color) {
    const int c = color != 0, w = n >> c, l2 = ilog2_dev(w), pit = pitch_of(color), spit = c ? LCU_C : LCU;
    PX *D = plane(S, color) + ((ly >> c) + 1) * pit + (lx >> c) + 1;
    int16_t *co = J.coeff + co_off(color) + (ly >> c) * spit + (lx >> c);
    const int off = cand_px_off(L, color);
    PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); D[r * pit + q] = S->cand_px[off + e]; co[r * spit + q] = CTU_GLOAD(&J.W->cand_co[off + e]); }
  }
  SERIAL place_inter_cu(S, L);
  CTU_SYNC();
}

// four waves: the 64x64 candidate, evaluated beside its children into cand64_px / cand64_co, becomes the CTU
template <typename PX> CTU_NOINLINE CTU_DEV void unpark64_pb(lds<PX> *S, const job<PX> &J)
{
  for (int color = 0; color < 3; ++color) {
    const int w = color ? 32 : 64, l2 = color ? 5 : 6, pit = pitch_of(color);
    PX *D = plane(S, color) + pit + 1;
    const PX *from = S->cand64_px + co_off(color);
    const int16_t *cf = S->cand64_co + co_off(color);
    int16_t *co = J.coeff + co_off(color);
    PAR_FOR(e, w * w) { D[(e >> l2) * pit + (e & (w - 1))] = from[e]; co[e] = cf[e]; }
  }
  SERIAL place_inter_cu(S, 0);
  CTU_SYNC();
}

// the 64x64 candidate is set aside while its split is tried; the CTU's side information starts empty again for the children
template <typename PX> CTU_NOINLINE CTU_DEV void save64_pb(lds<PX> *S, const job<PX> &J)
{
  scratch *W = J.W;
  for (int color = 0; color < 3; ++color) {            // four samples / levels per step: 64-bit accesses to the scratch and the level array
    const int w = color ? 32 : 64, l2 = color ? 5 : 6, pit = pitch_of(color);
    CTU_LDS const PX *D = LDSP(const PX, plane(S, color) + pit + 1);
    uint64_t *const spx = (uint64_t *)(W->save_px + co_off(color)), *const sco = (uint64_t *)(W->save_co + co_off(color)), *const co = (uint64_t *)(J.coeff + co_off(color));
    PAR_FOR(e4, (w * w) >> 2) {
      const int e = e4 << 2;
      CTU_LDS const PX *d = D + (e >> l2) * pit + (e & (w - 1));
      spx[e4] = (uint64_t)d[0] | (uint64_t)d[1] << 16 | (uint64_t)d[2] << 32 | (uint64_t)d[3] << 48;
      sco[e4] = CTU_GLOAD(&co[e4]);
      co[e4] = 0;
    }
  }
  PAR_FOR(e, 256) {
    const int lx = (e & 15) * 4, ly = (e >> 4) * 4, u = u_idx(lx, ly);
    W->save_cu[e] = *cu_at(S, lx, ly); W->save_tree[e] = CTU_GLOAD(&W->tree[e]); W->save_tree[256 + e] = CTU_GLOAD(&W->mtt[e]);
    const icand::unit &m = S->pb.mot[u];
    W->save_mot[e][0] = m.type; W->save_mot[e][1] = m.mv[0][0]; W->save_mot[e][2] = m.mv[0][1]; W->save_mot[e][3] = m.mv[1][0]; W->save_mot[e][4] = m.mv[1][1];
    W->save_mot[e][5] = m.ref[0]; W->save_mot[e][6] = m.ref[1]; W->save_mot[e][7] = m.dir;
    for (int k = 0; k < 8; ++k) W->save_fl[e][k] = S->pb.fl[u][k];
  }
  CTU_SYNC();
  PAR_FOR(e, 256) {
    const int lx = (e & 15) * 4, ly = (e >> 4) * 4, u = u_idx(lx, ly);
    cu4 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *cu_at(S, lx, ly) = z;
    W->tree[e] = 0; W->mtt[e] = 0;
    icand::unit &m = S->pb.mot[u];
    m.type = 0; m.mv[0][0] = m.mv[0][1] = m.mv[1][0] = m.mv[1][1] = 0; m.ref[0] = m.ref[1] = 0; m.dir = 0;
    for (int k = 0; k < 8; ++k) S->pb.fl[u][k] = 0;
  }
  CTU_SYNC();
}
template <typename PX> CTU_NOINLINE CTU_DEV void restore64_pb(lds<PX> *S, const job<PX> &J)
{
  scratch *W = J.W;
  for (int color = 0; color < 3; ++color) {
    const int w = color ? 32 : 64, l2 = color ? 5 : 6, pit = pitch_of(color);
    CTU_LDS PX *D = LDSP(PX, plane(S, color) + pit + 1);
    const uint64_t *const spx = (const uint64_t *)(W->save_px + co_off(color)), *const sco = (const uint64_t *)(W->save_co + co_off(color));
    uint64_t *const co = (uint64_t *)(J.coeff + co_off(color));
    PAR_FOR(e4, (w * w) >> 2) {
      const int e = e4 << 2;
      CTU_LDS PX *d = D + (e >> l2) * pit + (e & (w - 1));
      const uint64_t v = CTU_GLOAD(&spx[e4]);
      d[0] = (PX)(v & 0xffff); d[1] = (PX)((v >> 16) & 0xffff); d[2] = (PX)((v >> 32) & 0xffff); d[3] = (PX)(v >> 48);
      co[e4] = CTU_GLOAD(&sco[e4]);
    }
  }
  PAR_FOR(e, 256) { *cu_at(S, (e & 15) * 4, (e >> 4) * 4) = W->save_cu[e]; W->tree[e] = (uint16_t)CTU_GLOAD(&W->save_tree[e]); W->mtt[e] = (uint16_t)CTU_GLOAD(&W->save_tree[256 + e]); }
  CTU_SYNC();
  PAR_FOR(e, 256) {
    const int lx = (e & 15) * 4, ly = (e >> 4) * 4, u = u_idx(lx, ly);
    icand::unit &m = S->pb.mot[u];
    m.type = CTU_GLOAD(&W->save_mot[e][0]); m.mv[0][0] = CTU_GLOAD(&W->save_mot[e][1]); m.mv[0][1] = CTU_GLOAD(&W->save_mot[e][2]); m.mv[1][0] = CTU_GLOAD(&W->save_mot[e][3]);
    m.mv[1][1] = CTU_GLOAD(&W->save_mot[e][4]); m.ref[0] = CTU_GLOAD(&W->save_mot[e][5]); m.ref[1] = CTU_GLOAD(&W->save_mot[e][6]); m.dir = CTU_GLOAD(&W->save_mot[e][7]);
    for (int k = 0; k < 8; ++k) S->pb.fl[u][k] = CTU_GLOAD(&W->save_fl[e][k]);
  }
  CTU_SYNC();
}

// ---- the leaf wave (two-wave build: 128 threads per CTU) -------------------------------------------------------------------------
// The four 4x4 CUs an 8x8 area splits into are intra CUs: they read only what is decided before the area (samples, side information,
// the models at the area's entry + its split flag) and write the area's own decided state -- nothing the evaluation of the unsplit 8x8
// CU reads or writes (that one works on the depth's candidate buffers and its own copy of the models).  So the second wave takes them
// while the walk evaluates the 8x8 CU.  The reference evaluates the CU first and may cut the split short (the pruning test and the
// child-by-child comparison, search.c:1952-1956, 2002-2005); every cut decides "not split" and so does the final comparison whenever a cut
// would have applied (the split's cost only grows child by child), so doing all four changes no decision -- and when the split loses,
// the unsplit candidate is put back over whatever the children left, exactly as after a split that was tried and lost.  On low-QP
// pictures of a low-delay GOP the leaves are 38 % of a CTU's time (profiles/r06_ctu_pb_phases.txt).
template <typename PX> CTU_DEV void leaves_run(lds<PX> *S, const job<PX> &J)          // the leaf wave (or, one wave / host emulation, the walk itself)
{
  const params &P = J.P;
  for (int k = 0; k < 4; ++k) {
    LANE0 {
      level_state &C = S->lvl[4];
      const level_state &N = S->lvl[3];
      C.x = N.x + (k & 1) * 4; C.y = N.y + (k >> 1) * 4; C.has_chroma = k == 3;
    }
    CTU_SYNC();
#if defined(__HIPCC__)
    // the walk has the 8x8 CU's cost by now and the split is already beaten (or was pruned): the rest is not wanted
    if (k > 0 && S->leaf_wave) {
      double sum = 0;
      for (int j = 0; j < k; ++j) sum += S->leaf_cost[j];
      const double lim = __hip_atomic_load(&S->leaf_limit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (sum > lim) { LANE0 { for (int j = k; j < 4; ++j) S->leaf_cost[j] = CTU_MAX_DOUBLE; } CTU_SYNC(); break; }
    }
#endif
    const level_state &C = S->lvl[4];
    const int mode_type_parent = (int)((C.mode_type_tree >> (3 * 2)) & 3);
    const int can_intra = mode_type_parent != 1 && P.depth_max >= 4;          // (an 8x8 area of a picture whose sides are multiples of 8 lies inside it)
#if defined(CTU_LEAF4)
    if (can_intra) { PB_T0(); eval_cu4(S, J); PB_T1(J.W, 15); }          // the register-resident 4x4 CU of ctu_leaf4.h (device builds)
#else
    if (can_intra) { PB_T0(); eval_cu(S, J, 4, 0); PB_T1(J.W, 15); }
#endif
    else { SERIAL { S->lvl[4].cost = CTU_MAX_DOUBLE; S->lvl[4].type = CU_NOTSET; } CTU_SYNC(); }
    LANE0 S->leaf_cost[k] = S->lvl[4].cost;
    CTU_SYNC();
  }
}
template <typename PX> CTU_DEV void post_leaves(lds<PX> *S, const job<PX> &J)
{
#if defined(__HIPCC__)
  CTU_SYNC();
  LANE0 mb_store(&S->req[0], S->req[0] + 1);
#else
  const int me = g_emul_wave;
  g_emul_wave = 1;                      // host emulation: the leaf wave's work happens right here
  leaves_run(S, J);
  g_emul_wave = me;
#endif
}
template <typename PX> CTU_DEV void wait_leaves(lds<PX> *S)
{
#if defined(__HIPCC__)
  while (mb_load(&S->done[0]) != S->req[0]) __builtin_amdgcn_s_sleep(1);
  CTU_SYNC();
#endif
}
#if defined(__HIPCC__)
template <typename PX> CTU_DEV void leaf_worker_loop(lds<PX> *S, const job<PX> &J)
{
  int seen = 0;
  for (;;) {
    int r;
    while ((r = mb_load(&S->req[0])) == seen) __builtin_amdgcn_s_sleep(1);
    if (r < 0) break;
    seen = r;
    CTU_SYNC();
    { PB_T0();
    leaves_run(S, J);
    PB_T1(J.W, 21); }
    CTU_SYNC();
    LANE0 mb_store(&S->done[0], r);
  }
}
#endif

// ---- the depth wave (three-wave build: 192 threads per CTU) ------------------------------------------------------------------------
// A 32x32 or 16x16 CU whose split will be tried needs only what is decided before it -- samples, side information, motion, the models
// and the history table at its entry -- and writes only its depth's candidate buffers, its own copy of the models and lvl[L]: the
// third wave evaluates it (eval_pb, on its own search state `pbx` and the big scratch region) while the walk goes on into the
// children, as the I-picture kernel's depth waves do (ctu_core.h post_eval).  The reference evaluates the CU first and lets its cost cut
// the split short: the pruning test (search.c:1952-1956) before the first child, the child-by-child comparison after each.  Both cuts
// decide "not split", both are re-applied as soon as the evaluation is in -- after every child, and on the way into every node below
// (a pruned ancestor ends the walk under it at once) -- and a split that was started in vain is undone exactly like one that was tried
// and lost: the candidate is put back over the CU's whole area, models and history table restart from the node's entry.
CTU_DEV int mb_of(int L) { return L ? L : 3; }          // the mailbox of depth L (index 3, the 8x8 depth's, is free: the walk evaluates those itself)
template <typename PX> CTU_DEV void post_eval_pb(lds<PX> *S, const job<PX> &J, int L)
{
#if defined(__HIPCC__)
  CTU_SYNC();
  LANE0 mb_store(&S->req[mb_of(L)], S->req[mb_of(L)] + 1);
#else
  const int me = g_emul_wave;
  g_emul_wave = S->depth_wave == 2 && L <= 1 ? 3 : 2;                      // host emulation: the depth wave's work happens right here
  eval_pb(S, J, L, S->lvl[L].can & 1, S->lvl[L].can >> 1);
  g_emul_wave = me;
  S->done[mb_of(L)] = ++S->req[mb_of(L)];
#endif
}
template <typename PX> CTU_DEV bool eval_ready_pb(lds<PX> *S, int L)
{
#if defined(__HIPCC__)
  return __builtin_amdgcn_readfirstlane(mb_load(&S->done[mb_of(L)]) == S->req[mb_of(L)]) != 0;
#else
  return !g_emul_lazy;
#endif
}
template <typename PX> CTU_DEV void wait_eval_pb(lds<PX> *S, int L)
{
#if defined(__HIPCC__)
  while (mb_load(&S->done[mb_of(L)]) != S->req[mb_of(L)]) __builtin_amdgcn_s_sleep(1);
  CTU_SYNC();
#endif
}
#if defined(__HIPCC__)
template <typename PX> CTU_DEV void depth_worker_loop(lds<PX> *S, const job<PX> &J)
{
  int seen[4] = {0, 0, 0, 0};
  // one depth wave (three-wave build): both depths; two: role 2 the 16x16 CUs, role 3 the 32x32 CUs and the 64x64 CU
  const int role = CTU_WAVE, two = S->depth_wave == 2;
  const bool mine1 = !two || role == 3, mine2 = !two || role == 2, mine0 = two && role == 3;
  for (;;) {
    const int r1 = mb_load(&S->req[1]), r2 = mb_load(&S->req[2]), r0 = mb_load(&S->req[3]);
    if (r1 < 0) break;
    int L = -1, r = 0;
    if (mine2 && r2 != seen[2]) { L = 2; r = r2; }            // the deeper request first: the walk comes back for it sooner
    else if (mine1 && r1 != seen[1]) { L = 1; r = r1; }
    else if (mine0 && r0 != seen[3]) { L = 0; r = r0; }
    if (L < 0) { __builtin_amdgcn_s_sleep(2); continue; }
    seen[mb_of(L)] = r;
    CTU_SYNC();
    { PB_T0();
    if (!mb_load(&S->skip_eval[mb_of(L)])) eval_pb(S, J, L, S->lvl[L].can & 1, S->lvl[L].can >> 1);
    PB_T1(J.W, L == 2 ? 19 : (L == 1 ? 20 : 22)); }
    CTU_SYNC();
    LANE0 mb_store(&S->done[mb_of(L)], r);
  }
}
#endif
// the pruning test of a node whose evaluation is in (search.c:1952-1956); lane 0
template <typename PX> CTU_DEV void take_eval_pb(lds<PX> *S, const params &P, int L)
{
  level_state &N = S->lvl[L];
  const double factor = P.qp > 30 ? 1.1 : 1.075;
  N.known = 1;
  N.pending = N.split_bits * P.lambda + N.cost / factor > N.cost;
}

// search_cu (search.c:1299-2221) of a P / B slice as a depth-first loop over the quad tree, in the reference's order
template <typename PX> CTU_DEV void search_ctu_pb(lds<PX> *S, const job<PX> &J)
{
  const params &P = J.P;
  const pb_job &B = *J.pb;
  pb_state &Q = pbq(S);
  SERIAL {
    level_state &R = S->lvl[0];
    R.x = J.x; R.y = J.y; R.split_tree = 0; R.mode_type_tree = 0; R.has_chroma = 1; R.child = 0;
  }
  CTU_SYNC();
  int L = 0, entering = 1;
  double ret = 0;
  for (;;) {
    level_state &N = S->lvl[L];
    const int n = 64 >> L;
    bool decide = false;
    if (entering) {
      const int x = N.x, y = N.y;
      CTU_SYNC();
      if (x >= P.pic_w || y >= P.pic_h) { ret = 0; entering = 0; if (L == 0) break; --L; continue; }
      const int inside = x + n <= P.pic_w && y + n <= P.pic_h;
      const int mode_type_parent = (int)((N.mode_type_tree >> ((L - 1 > 0 ? L - 1 : 0) * 2)) & 3);
      // check_can_use_inter / check_can_use_intra (search.c:1212-1287)
      const int min_wi = 64 >> B.depth_inter_max, min_w = 64 >> P.depth_max;
      const int can_inter = inside && n > 4 && mode_type_parent != 2 &&
                            ((L >= B.depth_inter_min && L <= B.depth_inter_max) || (x & ~(min_wi - 1)) + min_wi > P.pic_w || (y & ~(min_wi - 1)) + min_wi > P.pic_h);
      const int can_intra = inside && mode_type_parent != 1 &&
                            ((L >= P.depth_min && L <= P.depth_max) || (x & ~(min_w - 1)) + min_w > P.pic_w || (y & ~(min_w - 1)) + min_w > P.pic_h);
      // a node under a CU whose evaluation has come in meanwhile and says "not split" (pruned, or the split already costs more): nothing
      // below it is wanted any more -- back to that CU's decision
      if (S->depth_wave && L >= (S->depth_wave == 2 ? 1 : 2)) {
        int A = -1;
        for (int a = S->depth_wave == 2 ? 0 : 1; a < L && A < 0; ++a) {
          level_state &M = S->lvl[a];
          if (M.evalp != 1) continue;          // (2: the 64x64 CU's request is not out yet)
          if (!M.known && eval_ready_pb(S, a)) { SERIAL take_eval_pb(S, P, a); CTU_SYNC(); }
          if (M.known && (M.pending || M.split_cost > M.cost)) A = a;
        }
        if (A >= 0) {
          for (int a = A + 1; a < L; ++a)
            if (S->lvl[a].evalp && !S->lvl[a].known) {          // an evaluation of a node in between is still out: not wanted, but its buffers are in use
#if defined(__HIPCC__)
              LANE0 mb_store(&S->skip_eval[mb_of(a)], 1);
              wait_eval_pb(S, a);
              LANE0 mb_store(&S->skip_eval[mb_of(a)], 0);
              CTU_SYNC();
#endif
            }
          L = A;
#if defined(__HIPCC__) && defined(CTU_PROFILE)
          LANE0 J.W->prof_pb[18] += 1;
#endif
          level_state &M = S->lvl[L];
          const int ntype = M.type;
          CTU_SYNC();
          if (L == 0) {
            // the 64x64 CU: a pruned CTU goes on as the one-wave build leaves it -- the models behind the CU's split flag, the history table
            // of the CTU's entry, the CU in place; a split that lost after its children ran keeps the children's models (search_ctu_pb below)
            if (M.pending) { copy_models(S->cur, S->cur64); SERIAL { for (int i = 0; i < 41; ++i) S->pb.hmvp[i] = S->pb.hmvp_entry[0][i]; } CTU_SYNC(); }
            if (ntype != CU_NOTSET) { PB_T0(); unpark64_pb(S, J); PB_T1(J.W, 10); }
            ret = M.cost;
            break;
          }
          copy_models(S->cur, S->work[L - 1]);
          SERIAL { for (int i = 0; i < 41; ++i) S->pb.hmvp[i] = S->pb.hmvp_entry[L][i]; if (ntype == CU_INTER) hmvp_add(S->pb.hmvp, M.mot); }
          CTU_SYNC();
          if (ntype != CU_NOTSET) { PB_T0(); unpark_pb(S, J, L); PB_T1(J.W, 10); }
          ret = M.cost;
          entering = 0;
          --L;
          continue;
        }
      }
      if (n == 4) {
        // a 4x4 CU: intra only, nothing to split, no history entry
        if (can_intra) {
          PB_T0();
#if defined(CTU_LEAF4)
          eval_cu4(S, J);
#else
          eval_cu(S, J, L, 0);
#endif
          PB_T1(J.W, 15);
        } else { SERIAL { N.cost = CTU_MAX_DOUBLE; N.type = CU_NOTSET; } CTU_SYNC(); }
        ret = N.cost; entering = 0; --L; continue;
      }
      SERIAL {
        for (int i = 0; i < 41; ++i) S->pb.hmvp_entry[L][i] = S->pb.hmvp[i];
        N.type = CU_NOTSET; N.cost = CTU_MAX_DOUBLE; N.pending = 0; N.evalp = 0; N.known = 1;
      }
      CTU_SYNC();
      if (S->depth_wave && (L == 1 || L == 2 || (L == 0 && S->depth_wave == 2)) && P.depth_max >= 4) {
        // a 32x32 / 16x16 CU with the depth wave (four waves: the 64x64 CU too): evaluated there from now on, the walk goes into the children
        const bool will_eval = inside && (can_inter || can_intra);
        if (will_eval) copy_models(L == 0 ? S->pb.work0 : S->work[L - 1], S->cur);
        SERIAL {
          double split_bits = 0;
          split_flag_bits(S, P, S->cur, 1, x, y, x & 63, y & 63, n, 1, split_bits);
          N.split_bits = split_bits;
          N.split_cost = split_bits * P.lambda;
          N.child = 0;
          N.can = can_inter | can_intra << 1;
          N.evalp = will_eval; N.known = !will_eval;
          level_state &C = S->lvl[L + 1];
          const uint32_t mode_type = (uint32_t)mode_type_parent;          // (MODE_TYPE_INFER only at n == 8)
          C.split_tree = N.split_tree | 1u << (L * 3);
          C.mode_type_tree = N.mode_type_tree | mode_type << (L * 2);
          C.x = N.x; C.y = N.y; C.has_chroma = 1;
        }
        CTU_SYNC();
        if (L == 0) copy_models(S->cur64, S->cur);          // (what a pruned CTU goes on with)
        // the 64x64 CU's request goes out behind the first 32x32 CU's: the wave that takes both takes the deeper one first, and the walk
        // comes back for that one long before it needs the 64x64 CU's cost
        if (will_eval && L > 0) post_eval_pb(S, J, L);
        if (L == 1 && S->lvl[0].evalp == 2) { SERIAL S->lvl[0].evalp = 1; CTU_SYNC(); post_eval_pb(S, J, 0); }
        if (L == 0 && will_eval) { SERIAL N.evalp = 2; CTU_SYNC(); }
        ++L;
        continue;
      }
      if (n == 8 && S->leaf_wave && P.depth_max >= 4) {
        // an 8x8 area with the leaf wave: its four 4x4 CUs go there now, the unsplit CU is evaluated here meanwhile
        const bool will_eval = inside && (can_inter || can_intra);
        if (will_eval) copy_models(S->work[L - 1], S->cur);
        SERIAL {
          double split_bits = 0;
          split_flag_bits(S, P, S->cur, 1, x, y, x & 63, y & 63, n, 1, split_bits);
          N.split_bits = split_bits;
          N.split_cost = split_bits * P.lambda;
          N.child = 0;
          level_state &C = S->lvl[L + 1];
          const int cond_infer = mode_type_parent == 0;
          const uint32_t mode_type = cond_infer ? 2u : (uint32_t)mode_type_parent;
          C.split_tree = N.split_tree | 1u << (L * 3);
          C.mode_type_tree = N.mode_type_tree | mode_type << (L * 2);
        }
        LANE0 S->leaf_limit = CTU_MAX_DOUBLE;
        post_leaves(S, J);
        if (will_eval) { PB_T0(); eval_pb(S, J, L, can_inter, can_intra); PB_T1(J.W, 23); }
#if defined(__HIPCC__)
        // the CU's cost is in: what the split may cost at most before it has lost (pruned: nothing) -- the leaf wave stops there
        LANE0 {
          const double factor = P.qp > 30 ? 1.1 : 1.075;
          const bool pruned = N.split_bits * P.lambda + N.cost / factor > N.cost;
          __hip_atomic_store(&S->leaf_limit, pruned ? -1.0 : N.cost - N.split_bits * P.lambda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#endif
        { PB_T0(); wait_leaves(S); PB_T1(J.W, 16); }
        SERIAL {
          const double factor = P.qp > 30 ? 1.1 : 1.075;
          N.pending = N.split_bits * P.lambda + N.cost / factor > N.cost;          // pruned (search.c:1952-1956): the reference would not have tried the split
          if (!N.pending)
            for (int k = 0; k < 4; ++k) {
              N.split_cost += S->leaf_cost[k];
              if (N.split_cost > N.cost) break;                                      // (where the reference stops: the split has lost)
            }
        }
        CTU_SYNC();
        decide = true;
      } else {
      if (inside && (can_inter || can_intra)) {
        copy_models(L == 0 ? S->pb.work0 : S->work[L - 1], S->cur);
        { PB_T0(); eval_pb(S, J, L, can_inter, can_intra); PB_T1(J.W, 22); }
      }
      const int ntype = N.type;
      const double ncost = N.cost;
      const int can_split = (ntype == CU_NOTSET || L < P.depth_max) && n > 4;
      if (!can_split) {
        // (cannot happen with pu-depth-intra max = 4: every CU above 4x4 may split)
        if (L > 0) { copy_models(S->cur, S->work[L - 1]); if (ntype != CU_NOTSET) unpark_pb(S, J, L); }
        SERIAL { if (ntype == CU_INTER) hmvp_add(S->pb.hmvp, N.mot); }        // (an intra CU adds nothing; N.mot is only the depth's last INTER candidate)
        CTU_SYNC();
        ret = ncost; entering = 0; if (L == 0) break; --L; continue;
      }
      // the split: its flag on the CU's entry models (cur still holds them), the pruning test, then the children
      SERIAL {
        double split_bits = 0;
        split_flag_bits(S, P, S->cur, 1, x, y, x & 63, y & 63, n, 1, split_bits);
        N.split_bits = split_bits;
        const double factor = P.qp > 30 ? 1.1 : 1.075;
        N.pending = split_bits * P.lambda + N.cost / factor > N.cost;          // pruned (search.c:1952-1956)
        N.split_cost = split_bits * P.lambda;
        N.child = 0;
        level_state &C = S->lvl[L + 1];
        const int cond_infer = mode_type_parent == 0 && n == 8;
        const uint32_t mode_type = cond_infer ? 2u : (uint32_t)mode_type_parent;
        C.split_tree = N.split_tree | 1u << (L * 3);
        C.mode_type_tree = N.mode_type_tree | mode_type << (L * 2);
        C.x = N.x; C.y = N.y; C.has_chroma = (n >> 1) == 4 ? 0 : 1;
      }
      CTU_SYNC();
      if (N.pending) decide = true;
      else {
        if (L == 0 && ntype != CU_NOTSET) { PB_T0(); save64_pb(S, J); PB_T1(J.W, 10); }
        ++L;
        continue;
      }
      }
    }
    if (!decide) {
      // a child of N came back with `ret`
      if (N.evalp == 1 && !N.known && eval_ready_pb(S, L)) { SERIAL take_eval_pb(S, P, L); CTU_SYNC(); }
      SERIAL {
        N.split_cost += ret;
        const int k = N.child;
        V_flag(S) = (N.known && (N.pending || N.split_cost > N.cost)) || k == 3;
        N.child = k + 1;
        if (!V_flag(S)) {
          level_state &C = S->lvl[L + 1];
          const int h = n >> 1, k1 = k + 1;
          C.x = N.x + (k1 & 1) * h; C.y = N.y + (k1 >> 1) * h;
          C.has_chroma = h == 4 ? (k1 == 3) : 1;
        }
      }
      CTU_SYNC();
      if (!V_flag(S)) { ++L; entering = 1; continue; }
      if (N.evalp && !N.known) { PB_T0(); wait_eval_pb(S, L); PB_T1(J.W, 17); SERIAL take_eval_pb(S, P, L); CTU_SYNC(); }          // the CU's own cost is needed now
    }
    // the decision between the CU and its split
    const bool pruned = N.pending != 0;
    const bool split_wins = !pruned && N.split_cost < N.cost;
    const int ntype = N.type;
    CTU_SYNC();
    if (split_wins) {
      SERIAL N.cost = N.split_cost;
      CTU_SYNC();
    } else {
      if (L > 0) {
        copy_models(S->cur, S->work[L - 1]);             // post_search_cabac
        SERIAL { for (int i = 0; i < 41; ++i) S->pb.hmvp[i] = S->pb.hmvp_entry[L][i]; if (ntype == CU_INTER) hmvp_add(S->pb.hmvp, N.mot); }      // (uvg_hmvp_add_mv ignores an intra CU)
        CTU_SYNC();
        if (ntype != CU_NOTSET) { PB_T0(); unpark_pb(S, J, L); PB_T1(J.W, 10); }
      } else if (S->depth_wave == 2) {
        // the 64x64 CU of the four-wave build (evaluated beside its children): see the abort above
        if (pruned) { copy_models(S->cur, S->cur64); SERIAL { for (int i = 0; i < 41; ++i) S->pb.hmvp[i] = S->pb.hmvp_entry[0][i]; } CTU_SYNC(); }
        if (ntype != CU_NOTSET) { PB_T0(); unpark64_pb(S, J); PB_T1(J.W, 10); }
      } else if (!pruned && ntype != CU_NOTSET) { PB_T0(); restore64_pb(S, J); PB_T1(J.W, 10); }
    }
    ret = N.cost;
    entering = 0;
    if (L == 0) break;
    --L;
  }
}

// ====================================================================== CTU in / out, the deblocking side effect, the coder ======
// init_lcu_t's inter part: the motion and flags of the row above, the column to the left and the corner; the row's history table
template <typename PX> CTU_NOINLINE CTU_DEV void load_ctu_pb(lds<PX> *S, const job<PX> &J)
{
  const params &P = J.P;
  const pb_job &B = *J.pb;
  
  const int x = J.x, y = J.y, W = P.pic_w, H = P.pic_h;
  BLK_FOR(i, 17 * 17 + 1) {
    icand::unit &m = S->pb.mot[i];
    m.type = 0; m.mv[0][0] = m.mv[0][1] = m.mv[1][0] = m.mv[1][1] = 0; m.ref[0] = m.ref[1] = 0; m.dir = 0;
    if (i < 17 * 17) for (int k = 0; k < 8; ++k) S->pb.fl[i][k] = 0;
  }
  BLK_SYNC();
  BLK_FOR(i, 33) {
    int ax, ay, ok;
    if (i == 0) { ax = x - 4; ay = y - 4; ok = x > 0 && y > 0; }
    else if (i <= 16) { ax = x + (i - 1) * 4; ay = y - 4; ok = y > 0 && ax < W; }
    else { ax = x - 4; ay = y + (i - 17) * 4; ok = x > 0 && ay < H; }
    if (ok) {
      const size_t at = (size_t)(ay >> 2) * J.cu_stride + (ax >> 2);
      const uvghip_scu_t *s = &J.cu_tab[at];
      const uvghip_inter4_t *f = &B.inter4[at];
      const int u = u_idx(ax - x, ay - y);
      icand::unit &m = S->pb.mot[u];
      m.type = s->type;
      if (s->type == CU_INTER) {
        m.mv[0][0] = s->mv[0][0]; m.mv[0][1] = s->mv[0][1]; m.mv[1][0] = s->mv[1][0]; m.mv[1][1] = s->mv[1][1];
        m.ref[0] = f->mv_ref0; m.ref[1] = f->mv_ref1; m.dir = s->mv_dir;
        S->pb.fl[u][0] = f->skipped; S->pb.fl[u][1] = f->merged; S->pb.fl[u][2] = f->merge_idx; S->pb.fl[u][3] = f->root_cbf; S->pb.fl[u][4] = f->mv_cand0; S->pb.fl[u][5] = f->mv_cand1;
        S->pb.fl[u][6] = f->mv_ref0; S->pb.fl[u][7] = f->mv_ref1;
      }
    }
  }
  BLK_FOR(i, 41) {
    const int32_t v = J.x > 0 ? B.hmvp_rows[(size_t)(y >> 6) * 41 + i] : 0;       // the row's table starts empty (encoderstate.c:1021-1028)
    S->pb.hmvp[i] = v; S->pb.hmvp_coder[i] = v;
  }
  if (BLK_TID == 0) for (int inst = 0; inst < 1 + S->depth_wave; ++inst) {          // the walk's search state and the depth waves'
    pb_state &Q = inst == 0 ? S->pb : (inst == 1 ? S->pbx : S->pbx2);
    Q.amvp_key[0] = Q.amvp_key[1] = Q.amvp_key[2] = -1; Q.amvp_have[0] = Q.amvp_have[1] = 0;
    Q.colc_idx[0] = Q.colc_idx[1] = -1;
    Q.hm = S->pb.hmvp;
    icand::frame_ctx &f = Q.f;
    f.x = f.y = f.w = f.h = 0;
    f.poc = B.poc; f.is_b = B.slice_type == 0; f.pic_w = W; f.pic_h = H;
    f.tmvp = B.tmvp; f.max_cands = B.max_merge; f.mer_level = B.merge_level; f.wpp = P.wpp;
    f.n_refs = B.n_refs;
    for (int i = 0; i < 16; ++i) f.ref_pocs[i] = B.ref_pocs[i];
    f.l_size[0] = B.l_size[0]; f.l_size[1] = B.l_size[1];
    for (int l = 0; l < 2; ++l) for (int i = 0; i < 8; ++i) f.l[l][i] = B.l[l][i];
    f.split_tree = 0;
  }
  BLK_SYNC();
}

// copy_lcu_to_cu_data's inter part: motion into the side information the filters read, the second table, the trees
template <typename PX> CTU_NOINLINE CTU_DEV void store_ctu_pb(lds<PX> *S, const job<PX> &J)
{
  const params &P = J.P;
  const pb_job &B = *J.pb;
  pb_state &Q = pbq(S);
  const int x = J.x, y = J.y, W = P.pic_w, H = P.pic_h;
  BLK_FOR(e, 256) {
    const int lx = (e & 15) * 4, ly = (e >> 4) * 4;
    if (x + lx < W && y + ly < H) {
      const size_t at = (size_t)((y + ly) >> 2) * J.cu_stride + ((x + lx) >> 2);
      const int u = u_idx(lx, ly);
      const icand::unit &m = S->pb.mot[u];
      uvghip_inter4_t f;
      f.skipped = S->pb.fl[u][0]; f.merged = S->pb.fl[u][1]; f.merge_idx = S->pb.fl[u][2]; f.root_cbf = S->pb.fl[u][3]; f.mv_cand0 = S->pb.fl[u][4]; f.mv_cand1 = S->pb.fl[u][5];
      f.mv_ref0 = (uint8_t)m.ref[0]; f.mv_ref1 = (uint8_t)m.ref[1];
      if (cu_at(S, lx, ly)->type == CU_INTER) {
        uvghip_scu_t *s = &J.cu_tab[at];
        s->mv[0][0] = m.mv[0][0]; s->mv[0][1] = m.mv[0][1]; s->mv[1][0] = m.mv[1][0]; s->mv[1][1] = m.mv[1][1];
        s->mv_dir = (uint8_t)m.dir;
        // state->frame->ref_LX[l][mv_ref[l]]: the deblocking filter compares the pictures two vectors point to (filter.c:770-773)
        s->ref_id[0] = (int16_t)((m.dir & 1) ? B.l[0][m.ref[0] & 15] : -1);
        s->ref_id[1] = (int16_t)((m.dir & 2) ? B.l[1][m.ref[1] & 15] : -1);
      } else { memset(&f, 0, sizeof f); }
      B.inter4[at] = f;
      if (B.trees) B.trees[at] = (uint32_t)CTU_GLOAD(&J.W->tree[e]) | (uint32_t)CTU_GLOAD(&J.W->mtt[e]) << 16;
    }
  }
  BLK_SYNC();
}

// What filter_deblock_edge_luma does to the picture's side information on its way (filter.c:745-765): where the boundary strength comes
// from motion -- both sides inter, no coded luma residual at the edge, a B slice or a bi-predicted side -- the vectors of the lists a
// side does not use are zeroed in the cu_array.  The entries feed the coder's history table, the next CTUs' neighbour rows and later
// pictures.  uvg_filter_deblock_lcu (filter.c:1372-1380) runs between a CTU's search and its coding tree and visits: the CTU's
// vertical edges, then the horizontal edges of the two 4-columns left of the CTU and of the CTU without its last two 4-columns
// (unless it is the picture's last).  Idempotent, and no condition depends on a vector: the order inside does not matter.
template <typename PX> CTU_DEV void deblock_zeroes_at(const job<PX> &J, int bx, int by, int dir_hor)
{
  const params &P = J.P;
  if ((!dir_hor && bx == 0) || (dir_hor && by == 0)) return;
  if (bx >= P.pic_w || by >= P.pic_h) return;
  uvghip_scu_t *q = &J.cu_tab[(size_t)(by >> 2) * J.cu_stride + (bx >> 2)];
  if (!(q->luma_edges & (dir_hor ? 2 : 1))) return;
  uvghip_scu_t *p = dir_hor ? q - J.cu_stride : q - 1;
  if (q->type != CU_INTER || p->type != CU_INTER) return;
  if ((q->cbf & 1) || (p->cbf & 1)) return;
  if (!(p->mv_dir == 3 || q->mv_dir == 3 || J.pb->slice_type == 0)) return;
  for (int l = 0; l < 2; ++l) {
    if (!(q->mv_dir & (1 << l))) { q->mv[l][0] = 0; q->mv[l][1] = 0; }
    if (!(p->mv_dir & (1 << l))) { p->mv[l][0] = 0; p->mv[l][1] = 0; }
  }
}
template <typename PX> CTU_NOINLINE CTU_DEV void deblock_zeroes_unused_vectors(lds<PX> *S, const job<PX> &J)
{
  const params &P = J.P;
  const int x = J.x, y = J.y, W = P.pic_w, H = P.pic_h;
  BLK_FOR(e, 256) deblock_zeroes_at(J, x + (e & 15) * 4, y + (e >> 4) * 4, 0);
  BLK_SYNC();
  if (x > 0) { BLK_FOR(e, 32) deblock_zeroes_at(J, x - 8 + (e & 1) * 4, y + (e >> 1) * 4, 1); }
  BLK_SYNC();
  BLK_FOR(e, 256) {
    const int bx = x + (e & 15) * 4, by = y + (e >> 4) * 4;
    if ((bx & 63) >= 56 && bx < W - 8) continue;
    deblock_zeroes_at(J, bx, by, 1);
  }
  BLK_SYNC();
  // the CTU's table again, as the coder will see the picture: its own units and the neighbours it reads
  BLK_FOR(i, 17 * 17) {
    const int lx = (i % 17) * 4 - 4, ly = (i / 17) * 4 - 4;
    const int ax = x + lx, ay = y + ly;
    if (ax < 0 || ay < 0 || ax >= W || ay >= H) continue;
    const uvghip_scu_t *s = &J.cu_tab[(size_t)(ay >> 2) * J.cu_stride + (ax >> 2)];
    if (s->type != CU_INTER) continue;
    icand::unit &m = S->pb.mot[i];
    if (m.type != CU_INTER) continue;
    m.mv[0][0] = s->mv[0][0]; m.mv[0][1] = s->mv[0][1]; m.mv[1][0] = s->mv[1][0]; m.mv[1][1] = s->mv[1][1];
  }
  BLK_SYNC();
  // ... and for the pictures that will refer to this one (ref_cu layout)
  if (J.pb->motion_out) {
    const pb_job &B = *J.pb;
    BLK_FOR(e, 256) {
      const int lx = (e & 15) * 4, ly = (e >> 4) * 4;
      if (x + lx >= W || y + ly >= H) continue;
      const size_t at = (size_t)((y + ly) >> 2) * J.cu_stride + ((x + lx) >> 2);
      const icand::unit &m = S->pb.mot[u_idx(lx, ly)];
      int32_t *o = B.motion_out + at * 8;
      o[0] = cu_at(S, lx, ly)->type;
      const bool inter = o[0] == CU_INTER;
      o[1] = inter ? m.mv[0][0] : 0; o[2] = inter ? m.mv[0][1] : 0; o[3] = inter ? m.mv[1][0] : 0; o[4] = inter ? m.mv[1][1] : 0;
      o[5] = inter ? m.dir : 0;
      o[6] = inter && (m.dir & 1) ? B.ref_pocs[B.l[0][m.ref[0] & 15]] : -1;
      o[7] = inter && (m.dir & 2) ? B.ref_pocs[B.l[1][m.ref[1] & 15]] : -1;
    }
  }
  BLK_SYNC();
}

// uvg_encode_coding_tree (encode_coding_tree.c:1365-1727) over the decided CTU of a P / B slice: which models see which bins, and the
// history table the coder keeps (uvg_hmvp_add_mv after every inter CU, :1482 / :1618)
template <typename PX> CTU_NOINLINE CTU_DEV void coder_pass_pb(lds<PX> *S, const job<PX> &J)
{
  wctx *const V = wv_of(S);
  const params &P = J.P;
  const pb_job &B = *J.pb;
  pb_state &Q = pbq(S);
  for (int z = 0; z < 256; ++z) {
    const int lx = z_to_x(z) * 4, ly = z_to_x(z >> 1) * 4;
    const int x = J.x + lx, y = J.y + ly;
    if (x >= P.pic_w || y >= P.pic_h) continue;
    const cu4 *c = cu_at(S, lx, ly);
    const int n = 1 << c->log2;
    if ((lx & (n - 1)) || (ly & (n - 1))) continue;
    const int u0 = u_idx(lx, ly);
    const int is_inter = c->type == CU_INTER;
    const int sep = n == 4, last4 = sep && (lx & 4) && (ly & 4);
    const int L = 6 - c->log2;
    const int mode_type_curr = (int)((CTU_GLOAD(&J.W->mtt[(ly >> 2) * 16 + (lx >> 2)]) >> (L * 2)) & 3);
    const int skipped = is_inter && S->pb.fl[u0][0];
    CTU_SYNC();
    // ---- the CU's header ----
    LANE0 {
      CTU_LDS uint32_t *const m = LDSP(uint32_t, S->coder);
      double dummy = 0;
      for (int d = 0; (64 >> d) > n; ++d) {
        const int s = 64 >> d;
        if (!(lx & (s - 1)) && !(ly & (s - 1))) split_flag_bits(S, P, S->coder, 1, x, y, lx, ly, s, 1, dummy);
      }
      split_flag_bits(S, P, S->coder, 1, x, y, lx, ly, n, 0, dummy);
      int cs, cp;
      pb_flag_ctx(S, x, y, lx, ly, &cs, &cp);
      if (n != 4 && mode_type_curr != 2) m_code(m, 1, MI_SKIP + cs, skipped, dummy);
      if (skipped) {
        hmvp_add(S->pb.hmvp_coder, (const int32_t *)&S->pb.mot[u0]);
        merge_idx_bits(m, 1, B.max_merge, S->pb.fl[u0][2], dummy);
      } else {
        if (n != 4 && mode_type_curr == 0) m_code(m, 1, MI_PRED_MODE + cp, !is_inter, dummy);
        if (is_inter) {
          pb_cand &k = Q.cur;                     // (free after the search)
          k.m = S->pb.mot[u0];
          k.skipped = 0; k.merged = S->pb.fl[u0][1]; k.merge_idx = S->pb.fl[u0][2]; k.cand0 = S->pb.fl[u0][4]; k.cand1 = S->pb.fl[u0][5];
          const uint32_t tree = CTU_GLOAD(&J.W->tree[(ly >> 2) * 16 + (lx >> 2)]);
          inter_pu_bits(S, J, m, 1, k, S->pb.hmvp_coder, x, y, n, tree, dummy);
          hmvp_add(S->pb.hmvp_coder, (const int32_t *)&S->pb.mot[u0]);
          const int has_coeffs = S->pb.fl[u0][3] || c->cbf;
          if (!k.merged) m_code(m, 1, M_ROOT_CBF, has_coeffs, dummy);
        } else {
          luma_mode_bits(S, S->coder, 1, x, y, lx, ly, n, c->mode, dummy);
          if (!sep) chroma_mode_bits(S->coder, 1, c->mode_chroma, c->mode, dummy);
        }
      }
    }
    CTU_SYNC();
    if (skipped) continue;
    if (is_inter && !(S->pb.fl[u0][3] || c->cbf)) continue;
    // ---- the transform units ----
    const int tus = n == 64 ? 4 : 1, tn = n == 64 ? 32 : n;
    for (int tu = 0; tu < tus; ++tu) {
      const int tlx = lx + (tu & 1) * 32, tly = ly + (tu >> 1) * 32;
      const cu4 *t = cu_at(S, tlx, tly);
      {
        const int16_t *co = J.coeff + tly * LCU + tlx;
        const int l2 = ilog2_dev(tn);
        PAR_FOR(e, tn * tn) lv_of(V, 0)[e] = CTU_GLOAD(&co[(e >> l2) * LCU + (e & (tn - 1))]);
        if (!sep || last4) {
          const int cw = sep ? 4 : tn >> 1, cl2 = ilog2_dev(cw);
          const int cbx = (sep ? (tlx & ~7) : tlx) >> 1, cby = (sep ? (tly & ~7) : tly) >> 1;
          PAR_FOR(e, cw * cw) {
            lv_of(V, 1)[e] = CTU_GLOAD(&J.coeff[4096 + (cby + (e >> cl2)) * LCU_C + cbx + (e & (cw - 1))]);
            lv_of(V, 2)[e] = CTU_GLOAD(&J.coeff[5120 + (cby + (e >> cl2)) * LCU_C + cbx + (e & (cw - 1))]);
          }
        }
      }
      CTU_SYNC();
      {
        uint32_t *m = S->coder;
        double dummy = 0;
        const int cb_y = t->cbf & 1, cb_u = (t->cbf >> 1) & 1, cb_v = (t->cbf >> 2) & 1;
        LANE0 {
          if (!sep) {
            m_code(m, 1, M_CBF_CB + 0, cb_u, dummy);
            m_code(m, 1, M_CBF_CR + cb_u, cb_v, dummy);
          }
          // encode_transform_coeff :700-716: an inter CU that is one transform unit with no chroma residual implies its luma flag
          if (!is_inter || n == 64 || cb_u || cb_v) m_code(m, 1, M_CBF_LUMA + 0, cb_y, dummy);
        }
        WSYNC();
        if (cb_y) (void)coeff_bits(S, m, 1, lv_of(V, 0), tn, 0);
        if (!sep) {
          if (cb_u) (void)coeff_bits(S, m, 1, lv_of(V, 1), tn >> 1, 1);
          if (cb_v) (void)coeff_bits(S, m, 1, lv_of(V, 2), tn >> 1, 2);
        } else if (last4) {
          const cu4 *a = cu_at(S, lx & ~7, ly & ~7);
          const int au = (a->cbf >> 1) & 1, av = (a->cbf >> 2) & 1;
          LANE0 {
            chroma_mode_bits(m, 1, c->mode_chroma, c->mode, dummy);
            m_code(m, 1, M_CBF_CB + 0, au, dummy);
            m_code(m, 1, M_CBF_CR + au, av, dummy);
          }
          WSYNC();
          if (au) (void)coeff_bits(S, m, 1, lv_of(V, 1), 4, 1);
          if (av) (void)coeff_bits(S, m, 1, lv_of(V, 2), 4, 2);
        }
      }
      CTU_SYNC();
    }
  }
}

// one CTU of a P / B picture, start to finish (one wave)
template <typename PX> CTU_DEV void run_ctu_pb(lds<PX> *S, const job<PX> &J)
{
#if defined(__HIPCC__) && defined(CTU_PROFILE)
  BLK_FOR(i, 24) J.W->prof_pb[i] = 0;
  BLK_FOR(i, 4 * 32) J.W->prof[i >> 5][i & 31] = 0;
  S->prof_w = J.W;
#endif
  PB_T0();
  { PB_T0();
  static_assert(sizeof(icand::unit) == 32, "scratch::pb_mot holds icand::unit as eight int32");
  if (BLK_TID == 0) { S->pb.mot = reinterpret_cast<icand::unit *>(J.W->pb_mot); S->pb.fl = J.W->pb_fl; }
  setup_waves(S, J.W);
#if defined(CTU_LEAF4)
  leaf_tables(S, J.P);
#endif
  BLK_FOR(k, 4) S->wv[k].rq_root = 0;
  if (BLK_TID == 0) {
#if defined(__HIPCC__)
    S->leaf_wave = BLK_NT > 64;           // the two-wave build
#else
    S->leaf_wave = g_emul_leafwave;
#endif
    S->vsel[1] = 0;                       // the leaf wave works on the 4x4 scratch (the walk's wave picks its scratch per depth)
#if defined(__HIPCC__)
    S->depth_wave = BLK_NT > 192 ? 2 : (BLK_NT > 128 ? 1 : 0);         // the three- / four-wave build
#else
    S->depth_wave = g_emul_depthwave;
#endif
    for (int k = 0; k < 4; ++k) S->skip_eval[k] = 0;
  }
  build_scans(S);
  BLK_SYNC();
  load_ctu(S, J);
  load_ctu_pb(S, J);
  BLK_FOR(e, 6144) J.coeff[e] = 0;
  BLK_SYNC();
  PB_T1(J.W, 11); }
#if defined(__HIPCC__)
  if (CTU_WAVE == 0) {
#endif
    search_ctu_pb(S, J);
    PAR_FOR(i, NMODELS) J.models_out[NMODELS + i] = S->cur[i];
    PAR_FOR(i, NMX - NMODELS) J.pbm_out[(NMX - NMODELS) + i] = S->cur[NMODELS + i];
#if defined(__HIPCC__)
    LANE0 { if (S->leaf_wave) mb_store(&S->req[0], -1); if (S->depth_wave) mb_store(&S->req[1], -1); }
  } else if (CTU_WAVE == 1) leaf_worker_loop(S, J);
  else depth_worker_loop(S, J);
#endif
  BLK_SYNC();
  { PB_T0();
  store_ctu(S, J);
  store_ctu_pb(S, J);
  deblock_zeroes_unused_vectors(S, J);
  PB_T1(J.W, 12); }
  if (CTU_WAVE == 0) {
    LANE0 S->vsel[CTU_WAVE] = 3;
    CTU_SYNC();
    { PB_T0();
    coder_pass_pb(S, J);
    PB_T1(J.W, 13); }
    LANE0 S->vsel[CTU_WAVE] = 0;
  }
  BLK_SYNC();
  BLK_FOR(i, NMODELS) J.models_out[2 * NMODELS + i] = S->coder[i];
  BLK_FOR(i, NMX - NMODELS) J.pbm_out[2 * (NMX - NMODELS) + i] = S->coder[NMODELS + i];
  BLK_FOR(i, 41) J.pb->hmvp_rows[(size_t)(J.y >> 6) * 41 + i] = S->pb.hmvp_coder[i];
  BLK_SYNC();
  PB_T1(J.W, 14);
}

}  // namespace ctu
