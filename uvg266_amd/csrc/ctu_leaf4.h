// The 4x4 leaf of the closed-loop intra search (ctu_core.h) on ONE wave, register-resident.
//
// A CTU's time is the chain of its 256 4x4 CUs (each predicts from the previous one's reconstruction), and a lone wave pays a full
// LDS round trip (~100 cycles) for every dependent memory access and ~5 cycles for every instruction it issues.  The general CU
// evaluation (eval_cu) walks a 4x4 CU the way it walks a 32x32 one: block-sized arrays in LDS, a wave fence between every two
// steps, four passes of prediction + SATD with three selections between them.  This file is the same arithmetic with the data where
// a 4x4 block fits -- in the lanes' registers:
//   * the 8x8 area's source samples are fetched from the picture ONCE per area (not three times per CU) and parked in LDS;
//   * the rough search evaluates every angular mode in ONE pass, a lane per MODE (a lane predicts the whole 4x4 block and takes its
//     Hadamard in registers); planar / DC run beside it on eight lanes.  The reference's three refinement rounds
//     (search_intra.c:1071-1215) become selections over the finished cost table: a mode's cost does not depend on the round;
//   * transforms: a lane per coefficient, the 4-point butterflies' operands fetched from the other lanes (ds_bpermute), no arrays;
//   * uvg_rdoq (rdo.c:1449-1870) for the one coefficient group of a 4x4 block: a lane per position in RASTER order -- the context
//     template's neighbours are the lanes +1, +2, +4, +5, +8 (DPP row shifts), the regular-bin budget a position sees is a
//     population count over the positions later in scan order, and the positions are decided together by the fixed-point iteration
//     of rdoq_wave, which here also covers the blocks where the budget runs out (a position's "regular" flag is a function of the
//     levels later in scan order, like its contexts: same DAG);  the sums the reference forms in scan order are formed from the
//     owners' registers with v_readlane in that order.
// Device only (the host emulation keeps the general path: the two agree only if this reformulation is right; the GPU tests hold the
// device to the reference-run goldens).
// (included by ctu_core.h inside namespace ctu, behind the helpers it builds on)
#pragma once

// the up-right diagonal scan of a 4x4 block (tables.c g_scan_order): raster position of scan index k, and the inverse
#define LF_SCAN(k) ((int)((0xFBE7AD369C258140ull >> (4 * (k))) & 15))
#define LF_INV(r) ((int)((0xFDA6EB73C8419520ull >> (4 * (r))) & 15))

CTU_DEV int lf_shfl(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
// the value of lane + OFF of the same row of 16 lanes (0 beyond the row)
#if defined(LEAF_NO_DPP)
template <int OFF> CTU_DEV int lf_nb(int v) { const int l = (int)(threadIdx.x & 63); const int o = lf_shfl(v, l + OFF); return ((l & 15) + OFF) < 16 ? o : 0; }
#else
template <int OFF> CTU_DEV int lf_nb(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + OFF, 0xf, 0xf, true); }      // row_shl:OFF
#endif
// sum over the 16 lanes of a row, valid in every lane of the row
CTU_DEV int lf_row_sum(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);      // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);      // row_mirror
  return v;
}

// ---- per-CTU tables (LDS) ------------------------------------------------------------------------------------------------------
template <typename PX> CTU_DEV void leaf_tables(lds<PX> *S, const params &P)
{
  BLK_FOR(t, 2) {           // uvg_rdoq's per-block constants (rdo.c:1500-1530) for a 4x4 block of luma / chroma
    const int qps = (t ? P.qp_c : P.qp) + 6 * ((int)px_info<PX>::depth - 8);
    const int transform_shift = 15 - (int)px_info<PX>::depth - 2;
    const int q = kQuantScales[qps % 6];
    double scale = 32768;
    scale = transform_shift >= 0 ? scale / kPow2[2 * transform_shift] : scale * kPow2[-2 * transform_shift];
    S->lf_escale[t] = scale / q / q;
    S->lf_qbits[t] = 14 + qps / 6 + transform_shift;
    S->lf_q[t] = q;
  }
  BLK_FOR(i, 32) S->lf_cubic[i] = (uint32_t)(uint8_t)kCubic[i][0] | (uint32_t)(uint8_t)kCubic[i][1] << 8 | (uint32_t)(uint8_t)kCubic[i][2] << 16 | (uint32_t)(uint8_t)kCubic[i][3] << 24;
  BLK_FOR(i, 17) S->lf_disp[i] = (uint32_t)kSampleDisp[i] | (uint32_t)kInvDisp[i] << 8;
  BLK_FOR(r, 16) {
    const int sp = LF_INV(r);
    uint32_t later = 0;
    for (int q = 0; q < 16; ++q) if (LF_INV(q) > sp) later |= 1u << q;
    S->lf_rq[r] = later | (uint32_t)sp << 16 | (uint32_t)(sp < 15 ? LF_SCAN(sp + 1) : 0) << 20;
  }
  if (BLK_TID == 0) S->lf_tag = -1;
}

// the 8x8 area's source samples, once per area: [0, 64) luma (row * 8 + column), [64, 80) Cb, [80, 96) Cr
template <typename PX> CTU_DEV void leaf_load_area(lds<PX> *S, const job<PX> &J, int lx, int ly)
{
  const int ax = lx & ~7, ay = ly & ~7, tag = ay << 8 | ax;
  if (__builtin_amdgcn_readfirstlane(S->lf_tag) == tag) return;
  const int l = CTU_TID;
  CTU_LDS PX *const dst = LDSP(PX, S->lf_src);
  {
    const CTU_GLB PX *p = (const CTU_GLB PX *)J.src_y + (size_t)(J.y + ay + (l >> 3)) * J.src_stride + J.x + ax + (l & 7);
    const PX v = *p;
    PX c = 0;
    if (l < 32) {
      const PX *pl = l < 16 ? J.src_u : J.src_v;
      const int e = l & 15;
      c = ((const CTU_GLB PX *)pl)[(size_t)(((J.y + ay) >> 1) + (e >> 2)) * J.src_stride_c + ((J.x + ax) >> 1) + (e & 3)];
    }
    dst[l] = v;
    if (l < 32) dst[64 + l] = c;
  }
  LANE0 S->lf_tag = tag;
  CTU_SYNC();
}
// lane e (0..15; the lanes above repeat) -> source sample e of the 4x4 block of `color` at CTU-local luma (lx, ly)
template <typename PX> CTU_DEV int leaf_src(lds<PX> *S, int color, int lx, int ly)
{
  const int e = CTU_TID & 15;
  CTU_LDS const PX *const src = LDSP(const PX, S->lf_src);
  if (color == 0) return (int)src[((ly & 4) + (e >> 2)) * 8 + (lx & 4) + (e & 3)];
  return (int)src[64 + (color - 1) * 16 + e];
}

// ---- reference rows ------------------------------------------------------------------------------------------------------------
// uvg_intra_build_reference (intra.c:756-1341) for a 4x4 block of `color`: the luma block of the 4x4 CU at (lx, ly) (n = 4), or a chroma
// block of the 8x8 area at (lx, ly) (n = 8).  Entries 0..16 of V->top / V->left (a 4x4 block reads 0..10); no smoothed rows (never used
// for 4x4 blocks, intra.c:715-725).  Every lane derives the availability itself: no hand-over from lane 0.
template <typename PX> CTU_DEV void leaf_refs(lds<PX> *S, const params &P, wctx *V, int color, int x, int y, int lx, int ly, int n)
{
  const int c = color != 0, w = 4;
  const int px_x = lx >> c, px_y = ly >> c, pit = pitch_of(color);
  CTU_LDS const PX *const D = LDSP(const PX, plane(S, color)) + (px_y + 1) * pit + px_x + 1;
#define LF_TYPE(xx, yy) __builtin_amdgcn_readfirstlane((int)cu_at(S, (xx), (yy))->type)
  int al = 0, at = 0;
  if (x > 0) {
    int units;
    if (lx == 0) units = (LCU - ly) / 4;
    else {
      int amount = n;
      if (ly + amount < LCU && LF_TYPE(lx - 4, ly + amount) != CU_NOTSET) {
        amount += 4;
        if (n == 8 && ly + amount < LCU && LF_TYPE(lx - 4, ly + amount) != CU_NOTSET) amount += 4;
      }
      units = amount / 4;
    }
    al = units * (c ? 2 : 4);
    if (al > 2 * w) al = 2 * w;
    if (al > ((P.pic_h - y) >> c)) al = (P.pic_h - y) >> c;
  }
  if (y > 0) {
    int units;
    if (ly == 0) units = n / 2;
    else {
      int amount = n;
      if (lx + amount < LCU && LF_TYPE(lx + amount, ly - 4) != CU_NOTSET) {
        amount += 4;
        if (n == 8 && lx + amount < LCU && LF_TYPE(lx + amount, ly - 4) != CU_NOTSET) amount += 4;
      }
      units = amount / 4;
    }
    at = units * (c ? 2 : 4);
    if (at > 2 * w) at = 2 * w;
    if (at > ((P.pic_w - x) >> c)) at = (P.pic_w - x) >> c;
    if (x > 0 && P.wpp && px_y == 0 && at > (LCU >> c) - px_x) at = (LCU >> c) - px_x;
  }
#undef LF_TYPE
  const int dc = 1 << (px_info<PX>::depth - 1);
  const int l = CTU_TID, i = l & 15;
  int v;
  if (l < 16) {               // left[1 + i]
    if (x > 0) v = D[(i < al ? i : al - 1) * pit - 1];
    else v = y > 0 ? (int)D[-pit] : dc;
  } else if (l < 32) {        // top[1 + i]
    if (y > 0) v = D[-pit + (i < at ? i : at - 1)];
    else v = x > 0 ? (int)D[-1] : dc;
  } else {                    // the corner ("copy reference clockwise": left[1] when the corner itself is missing)
    if (x > 0 && y > 0) v = D[-pit - 1];
    else v = x > 0 ? (int)D[-1] : (y > 0 ? (int)D[-pit] : dc);
  }
  CTU_LDS uint16_t *const r_top = LDSP(uint16_t, V->top), *const r_left = LDSP(uint16_t, V->left);
  if (l < 16) r_left[1 + i] = (uint16_t)v;
  else if (l < 32) r_top[1 + i] = (uint16_t)v;
  else if (l == 32) { r_left[0] = (uint16_t)v; r_top[0] = (uint16_t)v; }
  CTU_SYNC();
}

// ---- prediction ----------------------------------------------------------------------------------------------------------------
// what a lane needs to know about an angular mode of a 4x4 block (make_mode_info for w = h = 4: never a smoothed reference, always
// the cubic filter for luma -- the distance threshold of 4x4 blocks is 24 --, PDPC for the 6 modes at the ends and the 2 pure ones)
struct lf_mode {
  CTU_LDS const uint16_t *mainr, *side;
  int sd, inv;
  bool vertical, pd_ang, pd_sd0;
};
template <typename PX> CTU_DEV lf_mode leaf_mode(lds<PX> *S, wctx *V, int mode)
{
  lf_mode M;
  M.vertical = mode >= 34;
  const int md = M.vertical ? mode - 50 : 18 - mode, amd = md < 0 ? -md : md;
  const uint32_t t = LDSP(const uint32_t, S->lf_disp)[amd > 16 ? 16 : amd];
  M.sd = md < 0 ? -(int)(t & 0xff) : (int)(t & 0xff);
  M.inv = (int)(t >> 8);
  M.pd_ang = md >= 14;
  M.pd_sd0 = md == 0;
  CTU_LDS const uint16_t *const top = LDSP(const uint16_t, V->top), *const left = LDSP(const uint16_t, V->left);
  M.mainr = M.vertical ? top : left;
  M.side = M.vertical ? left : top;
  return M;
}
// row yd of the WORK domain (the block for vertical modes, its transpose for horizontal ones), 4 samples
// (intra-generic.c:118-246 for a 4x4 block; chroma interpolates linearly)
template <typename PX, bool CHROMA> CTU_DEV void leaf_ang_row(lds<PX> *S, const lf_mode &M, int yd, int (&out)[4])
{
  const int maxv = (int)px_info<PX>::maxv;
  const int delta = M.sd * (yd + 1), di = delta >> 5, df = delta & 31;
  int p[7];
#pragma unroll
  for (int k = 0; k < (CHROMA ? 6 : 7); ++k) {
    const int idx = di + k;
    int s = (-idx * M.inv + 256) >> 9;
    s = s < 4 ? s : 4;
    CTU_LDS const uint16_t *const q = idx >= 0 ? M.mainr + idx : M.side + s;
    p[k] = *q;
  }
  if (!CHROMA) {
    const uint32_t f = LDSP(const uint32_t, S->lf_cubic)[df];
    const int f0 = (int)(int8_t)(f & 0xff), f1 = (int)(int8_t)((f >> 8) & 0xff), f2 = (int)(int8_t)((f >> 16) & 0xff), f3 = (int)(int8_t)(f >> 24);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = clampi((f0 * p[i] + f1 * p[i + 1] + f2 * p[i + 2] + f3 * p[i + 3] + 32) >> 6, 0, maxv);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = p[i + 1] + ((df * (p[i + 2] - p[i + 1]) + 16) >> 5);
  }
  if (M.pd_ang || M.pd_sd0) {
    const int tl = M.mainr[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int wl = 32 >> (2 * i);
      const int li = M.pd_sd0 ? 1 + yd : yd + ((256 + (i + 1) * M.inv) >> 9) + 1;
      const int l = M.side[li];
      const int base = M.pd_sd0 ? tl : out[i];
      int v = out[i] + ((wl * (l - base) + 32) >> 6);
      if (M.pd_sd0) v = clampi(v, 0, maxv);
      out[i] = v;
    }
  }
}

// the prediction of sample e = lane & 15 of the 4x4 block for the (wave-uniform) mode
template <typename PX> CTU_DEV int leaf_predict(lds<PX> *S, wctx *V, int mode, int color)
{
  const int e = CTU_TID & 15, x = e & 3, y = e >> 2;
  int out[4];
  if (mode < 2) {
    const ref_rows_t<CTU_LDS const uint16_t *> R = {LDSP(const uint16_t, V->top), LDSP(const uint16_t, V->left), LDSP(const uint16_t, V->top), LDSP(const uint16_t, V->left)};
    const mode_info M = make_mode_info(mode, 4, 4, color != 0);        // (planar / DC: no table behind it; never the smoothed rows for 4x4)
    const int dc = mode == 1 ? dc_value(R.top, R.left, 4, 4) : 0;
    predict_row<4>(M, R, dc, color != 0, 4, 4, y, 0, (int)px_info<PX>::maxv, out);
    return x == 0 ? out[0] : x == 1 ? out[1] : x == 2 ? out[2] : out[3];
  }
  const lf_mode M = leaf_mode(S, V, mode);
  const int xd = M.vertical ? x : y, yd = M.vertical ? y : x;
  if (color) leaf_ang_row<PX, true>(S, M, yd, out); else leaf_ang_row<PX, false>(S, M, yd, out);
  return xd == 0 ? out[0] : xd == 1 ? out[1] : xd == 2 ? out[2] : out[3];
}

// ---- most probable modes, mode bits ----------------------------------------------------------------------------------------------
// uvg_intra_get_dir_luma_predictor (intra.c:88-188, MIP off) from the two neighbours' modes (0 where there is no intra neighbour)
CTU_DEV void lf_mpm(int left_dir, int above_dir, int (&p)[6])
{
  const int offset = 61, mod = 64;
  p[0] = 0; p[1] = 1; p[2] = 50; p[3] = 18; p[4] = 46; p[5] = 54;
  if (left_dir == above_dir) {
    if (left_dir > 1) {
      p[1] = left_dir;
      p[2] = ((left_dir + offset) % mod) + 2; p[3] = ((left_dir - 1) % mod) + 2;
      p[4] = ((left_dir + offset - 1) % mod) + 2; p[5] = (left_dir % mod) + 2;
    }
  } else if (left_dir > 1 && above_dir > 1) {
    p[1] = left_dir; p[2] = above_dir;
    const int mx = left_dir > above_dir ? left_dir : above_dir, mn = left_dir > above_dir ? above_dir : left_dir;
    const int diff = mx - mn;
    if (diff == 1) {
      p[3] = ((mn + offset) % mod) + 2; p[4] = ((mx - 1) % mod) + 2; p[5] = ((mn + offset - 1) % mod) + 2;
    } else if (diff >= 62) {
      p[3] = ((mn - 1) % mod) + 2; p[4] = ((mx + offset) % mod) + 2; p[5] = (mn % mod) + 2;
    } else if (diff == 2) {
      p[3] = ((mn - 1) % mod) + 2; p[4] = ((mn + offset) % mod) + 2; p[5] = ((mx - 1) % mod) + 2;
    } else {
      p[3] = ((mn + offset) % mod) + 2; p[4] = ((mn - 1) % mod) + 2; p[5] = ((mx + offset) % mod) + 2;
    }
  } else if (left_dir + above_dir >= 2) {
    p[1] = left_dir < above_dir ? above_dir : left_dir;
    p[2] = ((p[1] + offset) % mod) + 2; p[3] = ((p[1] - 1) % mod) + 2;
    p[4] = ((p[1] + offset - 1) % mod) + 2; p[5] = (p[1] % mod) + 2;
  }
}
// uvg_encode_intra_luma_coding_unit (encode_coding_tree.c:992-1238) in count mode with the list already derived; lane 0.
// (every term is a multiple of 2^-15: the order of the additions is immaterial)
CTU_DEV void lf_luma_mode_bits(uint32_t *m_, const int (&p)[6], int mode, double &bits_out)
{
  CTU_LDS uint32_t *const m = LDSP(uint32_t, m_);
  int mpm = -1;
#pragma unroll
  for (int i = 5; i >= 0; --i) if (p[i] == mode) mpm = i;
  double bits = 0;
  m_code(m, 1, M_MPM, mpm != -1, bits);
  if (mpm != -1) {
    m_code(m, 1, M_PLANAR + 1, mpm > 0, bits);
    bits += (mpm > 0) + (mpm > 1) + (mpm > 2) + (mpm > 3);
  } else {
    int tmp = mode;
#pragma unroll
    for (int i = 0; i < 6; ++i) tmp -= p[i] < mode;
    bits += (tmp < 3) ? 5 : 6;                      // truncated binary code of 61 symbols (cabac.c:203-229)
  }
  bits_out += bits;
}

// ---- coefficient bit cost of a 4x4 block -------------------------------------------------------------------------------------------
// coeff_bits4 (ctu_core.h) with the block in RASTER order over the lanes 0..15 (`lev`: the signed level of position lane & 15; every
// row of 16 lanes may repeat it): the context template's neighbours are DPP row shifts, what is counted along the scan reads the
// owners' registers in scan order, and the entropy-table lookups are taken out of the adaptation chain (a model's state after a bin
// does not depend on the bin's cost): the sixteen steps of the sweep are pure register arithmetic, the sixteen lookups go out together.
// Same bins, same adaptation, same sum as uvg_encode_coeff_nxn in count mode (encode_coding_tree-generic.c:53-323).
// BITS = false: the models' adaptation only (the coder's pass: nobody reads the count) -- no entropy-table lookups, no Rice
// parameters, no bypass bits, no sums.
template <typename PX, bool BITS = true> CTU_DEV double coeff_bits4r(lds<PX> *S, CTU_LDS uint32_t *m, int update, int lev, int color)
{
  const int lane = CTU_TID, r = lane & 15, px = r & 3, py = r >> 2, t = color ? 1 : 0;
  const int a = iabs_(lev);
  const unsigned nz = (unsigned)__ballot(a != 0) & 0xffffu;
  if (nz == 0) return 0.0;
  const uint32_t tab = LDSP(const uint32_t, S->lf_rq)[r];
  const unsigned later = tab & 0xffffu;
  const int sp = (int)((tab >> 16) & 15);
  const bool is_last = a != 0 && (nz & later) == 0;
  const int last_lane = __builtin_ctzll(__ballot(is_last));
  const int last = __builtin_amdgcn_readlane(sp, last_lane);           // scan index of the last significant position
  // this lane's models: the sweep's (one per lane, 12 + 3 * 16 luma / 8 + 3 * 11 chroma) and, lanes 0..5, a last-position prefix model
  const int nk0 = t ? 8 : 12, nks = t ? 11 : 16;
  int role = -1, k = 0;
  if (lane < nk0) { role = 0; k = lane; }
  else if (lane < nk0 + 3 * nks) { role = 1 + (lane - nk0) / nks; k = (lane - nk0) % nks; }
  if (!t && role > 0 && k > 0) k += 5;          // 4x4 luma: set offsets 0, 6..20
  const int model = role < 0 ? 0 : (role == 0 ? M_SIG + 12 * t : role == 1 ? M_GT1 + 21 * t : role == 2 ? M_PAR + 21 * t : M_GT2 + 21 * t) + k;
  const int paxis = lane >= 3, pq = lane - 3 * paxis;                // prefix: lanes 0..2 x, 3..5 y; a 4x4 block: offset 0, shift 0, three models per axis
  const int pmodel = (paxis ? M_LASTY : M_LASTX) + 20 * t + (lane < 6 ? pq : 0);
  uint32_t st = m[model];
  const uint32_t pst = m[pmodel];
  CTU_LDS const uint8_t *const rate = LDSP(const uint8_t, kRate);
  const int rw = rate[model], prw = rate[pmodel];
  // ---- per position: contexts, Rice parameters, whether its sig flag is coded, the regular bins it spends ----
  int ctx_sig, ofs = 0, r4 = 0, r0 = 0;
  {
    const int a1 = lf_nb<1>(a), a2 = lf_nb<2>(a), a5 = lf_nb<5>(a), a4 = lf_nb<4>(a), a8 = lf_nb<8>(a);
    int num_pos = 0, sum_abs = 0, sum = 0;
#define LF_UPD(v) { const int q = (v); sum_abs += (4 + (q & 1)) < q ? (4 + (q & 1)) : q; num_pos += q ? 1 : 0; sum += q; }
    if (px < 3) { LF_UPD(a1); if (px < 2) LF_UPD(a2); if (py < 3) LF_UPD(a5); }
    if (py < 3) { LF_UPD(a4); if (py < 2) LF_UPD(a8); }
#undef LF_UPD
    const int diag = px + py, tsum = sum_abs - num_pos;
    ctx_sig = (((sum_abs + 1) >> 1) < 3 ? ((sum_abs + 1) >> 1) : 3) + (diag < 2 ? 4 : 0);
    if (color == 0) ctx_sig += diag < 5 ? 4 : 0;
    if (t && ctx_sig > 7) ctx_sig = 7;
    if (sp != last) ofs = ((tsum < 4 ? tsum : 4) + 1) + (!diag ? (color == 0 ? 15 : 5) : color == 0 ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
    if (BITS) {
      int v4 = sum - 20, v0 = sum;
      v4 = v4 < 31 ? v4 : 31; v0 = v0 < 31 ? v0 : 31;
      r4 = go_rice_par((unsigned)(v4 > 0 ? v4 : 0)); r0 = go_rice_par((unsigned)(v0 > 0 ? v0 : 0));
    }
  }
  const bool live = sp <= last;
  const int sig_coded = live && sp != last;
  const int spend = live ? sig_coded + (a ? 1 + (a > 1 ? 2 : 0) : 0) : 0;
  const uint32_t rec = (uint32_t)(a > 0xffff ? 0xffff : a) | (uint32_t)ctx_sig << 16 | (uint32_t)ofs << 20 | (uint32_t)sig_coded << 29;
  // where the regular-bin budget (28 for 16 coefficients) runs out: scan positions <= sw are bypass-coded
  const int tot = __popc((unsigned)__ballot(spend & 1) & 0xffffu) + 2 * __popc((unsigned)__ballot(spend & 2) & 0xffffu) + 4 * __popc((unsigned)__ballot(spend & 4) & 0xffffu);
  int sw = -1;
  if (28 - tot < 4) {
    int rb = 28;
    for (int j = last; j >= 0; --j) {
      if (rb < 4) { sw = j; break; }
      rb -= __builtin_amdgcn_readlane(spend, LF_SCAN(j));
    }
  }
  // ---- the sweep: every model along the positions in coding order; states only ----
  const int r0w = rw >> 4, r1w = rw & 15;
  const uint32_t add0 = (0x7fffu >> r0w) & 0x7fe0u, add1 = (0x7fffu >> r1w) & 0x7ffeu;
  const uint32_t fsh = role == 0 ? 16 : 20, fmask = role == 0 ? 15u : 31u, rsel = role < 0 ? 31u : (uint32_t)role;
  uint32_t idx[16];
  uint32_t hits = 0;
#pragma unroll
  for (int j = 15; j >= 0; --j) {
    idx[j] = 0;
    if (j > last || j <= sw) continue;              // (wave-uniform: the steps beyond the last position / behind the budget's end cost nothing)
    const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)rec, LF_SCAN(j));
    const uint32_t aj = rj & 0xffffu;
    // per role (sig, gt1, parity, gt2): does the position code a bin with one of the role's models, and which
    const uint32_t gates = ((rj >> 29) & 1u) | (aj != 0 ? 2u : 0u) | (aj > 1 ? 12u : 0u);
    if (gates == 0) continue;
    const uint32_t bins = (aj != 0 ? 1u : 0u) | (aj > 1 ? 2u : 0u) | ((aj & 1u) << 2) | (aj >= 4 ? 8u : 0u);
    const bool hit = ((gates >> rsel) & 1u) && ((rj >> fsh) & fmask) == (uint32_t)k;
    const uint32_t bin = (bins >> rsel) & 1u;
    uint32_t s0 = st & 0xffffu, s1 = st >> 16;
    if (BITS) idx[j] = (((s0 + s1) >> 8) << 1) ^ bin;
    s0 -= (s0 >> r0w) & 0x7fe0u;
    s1 -= (s1 >> r1w) & 0x7ffeu;
    s0 += bin ? add0 : 0u;
    s1 += bin ? add1 : 0u;
    st = hit ? ((s0 & 0xffffu) | (s1 << 16)) : st;
    hits |= hit ? 1u << j : 0u;
  }
  if (role >= 0 && update) m[model] = st;
  // the last-position prefix (uvg_encode_last_significant_xy, :415-470): model q of an axis sees one bin at most -- 1 below the
  // position's coordinate, 0 at it (none at 3)
  const int last_r = LF_SCAN(last), pos = paxis ? last_r >> 2 : last_r & 3;
  const bool phit = lane < 6 && pq <= pos && pq < 3;
  const uint32_t pbin = pq < pos ? 1u : 0u;
  uint32_t pidx;
  {
    uint32_t s0 = pst & 0xffffu, s1 = pst >> 16;
    pidx = (((s0 + s1) >> 8) << 1) ^ pbin;
    const int p0w = prw >> 4, p1w = prw & 15;
    s0 -= (s0 >> p0w) & 0x7fe0u;
    s1 -= (s1 >> p1w) & 0x7ffeu;
    if (pbin) { s0 += (0x7fffu >> p0w) & 0x7fe0u; s1 += (0x7fffu >> p1w) & 0x7ffeu; }
    if (phit && update) m[pmodel] = (s0 & 0xffffu) | (s1 << 16);
  }
  if (!BITS) { WSYNC(); return 0.0; }
  // ---- the bins' costs, all lookups in flight together ----
  CTU_LDS const uint32_t *const ebits = LDSP(const uint32_t, tab_ebits());
  uint32_t acc = 0;
  {
    uint32_t c[16];
    const uint32_t pc = ebits[pidx];
#pragma unroll
    for (int j = 0; j < 16; ++j) { c[j] = 0; if (j <= last && j > sw) c[j] = ebits[idx[j]]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += (hits >> j) & 1u ? c[j] : 0u;
    acc += phit ? pc : 0u;
  }
  // ---- bypass-coded parts: remainders, bypass positions, signs ----
  int ibits = 0;
  if (lane < 16 && live) {
    if (sp > sw) { if (a >= 4) ibits += coeff_remain_bits(((unsigned)a - 4) >> 1, (uint32_t)r4, 5); }
    else {
      const unsigned pos0 = 1u << r0;
      ibits += coeff_remain_bits(a == 0 ? pos0 : ((unsigned)a <= pos0 ? (unsigned)a - 1 : (unsigned)a), (uint32_t)r0, 5);
    }
    ibits += a != 0;
  }
  const int racc = lf_row_sum((int)acc);
  const unsigned q15 = (unsigned)__builtin_amdgcn_readlane(racc, 0) + (unsigned)__builtin_amdgcn_readlane(racc, 16) + (unsigned)__builtin_amdgcn_readlane(racc, 32) +
                       (unsigned)__builtin_amdgcn_readlane(racc, 48);
  const int ib = __builtin_amdgcn_readlane(lf_row_sum(ibits), 0);
  WSYNC();
  return (double)q15 / 32768.0 + (double)ib;
}

// ---- rough search --------------------------------------------------------------------------------------------------------------
// count_bits (search_intra.c:949-984) with the predictor list in six scalars
CTU_DEV double lf_count_bits(int p0, int p1, int p2, int p3, int p4, int p5, double planar, double not_planar, double mpm_bit, double not_mpm_bit, int mode)
{
  int i = 6, smaller = 0;
  const int p[6] = {p0, p1, p2, p3, p4, p5};
  bool found = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (!found) {
      if (p[k] == mode) { found = true; i = k; }
      else if (mode > p[k]) smaller += 1;
    }
  }
  if (i == 0) return planar + mpm_bit;
  if (i < 6) return not_planar + mpm_bit + (i < 4 ? i : 4);
  return not_mpm_bit + 5 + (mode - smaller > 2);
}

// min(SATD, 2 SAD) + mode bits * sqrt(lambda) of the 4x4 block against this lane's 16 differences (work domain; both measures are
// transpose-invariant); get_cost_dual, search_intra.c:133-192
template <typename PX> CTU_DEV double lf_cost(int (&d)[16], double bits, double lambda_sqrt)
{
  unsigned sad = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) sad += (unsigned)iabs_(d[i]);
  const unsigned satd = satd4_tile(d);                       // (the 4x4 function does not shift by the bit depth, picture-generic.c:170)
  sad >>= (px_info<PX>::depth - 8);
  double c = (double)(satd < sad * 2 ? satd : sad * 2);
  c += bits * lambda_sqrt;
  return c;
}

// search_intra_rough (search_intra.c:986-1229) for the 4x4 luma CU at (lx, ly) whose reference rows are in V; srcv: lane e < 16 holds
// source sample e.  Returns the mode (wave-uniform).
template <typename PX> CTU_INLINE1 CTU_DEV int leaf_rough(lds<PX> *S, const job<PX> &J, wctx *V, int x, int y, int lx, int ly, int srcv, int (&mpm)[6])
{
  const params &P = J.P;
  const int lane = CTU_TID;
  LF_T0();
  int s[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_readlane(srcv, e);
  // the most probable modes and the four flag costs (every lane; wave-uniform)
  int p0, p1, p2, p3, p4, p5;
  {
    // uvg_intra_get_dir_luma_predictor (intra.c:88-188) on scalars: the two neighbours' modes are wave-uniform
    const cu4 *l, *a;
    mpm_neighbours(S, x, y, lx, ly, 4, &l, &a);
    int left_dir = 0, above_dir = 0;
    if (l && l->type == CU_INTRA) left_dir = l->mode;
    if (a && a->type == CU_INTRA && y % LCU != 0) above_dir = a->mode;
    lf_mpm(__builtin_amdgcn_readfirstlane(left_dir), __builtin_amdgcn_readfirstlane(above_dir), mpm);
    p0 = mpm[0]; p1 = mpm[1]; p2 = mpm[2]; p3 = mpm[3]; p4 = mpm[4]; p5 = mpm[5];
  }
  CTU_LDS const uint32_t *const mdl = LDSP(const uint32_t, V->cur);
  const double mpm_bit = m_fbits(mdl, M_MPM, 1), not_mpm_bit = m_fbits(mdl, M_MPM, 0);
  const double planar = m_fbits(mdl, M_PLANAR + 1, 0), not_planar = m_fbits(mdl, M_PLANAR + 1, 1);
  // ---- planar and DC: eight lanes, a lane per (mode, row) ----
  double c_planar, c_dc;
  {
    const int mi = (lane >> 2) & 1, r = lane & 3;
    const ref_rows_t<CTU_LDS const uint16_t *> R = {LDSP(const uint16_t, V->top), LDSP(const uint16_t, V->left), LDSP(const uint16_t, V->top), LDSP(const uint16_t, V->left)};
    const mode_info M = make_mode_info(mi, 4, 4, 0);
    const int dcv = dc_value(R.top, R.left, 4, 4);
    int out[4], d[4], sad = 0;
    predict_row<4>(M, R, dcv, 0, 4, 4, r, 0, (int)px_info<PX>::maxv, out);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sv = r == 0 ? s[i] : r == 1 ? s[4 + i] : r == 2 ? s[8 + i] : s[12 + i];
      d[i] = sv - out[i];
      sad += iabs_(d[i]);
    }
    int satd = satd4_cost(d, r);
    sad = dpp_group_sum<4>(sad);
    sad >>= (px_info<PX>::depth - 8);
    const double c = (double)(satd < sad * 2 ? satd : sad * 2) + lf_count_bits(p0, p1, p2, p3, p4, p5, planar, not_planar, mpm_bit, not_mpm_bit, mi) * P.lambda_sqrt;
    c_planar = rl64(c, 0);
    c_dc = rl64(c, 4);
  }
  LF_T(2);
  // ---- pass A: lanes 2..63 = modes 4..65 (lanes 0, 1 idle along) ----
  double cA;
  {
    const int mode = lane < 2 ? 18 : lane + 2;
    const lf_mode M = leaf_mode(S, V, mode);
    int d[16];
#pragma unroll
    for (int yd = 0; yd < 4; ++yd) {
      int out[4];
      leaf_ang_row<PX, false>(S, M, yd, out);
#pragma unroll
      for (int i = 0; i < 4; ++i) d[yd * 4 + i] = (M.vertical ? s[yd * 4 + i] : s[i * 4 + yd]) - out[i];
    }
    cA = lf_cost<PX>(d, lf_count_bits(p0, p1, p2, p3, p4, p5, planar, not_planar, mpm_bit, not_mpm_bit, mode), P.lambda_sqrt);
  }
  cA = lane == 0 ? c_planar : (lane == 1 ? c_dc : cA);
  LF_T(3);
  // cost of a (wave-uniform) mode: pass A's lanes, or pass B's (lanes 0..2 = modes 2, 3, 66)
  double cB = 0;
#define LF_LANE_A(m) ((m) < 2 ? (m) : (m) - 2)
#define LF_IN_B(m) ((m) == 2 || (m) == 3 || (m) == 66)
#define LF_LANE_B(m) ((m) == 66 ? 2 : (m) - 2)
#define LF_COST(m) (LF_IN_B(m) ? rl64(cB, LF_LANE_B(m)) : rl64(cA, LF_LANE_A(m)))
  // ---- round 0: planar, DC, every 2^levels-th angular mode (search_intra.c:1071-1143) ----
  // The reference's three-entry list under strict "<" insertion = the three smallest under (cost, insertion sequence); DC is inserted
  // ahead of planar when the two tie (:1089-1106).  Candidates sit one per lane: (mode, cost, sequence); a lane counts who is ahead of it.
  const int levels = __builtin_amdgcn_readfirstlane(P.rough_levels);
  int offset = 1 << levels;
  const int first_m = 2 + offset / 2;                                   // listed angular modes: first_m + k * offset <= 66
  int ncand = 2 + (66 - first_m) / offset + 1;
  int cm = lane < 2 ? lane : first_m + (lane - 2) * offset, cseq = lane == 0 ? 1 : (lane == 1 ? 0 : 3 + lane);
  if (lane >= ncand) cm = 0;
  double cc;
  {
    const int src = LF_LANE_A(cm);                                      // (round 0 never lists a mode of pass B: first_m >= 4)
    cc = __hiloint2double(lf_shfl(__double2hiint(cA), src), lf_shfl(__double2loint(cA), src));
  }
  // modes costed so far: a flag in the mode's own lane of pass A / pass B
  int seenA = (lane < 2 || (lane + 2 >= first_m && ((lane + 2 - first_m) & (offset - 1)) == 0)) ? 1 : 0, seenB = 0;
  int b0, b1, b2;
  double k0, k1, k2;
  bool have_B = false;
  bool differs = false;
  for (int round = 0;; ++round) {
    // rank of every candidate
    int rank = 0;
    for (int j = 0; j < ncand; ++j) {
      const double o = rl64(cc, j);
      const int oseq = __builtin_amdgcn_readlane(cseq, j);
      rank += (o < cc || (o == cc && oseq < cseq)) ? 1 : 0;
    }
    const bool in = lane < ncand;
    if (round == 0) differs = __ballot(in && cc != rl64(cc, 0)) != 0;      // min_cost != max_cost (:1082-1143): only the first round moves them
    const int w0 = __builtin_ctzll(__ballot(in && rank == 0)), w1 = __builtin_ctzll(__ballot(in && rank == 1)), w2 = __builtin_ctzll(__ballot(in && rank == 2));
    b0 = __builtin_amdgcn_readlane(cm, w0); b1 = __builtin_amdgcn_readlane(cm, w1); b2 = __builtin_amdgcn_readlane(cm, w2);
    k0 = rl64(cc, w0); k1 = rl64(cc, w1); k2 = rl64(cc, w2);
    // next round's list (search_intra.c:1146-1215)
    const int off = offset >> 1;
    if (!(off > 0 && differs)) break;
    if (!have_B) {
      // (2, 3, 66 can only be listed from a survivor within 4 of them: 2 + off, 3 +- off, 66 - off with off <= 4)
#define LF_EDGE(b) (((b) >= 2 && (b) <= 7) || (b) >= 62)
      const bool edge = LF_EDGE(b0) || LF_EDGE(b1) || LF_EDGE(b2);
#undef LF_EDGE
      if (edge) {
        const int mode = lane == 0 ? 2 : (lane == 1 ? 3 : 66);
        const lf_mode M = leaf_mode(S, V, mode);
        int d[16];
#pragma unroll
        for (int yd = 0; yd < 4; ++yd) {
          int out[4];
          leaf_ang_row<PX, false>(S, M, yd, out);
#pragma unroll
          for (int i = 0; i < 4; ++i) d[yd * 4 + i] = (M.vertical ? s[yd * 4 + i] : s[i * 4 + yd]) - out[i];
        }
        cB = lf_cost<PX>(d, lf_count_bits(p0, p1, p2, p3, p4, p5, planar, not_planar, mpm_bit, not_mpm_bit, mode), P.lambda_sqrt);
        have_B = true;
      }
    }
    // survivors in lanes 0..2 (sequence 0..2), the new modes behind them (sequence 3 + index): written lane by lane from scalars
    int cclo = __double2loint(cc), cchi = __double2hiint(cc);
#define LF_PUT(ln, m_, c_) do { const double c__ = (c_); const bool me__ = lane == (ln); cm = me__ ? (m_) : cm; cclo = me__ ? __double2loint(c__) : cclo; \
      cchi = me__ ? __double2hiint(c__) : cchi; } while (0)
    LF_PUT(0, b0, k0); LF_PUT(1, b1, k1); LF_PUT(2, b2, k2);
    int n_new = 0;
#define LF_TRY(m_) do { const int m = (m_); if (m >= 2 && m <= 66) { const bool inb = LF_IN_B(m); const int ln = inb ? LF_LANE_B(m) : LF_LANE_A(m); \
      const int was = inb ? __builtin_amdgcn_readlane(seenB, ln) : __builtin_amdgcn_readlane(seenA, ln); \
      if (!was) { if (inb) { seenB = lane == ln ? 1 : seenB; LF_PUT(3 + n_new, m, rl64(cB, ln)); } \
                  else { seenA = lane == ln ? 1 : seenA; LF_PUT(3 + n_new, m, rl64(cA, ln)); } ++n_new; } } } while (0)
    if (b0 >= 3 && b0 <= 65) { LF_TRY(b0 - off); LF_TRY(b0 + off); }
    if (b1 >= 3 && b1 <= 65) { LF_TRY(b1 - off); LF_TRY(b1 + off); }
    if (b2 >= 3 && b2 <= 65) { LF_TRY(b2 - off); LF_TRY(b2 + off); }
#undef LF_TRY
#undef LF_PUT
    cc = __hiloint2double(cchi, cclo);
    ncand = 3 + n_new;
    cseq = lane;
    offset = off;
  }
#undef LF_LANE_A
#undef LF_IN_B
#undef LF_LANE_B
#undef LF_COST
  (void)k0; (void)k1; (void)k2; (void)b1; (void)b2;
  LF_T(4);
  return b0;
}

// ---- transforms ------------------------------------------------------------------------------------------------------------------
// dct_4x4 / idct_4x4 (dct-generic.c:396-446) one pass: lane e of a row of 16 holds element e of the block (raster)
CTU_DEV int lf_fwd_pass(int v, int shift)
{
  // dst[j * 4 + i] = (sum_k T[j][k] * src[i * 4 + k] + add) >> shift, truncated to int16
  const int l = CTU_TID, e = l & 15, j = e >> 2, i = e & 3, base = (l & 48) + i * 4;
  const int t0 = j == 0 ? 64 : (j == 1 ? 83 : (j == 2 ? 64 : 36)), t1 = j == 0 ? 64 : (j == 1 ? 36 : (j == 2 ? -64 : -83));
  const int t2 = j == 0 ? 64 : (j == 1 ? -36 : (j == 2 ? -64 : 83)), t3 = j == 0 ? 64 : (j == 1 ? -83 : (j == 2 ? 64 : -36));
  const int a0 = lf_shfl(v, base), a1 = lf_shfl(v, base + 1), a2 = lf_shfl(v, base + 2), a3 = lf_shfl(v, base + 3);
  const int acc = t0 * a0 + t1 * a1 + t2 * a2 + t3 * a3;
  const int add = shift > 0 ? 1 << (shift - 1) : 0;
  return (int)(int16_t)((acc + add) >> shift);
}
CTU_DEV int lf_inv_pass(int v, int shift)
{
  // dst[i * 4 + j] = clip16((sum_k src[k * 4 + i] * T[k][j] + add) >> shift)
  const int l = CTU_TID, e = l & 15, i = e >> 2, j = e & 3, base = (l & 48) + i;
  const int t0 = 64, t1 = j == 0 ? 83 : (j == 1 ? 36 : (j == 2 ? -36 : -83)), t2 = (j == 0 || j == 3) ? 64 : -64, t3 = j == 0 ? 36 : (j == 1 ? -83 : (j == 2 ? 83 : -36));
  const int a0 = lf_shfl(v, base), a1 = lf_shfl(v, base + 4), a2 = lf_shfl(v, base + 8), a3 = lf_shfl(v, base + 12);
  const int acc = a0 * t0 + a1 * t1 + a2 * t2 + a3 * t3;
  return clampi((acc + (1 << (shift - 1))) >> shift, -32768, 32767);
}

// ---- uvg_rdoq for a 4x4 block ----------------------------------------------------------------------------------------------------
// coef: lane r (raster; every row of 16 lanes repeats) holds coefficient r.  Returns the level of position r with its sign (0 beyond
// the chosen last position); *has_out = any level survived (wave-uniform).
template <typename PX> CTU_DEV int leaf_rdoq(lds<PX> *S, int coef, int color, int cbf_u, int qp_scaled, double lambda, int *has_out)
{
  const int bitdepth = (int)px_info<PX>::depth;
  const int lane = CTU_TID, r = lane & 15, px = r & 3, py = r >> 2;
  const uint32_t tab = LDSP(const uint32_t, S->lf_rq)[r];
  const unsigned later = tab & 0xffffu;                     // raster positions later in scan order
  const int sp = (int)((tab >> 16) & 15), next_r = (int)((tab >> 20) & 15);
  rdoq_env E;
  E.st = S->rdoq_state; E.t = color ? 1 : 0; E.lambda = lambda;
  E.q_bits = __builtin_amdgcn_readfirstlane(LDSP(const int32_t, S->lf_qbits)[E.t]);
  E.q = __builtin_amdgcn_readfirstlane(LDSP(const int32_t, S->lf_q)[E.t]);
  E.error_scale = LDSP(const double, S->lf_escale)[E.t];
  (void)bitdepth; (void)qp_scaled;
  const int cap_half = 1 << (E.q_bits - 1);
  const int32_t cap = 0x7fffffff - cap_half;
  const int ac = iabs_(coef);
  const int64_t prod = (int64_t)ac * E.q;
  const int32_t level_double = (int32_t)(prod < cap ? prod : cap);
  const int mal = (int)((uint32_t)(level_double + cap_half) >> E.q_bits);
  const double c0 = (double)level_double * (double)level_double * E.error_scale;
  const unsigned nz = (unsigned)__ballot(mal > 0) & 0xffffu;
  if (nz == 0) { *has_out = 0; return 0; }
  // the last candidate in scan order: the one with no candidate later than it
  const bool is_last = mal > 0 && (nz & later) == 0;
  const int last_lane = __builtin_ctzll(__ballot(is_last));
  const int last_sp = __builtin_amdgcn_readlane(sp, last_lane);
  const bool mine = sp <= last_sp;
  // the positions behind the last candidate only add their level-0 cost (rdo.c:1556-1583), in descending scan order
  double block_uncoded_cost = 0, base_cost = 0;
#pragma unroll
  for (int k = 15; k >= 1; --k)
    if (k > last_sp) { const double c = rl64(c0, LF_SCAN(k)); block_uncoded_cost += c; base_cost += c; }
  // the Rice parameter a regular-coded position inherits: from the INPUT coefficients around the position coded before it (rdo.c:1697)
  int go_rice_reg = 0;
  {
    // template_abs_sum(coef, 4, ...) at this lane's own position, then fetched from the position next in scan order
    int16_t sum = 0;
    const int a1 = lf_nb<1>(ac), a2 = lf_nb<2>(ac), a5 = lf_nb<5>(ac), a4 = lf_nb<4>(ac), a8 = lf_nb<8>(ac);
    if (px < 3) { sum = (int16_t)(sum + a1); if (px < 2) sum = (int16_t)(sum + a2); if (py < 3) sum = (int16_t)(sum + a5); }
    if (py < 3) { sum = (int16_t)(sum + a4); if (py < 2) sum = (int16_t)(sum + a8); }
    int v = sum - 20;
    v = v < 31 ? v : 31;
    const int t_in = go_rice_par((unsigned)(v > 0 ? v : 0));
    const int nb = lf_shfl(t_in, (lane & 48) + next_r);
    if (mine && sp != 15 && !is_last) go_rice_reg = nb;
  }
  // ---- the group's decisions: a fixed point over the DAG "later in scan order" ----
  int lev = mine ? mal : 0;
  double cc = 0, cs = 0;
  for (;;) {
    const int l1 = lf_nb<1>(lev), l2 = lf_nb<2>(lev), l5 = lf_nb<5>(lev), l4 = lf_nb<4>(lev), l8 = lf_nb<8>(lev);
    // regular bins left when this position is reached: 28 minus what the positions later in scan order (up to the last) spend
    const int spend = mine ? (lev < 2 ? lev : 3) + (is_last ? 0 : 1) : 0;
    const unsigned m0 = (unsigned)__ballot(spend & 1) & later, m1 = (unsigned)__ballot(spend & 2) & later, m2 = (unsigned)__ballot(spend & 4) & later;
    const int spent = __popc(m0) + 2 * __popc(m1) + 4 * __popc(m2);
    const bool regular = 28 - spent >= 4;
    int level = 0;
    if (mine) {
      int ctx_sig = 0, ctx_set = 0, tsum_lev = 0;
      {
        int num_pos = 0, sum_abs = 0;
#define LF_UPD(v) { const int a = (v); sum_abs += (4 + (a & 1)) < a ? (4 + (a & 1)) : a; num_pos += a ? 1 : 0; tsum_lev += a; }
        if (px < 3) { LF_UPD(l1); if (px < 2) LF_UPD(l2); if (py < 3) LF_UPD(l5); }
        if (py < 3) { LF_UPD(l4); if (py < 2) LF_UPD(l8); }
#undef LF_UPD
        if (!is_last) {
          const int diag = px + py;
          ctx_sig = (((sum_abs + 1) >> 1) < 3 ? ((sum_abs + 1) >> 1) : 3) + (diag < 2 ? 4 : 0);
          if (color == 0) ctx_sig += diag < 5 ? 4 : 0;
          const int tsum = sum_abs - num_pos;
          ctx_set = ((tsum < 4 ? tsum : 4) + 1) + (!diag ? ((color == 0) ? 15 : 5) : (color == 0) ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
        }
      }
      int go_rice = go_rice_reg;
      if (!regular) { const int v = tsum_lev < 31 ? tsum_lev : 31; go_rice = go_rice_par((unsigned)v); }      // template_abs_sum(dst, 0, ...)
      cs = 0;
      level = (int)coded_level(E, &cc, c0, &cs, level_double, (uint32_t)mal, ctx_sig, ctx_set, go_rice, regular ? 4u : 0u, is_last);
    }
    const bool changed = mine && level != lev;
    lev = level;
    if (__ballot(changed) == 0) break;
  }
  // ---- the sums in scan order (rdo.c:1689-1772; one group: no group decision) ----
#pragma unroll
  for (int k = 15; k >= 0; --k)
    if (k <= last_sp) { block_uncoded_cost += rl64(c0, LF_SCAN(k)); base_cost += rl64(cc, LF_SCAN(k)); }
  // ---- coded block flag and the last significant position (rdo.c:1774-1833) ----
  double best_cost;
  int best_last_idx_p1 = 0;
  {
    const int o_cbf = color == 0 ? M_CBF_LUMA : color == 1 ? M_CBF_CB : M_CBF_CR + (cbf_u ? 1 : 0);
    best_cost = block_uncoded_cost + lambda * rbits(E, o_cbf, 0);
    base_cost += lambda * rbits(E, o_cbf, 1);
  }
  double klast = 0;
  if (lev) {
    const int32_t *last_x_bits = S->last_bits + last_bits_off(E.t, 2, 0), *last_y_bits = S->last_bits + last_bits_off(E.t, 2, 1);
    const double cl = last_x_bits[px] + last_y_bits[py];
    klast = lambda * cl;
  }
  bool found_last = false;
#pragma unroll
  for (int k = 15; k >= 0; --k) {
    if (found_last || k > last_sp) continue;
    const int level = __builtin_amdgcn_readlane(lev, LF_SCAN(k));
    const double s_ = rl64(cs, LF_SCAN(k));
    if (level) {
      const double total = base_cost + rl64(klast, LF_SCAN(k)) - s_;
      if (total < best_cost) { best_last_idx_p1 = k + 1; best_cost = total; }
      if (level > 1) { found_last = true; continue; }
      base_cost -= rl64(cc, LF_SCAN(k));
      base_cost += rl64(c0, LF_SCAN(k));
    } else {
      base_cost -= s_;
    }
  }
  *has_out = best_last_idx_p1 > 0;
  return sp < best_last_idx_p1 ? (coef < 0 ? -lev : lev) : 0;
}

// ---- one 4x4 transform block, start to finish --------------------------------------------------------------------------------------
// predict + uvg_quantize_residual (quant-generic.c:460-612, RDOQ branch) + reconstruction of the 4x4 block of `color` belonging to the
// luma area (x, y) / (lx, ly) of size n (4: the luma block of a 4x4 CU; 8: a chroma block of an 8x8 area), into dst (pitch dp, LDS);
// levels to lv_of(V, color) (LDS) and to co (pitch cp, the CTU's coefficient array).  refs_ready: V's reference rows already belong
// to this block.  Returns has_coeffs, the block's SSD against the source (uvg_pixels_calc_ssd) and every lane's level.
struct lf_block { int has, ssd, level; };       // has_coeffs, SSD (wave-uniform); the level of position lane & 15
template <typename PX> CTU_INLINE1 CTU_DEV lf_block leaf_recon_inl(lds<PX> *S, const job<PX> &J, wctx *V, int color, int mode, int cbf_u, int x, int y, int lx, int ly, int n,
                                                                  int refs_ready, PX *dst_, int dp, int16_t *co, int cp)
{
  const int depth = (int)px_info<PX>::depth;
  const int lane = CTU_TID, e = lane & 15, r = e >> 2, q = e & 3;
  LF_T0();
  if (!refs_ready) leaf_refs(S, J.P, V, color, x, y, lx, ly, n);
  LF_T(5);
  const int pred = leaf_predict(S, V, mode, color);
  const int src = leaf_src(S, color, lx, ly);
  int v = (int)(int16_t)(src - pred);
  v = lf_fwd_pass(v, 2 - 1 + depth - 8);
  v = lf_fwd_pass(v, 2 + 6);
  const int qps = scaled_qp<PX>(J.P, color);
  int has;
  LF_T(6);
  const int level = leaf_rdoq(S, v, color, cbf_u, qps, color ? J.P.c_lambda_tu : J.P.lambda, &has);
  LF_T(7);
  CTU_LDS int16_t *const lv = LDSP(int16_t, lv_of(V, color));
  if (lane < 16) { lv[e] = (int16_t)level; co[r * cp + q] = (int16_t)level; }
  int rec = pred;
  if (has) {
    const int transform_shift = 15 - depth - 2;
    const int shift = 20 - 14 - transform_shift;
    const int32_t scale = (int32_t)kInvQuantScales[qps % 6] << (qps / 6);
    const int32_t add = 1 << (shift - 1);
    int t = clampi((level * scale + add) >> shift, -32768, 32767);          // uvg_dequant, quant-generic.c:618-669
    t = lf_inv_pass(t, 7);
    t = lf_inv_pass(t, 12 - (depth - 8));
    const int16_t val = (int16_t)(t + pred);
    rec = clampi(val, 0, (int)px_info<PX>::maxv);
  }
  if (lane < 16) LDSP(PX, dst_)[r * dp + q] = (PX)rec;
  int dd = src - rec;
  dd = lane < 16 ? dd * dd : 0;
  lf_block B;
  B.has = has; B.level = level;
  B.ssd = __builtin_amdgcn_readfirstlane(lf_row_sum(dd)) >> (2 * (depth - 8));
  CTU_SYNC();
  LF_T(8);
  return B;
}
// (an out-of-line call saves and reloads ~45 callee-saved VGPRs through the stack -- 2.5 k cycles a block: the CU evaluations have ONE
// call site each, a loop over the colours around the inlined body; the rare helper goes through this copy)
template <typename PX> CTU_NOINLINE CTU_DEV lf_block leaf_recon(lds<PX> *S, const job<PX> &J, wctx *V, int color, int mode, int cbf_u, int x, int y, int lx, int ly, int n,
                                                               int refs_ready, PX *dst_, int dp, int16_t *co, int cp)
{
  return leaf_recon_inl(S, J, V, color, mode, cbf_u, x, y, lx, ly, n, refs_ready, dst_, dp, co, cp);
}
